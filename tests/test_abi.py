"""CPU-only: the C-ABI library loads, exports every symbol include/tpg.h declares, and fails
loudly (no CPU fallback) when no HIP device is usable."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "tpg.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tpg_[A-Za-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from tidypopgen_amd import _lib

    syms = _header_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(_lib.lib, s)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS) == syms  # the python binding lists exactly the header's surface


def test_no_cpu_fallback_without_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import tidypopgen_amd as tpg

    with pytest.raises(tpg._lib.TpgError) as e:
        tpg.Context(0)
    assert "no HIP device" in str(e.value) or "hip" in str(e.value).lower()


def test_product_does_not_import_the_oracle():
    # the oracle is test infrastructure: nothing under tidypopgen_amd/ may reference it
    pkg = os.path.join(ROOT, "tidypopgen_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "tpg_oracle" not in src.replace(
                    "oracle/tpg_oracle.c: orc_synth_fbm", ""), f
