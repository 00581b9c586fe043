"""CPU-side sanitizer jobs: the equivalent of the reference's valgrind check of its native code
(/root/reference/.github/workflows/R-CMD-check-valgrind.yaml:18,50-51) for what this repo runs on the HOST -- GPU
AddressSanitizer is not available on the pool, and the host side is where thread teams, a condition-variable transport,
a session table of mapped files and caches live:

  * tidypopgen_amd/csrc/host/*.h (the b x b eigen solver of the PCA, band arithmetic of the sharded pairwise slabs, the
    greedy loop of filter_high_relatedness, the nibble pack of the FBM upload, the in-process all-reduce of tpg_multi_*)
    through tests/host/host_pieces.cpp,
    with -fsanitize=address,undefined and, for the transport, -fsanitize=thread;
  * oracle/tpg_oracle.c with -fsanitize=address,undefined under its golden-vector tests;
  * shim/tpg_rshim.c + tests/rmock/rmock.c with -fsanitize=address,undefined against a host stand-in for the library
    (tests/host/tpg_stub.c), driven like the R drivers drive it (tests/host/drive_shim_san.py)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")
CSRC = os.path.join(ROOT, "tidypopgen_amd", "csrc")
SAN = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
           TSAN_OPTIONS="halt_on_error=1")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")


def _sh(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, **kw)
    assert r.returncode == 0, (" ".join(cmd) + "\n" + r.stdout[-3000:] + r.stderr[-6000:])
    return r


def _asan_runtime():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip("gcc has no shared ASan runtime to preload into python")
    return os.path.realpath(p)


@pytest.fixture(scope="module")
def pieces(tmp_path_factory):
    d = tmp_path_factory.mktemp("host_pieces")
    out = {}
    for name, flags in (("asan", SAN), ("tsan", ["-O1", "-g", "-fsanitize=thread"])):
        exe = str(d / f"host_pieces_{name}")
        _sh(["g++", "-std=c++17", *flags, "-Wall", "-Wextra", "-I" + CSRC, os.path.join(HOST, "host_pieces.cpp"), "-o", exe, "-pthread"])
        out[name] = exe
    return out


@pytest.mark.parametrize("what", ["eig", "bands", "relfilter", "nibpack", "bedpack", "addcounts", "fsttiles", "bits2", "inproc", "inproc_mismatch"])
def test_host_pieces_under_address_and_undefined_sanitizers(pieces, what):
    r = _sh([pieces["asan"], what], env=ENV)
    assert r.stdout.strip() == f"ok {what}"


@pytest.mark.parametrize("args", [["inproc", "2"], ["inproc", "8"], ["inproc_mismatch"]])
def test_in_process_transport_under_thread_sanitizer(pieces, args):
    r = subprocess.run([pieces["tsan"], *args], capture_output=True, text=True, timeout=600, env=ENV)
    if r.returncode != 0 and "FATAL: ThreadSanitizer" in r.stderr and "memory layout" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]


def test_oracle_golden_vectors_under_sanitizers(tmp_path):
    so = str(tmp_path / "libtpg_oracle_san.so")
    _sh(["gcc", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", *SAN, "-shared", "-o", so,
         os.path.join(ROOT, "oracle", "tpg_oracle.c"), "-lm"])
    env = dict(ENV, TPG_ORACLE_SO=so, LD_PRELOAD=_asan_runtime(), OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-4000:]


def test_shim_table_cache_and_uploads_under_sanitizers(tmp_path):
    so = str(tmp_path / "libshim_san.so")
    _sh(["gcc", "-std=c11", *SAN, "-Wall", "-Wextra", "-Werror", "-Wno-cast-function-type", "-shared", "-fPIC",
         "-I" + os.path.join(ROOT, "tests", "rmock"), "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "shim", "tpg_rshim.c"),
         os.path.join(ROOT, "tests", "rmock", "rmock.c"), os.path.join(HOST, "tpg_stub.c"), "-o", so])
    env = dict(ENV, LD_PRELOAD=_asan_runtime())
    r = subprocess.run([sys.executable, os.path.join(HOST, "drive_shim_san.py"), so, str(tmp_path)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok shim under sanitizers" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-4000:]
