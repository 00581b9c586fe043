"""GPU: the sharded (N > 1) step gives the same answers as the unsharded one on the same panel.
Two ranks share the single GPU of the test box and exchange through gloo (RCCL refuses two ranks on one
device); the data path (shard ranges, integer partials, Fst sums, Gram all-reduce, replicated eigen step) is
the one the RCCL run uses."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env_extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **env_extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r


@pytest.mark.timeout(900)
def test_two_shards_equal_one(tmp_path):
    d1, d2 = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    common = ["--steps", "1", "--warmup", "0", "--indiv", "700", "--pops", "9", "--k", "8", "--no-cpu-baseline"]
    _run([sys.executable, "bench.py", "--gpus", "1", "--snps", "60000", "--digest", d1] + common, {})
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
          "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2", "--snps", "30000", "--digest", d2] + common,
         {"TPG_BENCH_BACKEND": "gloo", "TPG_BENCH_SHARE_GPU": "1"})
    a, b = json.load(open(d1)), json.load(open(d2))
    # integer cross-products are exact, so every epilogue value is identical
    for name in ("ibs", "king", "grm"):
        assert a[name + "_nan"] == b[name + "_nan"]
        assert np.allclose(a[name + "_corner"], b[name + "_corner"], rtol=1e-13, atol=0, equal_nan=True), name
        assert a[name + "_sum"] == pytest.approx(b[name + "_sum"], rel=1e-12)
    for name in ("fst_hudson", "fst_wc84"):
        assert np.allclose(a[name], b[name], rtol=1e-12, atol=0)
    assert np.allclose(a["pca_d"], b["pca_d"], rtol=1e-7)  # Gram partials use per-shard weight scaling
    assert a["pca_fro"] == pytest.approx(b["pca_fro"], rel=1e-12)
    assert np.allclose(a["pca_u_abs_colsum"], b["pca_u_abs_colsum"], rtol=1e-5)
