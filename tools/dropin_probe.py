"""snp_ibs() through the UNMODIFIED R block loop (compiled shim + tests/rmock, bench.py's `dropin` leg) with the per-call
phases of the increment_* mirror on stderr (TPG_INCREMENT_TRACE=1): where a block's ~10 ms go.
usage (GPU box): python tools/dropin_probe.py [n] [m] [reps]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tidypopgen_amd as tpg  # noqa: E402
from tests import rmock  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tmp = tempfile.mkdtemp(prefix="tpg_dropin_probe_", dir=os.environ.get("TMPDIR", "/tmp"))
try:
    X = tpg.FBM.synth(3, n, m, npop=51, miss=0.02, imputed_bytes=False)
    path = os.path.join(tmp, "panel.bk")
    X.to_numpy().T.tofile(path)
    X.free()
    rows = np.arange(1, n + 1, dtype=np.int32)
    cols = np.arange(1, m + 1, dtype=np.int32)
    block = tpg.block_size(n)
    lo, up = tpg.cut_by_size(m, block)
    os.environ["TPG_DEVICES"] = "1"
    os.environ.pop("TPG_RSHIM_DEFERRED", None)
    lib = rmock.build(tmp)
    r = rmock.Session(lib)
    BM = r.fbm(path, n, m, tpg.CODE_012)
    for rep in range(reps):
        files = []
        for nm in ("k", "k2"):
            f = os.path.join(tmp, f"rep{rep}_{nm}.bk")
            np.zeros(n * n).tofile(f)
            files.append(f)
        K, K2 = [r.fbm(f, n, n) for f in files]
        if rep == reps - 1:
            os.environ["TPG_INCREMENT_TRACE"] = "1"
        if rep == reps - 1:  # the loop by hand, with the time of every .Call
            ri = r.int(rows)
            scratch = [r.matrix(np.zeros((n, 1))) for _ in range(3)]
            t0 = time.perf_counter()
            for a, b in zip(lo, up):
                ta = time.perf_counter()
                cb = r.int(cols[a - 1:b])
                tb = time.perf_counter()
                r.call("increment_ibs_counts", K, K2, scratch[0], scratch[1], scratch[2], BM, ri, cb)
                print(f"[python] r.int {1e3 * (tb - ta):.2f} | .Call {1e3 * (time.perf_counter() - tb):.2f} ms", file=sys.stderr, flush=True)
            print(f"by hand: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
            break
        t0 = time.perf_counter()
        rmock.driver_loop(r, "ibs", BM, K, K2, rows, cols, lo, up, scratch_width=1)
        dt = time.perf_counter() - t0
        print(f"rep {rep}: {dt * 1e3:.1f} ms for {len(lo)} blocks of {block} loci = {dt * 1e3 / len(lo):.2f} ms per block", flush=True)
        for f in files:
            os.remove(f)
    lib.R_unload_tpgshim(None)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
