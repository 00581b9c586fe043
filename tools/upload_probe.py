#!/usr/bin/env python3
"""Cold-start cost: host FBM bytes -> HBM (tpg_fbm_from_host / open_bk), GB/s."""
import sys, time, os, tempfile
import numpy as np
sys.path.insert(0, ".")
import tidypopgen_amd as tpg

n, m = 5000, int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
ctx = tpg.default_context()
rng = np.random.default_rng(0)
a = np.asfortranarray(rng.integers(0, 3, size=(n, m), dtype=np.uint8))
for rep in range(3):
    t0 = time.perf_counter(); X = tpg.FBM.from_numpy(a); ctx.sync(); dt = time.perf_counter() - t0
    print(f"from_numpy {a.nbytes/1e9:.2f} GB in {dt*1e3:.0f} ms = {a.nbytes/dt/1e9:.1f} GB/s")
    if rep == 0:
        assert np.array_equal(X.to_numpy(), a)
    del X
d = tempfile.mkdtemp(dir="/tmp")
path = os.path.join(d, "panel.bk")
a.T.tofile(path)  # column-major bytes
for rep in range(2):
    t0 = time.perf_counter(); X = tpg.FBM.open_bk(path, n, m); ctx.sync(); dt = time.perf_counter() - t0
    print(f"open_bk (page cache warm) {a.nbytes/1e9:.2f} GB in {dt*1e3:.0f} ms = {a.nbytes/dt/1e9:.1f} GB/s")
    del X
os.remove(path); os.rmdir(d)
