#!/usr/bin/env python3
"""Where a block of the R drivers' loop spends its time: the phases of one increment_ibs_counts call (upload of 26 843
columns, view, kernel, the two count matrices into the caller's) from anonymous memory and from a fresh file mapping."""
import os, sys, time, tempfile
sys.path.insert(0, ".")
import numpy as np
import tidypopgen_amd as tpg
n, m, nb = 5000, 26843, 10
ctx = tpg.default_context()
rng = np.random.default_rng(1)
a = np.asfortranarray(rng.integers(0, 3, size=(n, m * nb), dtype=np.uint8))
pw = tpg.Pairwise(ctx, n)
rows = np.arange(1, n + 1, dtype=np.int32)
def t(f, reps=5):
    best = 1e9
    for _ in range(reps):
        ctx.sync(); t0 = time.perf_counter(); r = f(); ctx.sync(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
blk = np.asfortranarray(a[:, :m])
ms, X = t(lambda: tpg.FBM.from_numpy(blk)); print(f"upload 134 MB: {ms:.2f} ms")
ms, v = t(lambda: tpg.View(X, None, None, code256=None)); print(f"view: {ms:.2f} ms")
ms, _ = t(lambda: (pw.zero(), pw.accumulate(v, products=tpg.PW_FOR_IBS))); print(f"zero + accumulate: {ms:.2f} ms")
for label, src in (("anonymous memory", a), ("fresh file mapping", None)):
    if src is None:
        d = tempfile.mkdtemp(dir="/tmp")
        f = os.path.join(d, "g.bk"); a.T.tofile(f)  # column-major bytes
        src = np.memmap(f, dtype=np.uint8, mode="r", shape=(n, m * nb), order="F")
    for flush in (True, False):
        K = np.zeros((n, n), order="F"); K2 = np.zeros((n, n), order="F")
        ctx.sync(); times = []
        for b in range(nb):
            cols = np.arange(b * m + 1, (b + 1) * m + 1, dtype=np.int32)
            t0 = time.perf_counter()
            tpg.increment_ibs_counts(K, K2, src, rows, cols, flush=flush)
            times.append((time.perf_counter() - t0) * 1e3)
        if not flush:
            tpg.increment_flush()
        print(f"{label}, flush={flush}: first call {times[0]:.2f} ms, then {np.mean(times[1:]):.2f} ms per block (median {np.median(times[1:]):.2f})")
    tpg.resident_drop()
import shutil; del src; shutil.rmtree(d, ignore_errors=True)
