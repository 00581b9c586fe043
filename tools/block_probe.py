import sys, time
sys.path.insert(0, ".")
import numpy as np
import tidypopgen_amd as tpg
n, m = 5000, 26843
ctx = tpg.default_context()
rng = np.random.default_rng(1)
a = np.asfortranarray(rng.integers(0, 3, size=(n, m), dtype=np.uint8))
pw = tpg.Pairwise(ctx, n)
K = np.zeros((n, n), order="F"); K2 = np.zeros((n, n), order="F")
rows = np.arange(1, n + 1, dtype=np.int32); cols = np.arange(1, m + 1, dtype=np.int32)
def t(f, reps=5):
    best = 1e9
    for _ in range(reps):
        ctx.sync(); t0 = time.perf_counter(); r = f(); ctx.sync(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
ms, X = t(lambda: tpg.FBM.from_numpy(a)); print(f"upload 134 MB: {ms:.2f} ms")
ms, v = t(lambda: tpg.View(X, None, None, code256=None)); print(f"view: {ms:.2f} ms")
ms, _ = t(lambda: (pw.zero(), pw.accumulate(v, products=tpg.PW_FOR_IBS))); print(f"zero + accumulate: {ms:.2f} ms")
ms, _ = t(lambda: tpg.increment_ibs_counts(K, K2, a, rows, cols)); print(f"increment_ibs_counts (flush): {ms:.2f} ms")
ms, _ = t(lambda: tpg.increment_ibs_counts(K, K2, a, rows, cols, flush=False)); print(f"increment_ibs_counts (deferred): {ms:.2f} ms")
tpg.increment_flush()
