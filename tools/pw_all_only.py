#!/usr/bin/env python3
"""The five-product pairwise kernel alone on the bench panel (best of four launches)."""
import sys
sys.path.insert(0, ".")
import tidypopgen_amd as tpg
n, m = 5000, 1000000
ctx = tpg.default_context(); ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
v = tpg.View(X, code256=None)
pw = tpg.Pairwise(ctx, n)
best = 1e9
for rep in range(4):
    ctx.prof_reset(); pw.zero(); pw.accumulate(v); ctx.sync()
    best = min(best, ctx.prof_dump()["pairwise_mfma"][1])
print(f"all five: {best:.3f} ms = {5.0 * n * n * m / best / 1e13:.3f} of the FP4 peak")
