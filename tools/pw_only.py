#!/usr/bin/env python3
"""Time the pairwise cross-product kernel alone (HIP events inside the library)."""
import sys
sys.path.insert(0, ".")
import tidypopgen_amd as tpg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
ctx = tpg.default_context()
ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
v = tpg.View(X, code256=None)
pw = tpg.Pairwise(ctx, n)
for rep in range(3):
    ctx.prof_reset()
    pw.zero(); pw.accumulate(v); ctx.sync()
    ms, cnt = ctx.prof_get("pairwise_mfma")
    print(f"rep{rep}: {ms:.3f} ms  {5.0*n*n*m/ms/1e9:.1f} TOP/s", flush=True)
