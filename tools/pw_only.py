import sys, time
sys.path.insert(0, ".")
import tidypopgen_amd as tpg
n, m = 5000, 1000000
ctx = tpg.default_context()
ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
v = tpg.View(X, code256=None)
pw = tpg.Pairwise(ctx, n)
for rep in range(3):
    ctx.prof_reset()
    pw.zero(); pw.accumulate(v); ctx.sync()
    ms, cnt = ctx.prof_get("pairwise_mfma")
    print(f"rep{rep}: {ms:.3f} ms  {5.0*n*n*m/ms/1e9:.1f} TOP/s")
