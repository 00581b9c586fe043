#!/usr/bin/env python3
"""Time the pairwise cross-product kernels alone (HIP events inside the library): the five-product kernel and the
product-set kernels, every wave-tile variant (TPG_PW_VARIANT).   tools/pw_only.py [n] [m] [variants...]"""
import os
import sys
sys.path.insert(0, ".")
import tidypopgen_amd as tpg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
variants = [int(x) for x in sys.argv[3:]] or [0, 1, 2]
ctx = tpg.default_context()
ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
v = tpg.View(X, code256=None)
pw = tpg.Pairwise(ctx, n)
sets = (("all", None, 5.0, "pairwise_mfma"), ("as", tpg.PW_FOR_AS, 2.0, "pairwise_mfma_as"),
        ("ibs", tpg.PW_FOR_IBS, 3.0, "pairwise_mfma_ibs"), ("ibs1", tpg.PW_FOR_IBS_ALONE, 3.0, "pairwise_mfma_ibs1"),
        ("king", tpg.PW_FOR_KING, 4.0, "pairwise_mfma_king"))
if os.environ.get("PW_ONLY_SETS"):
    sets = tuple(s for s in sets if s[0] in os.environ["PW_ONLY_SETS"].split(","))
for name, products, ops, key in sets:
    for var in variants:
        os.environ["TPG_PW_VARIANT"] = str(var)
        best = 1e9
        for rep in range(3):
            ctx.prof_reset()
            pw.zero(); pw.accumulate(v, products=products); ctx.sync()
            prof = ctx.prof_dump()
            ms = prof[key][1] if key != "pairwise_mfma" else sum(t for k, (c, t) in prof.items() if k == "pairwise_mfma")
            best = min(best, ms)
        print(f"{name:5s} variant {var}: {best:.3f} ms  {ops*n*n*m/best/1e9:.1f} TOP/s = {ops*n*n*m/best/1e9/10000:.3f} of FP4 peak", flush=True)
