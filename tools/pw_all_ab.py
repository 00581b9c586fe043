import os, sys
sys.path.insert(0, ".")
import tidypopgen_amd as tpg
n, m = 5000, 1000000
ctx = tpg.default_context(); ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
v = tpg.View(X, code256=None)
pw = tpg.Pairwise(ctx, n)
ref = None
import numpy as np
for var in [int(x) for x in sys.argv[1:]] or (0, 2, 0, 2):
    os.environ["TPG_PW_VARIANT"] = str(var)
    best = 1e9
    for rep in range(3):
        ctx.prof_reset(); pw.zero(); pw.accumulate(v); ctx.sync()
        best = min(best, ctx.prof_dump()["pairwise_mfma"][1])
    c = pw.counts(("ibs", "king_num"))
    if ref is None: ref = c
    print(f"all variant {var}: {best:.3f} ms  same counts: {np.array_equal(c['ibs'], ref['ibs']) and np.array_equal(c['king_num'], ref['king_num'])}", flush=True)
