#!/usr/bin/env python3
"""Time the Fst kernels alone (Hudson and WC84 sums, 51 populations) on the bench panel.  TPG_FST_TILES=0: the WC84 totals
kernel with 8 unrelated pairs per thread instead of a tile of populations (A/B; run the script once per setting).
tools/fst_only.py [n] [m] [G]"""
import sys, numpy as np
sys.path.insert(0, ".")
import tidypopgen_amd as tpg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
G = int(sys.argv[3]) if len(sys.argv) > 3 else 51
ctx = tpg.default_context(); ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=G, imputed_bytes=True)
gid = (np.arange(n) % G).astype(np.int32)
for rep in range(3):
    ctx.prof_reset()
    h = tpg.pairwise_pop_fst(X, None, None, gid, G, method="Hudson")
    w = tpg.pairwise_pop_fst(X, None, None, gid, G, method="WC84")
    d = ctx.prof_dump()
print("hudson %.3f wc84 %.3f reduce %.3f" % (d["fst_hudson"][1], d["fst_wc84"][1], d["fst_reduce"][1]))
w = np.asarray(w["fst_tot"] if isinstance(w, dict) else w).ravel()
print("wc84 totals: first %.17g  last %.17g  sum %.17g" % (w[0], w[-1], float(np.nansum(w))))
