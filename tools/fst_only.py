#!/usr/bin/env python3
"""Time the Fst kernels alone (Hudson and WC84 sums, 51 populations) on the bench panel."""
import sys, numpy as np
sys.path.insert(0, ".")
import tidypopgen_amd as tpg
n, m, G = 5000, 1000000, 51
ctx = tpg.default_context(); ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=G, imputed_bytes=True)
gid = (np.arange(n) % G).astype(np.int32)
for rep in range(3):
    ctx.prof_reset()
    h = tpg.pairwise_pop_fst(X, None, None, gid, G, method="Hudson")
    w = tpg.pairwise_pop_fst(X, None, None, gid, G, method="WC84")
    d = ctx.prof_dump()
print("hudson %.3f wc84 %.3f reduce %.3f" % (d["fst_hudson"][1], d["fst_wc84"][1], d["fst_reduce"][1]))
