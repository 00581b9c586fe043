#!/usr/bin/env python3
"""K split of the eigen solver's K Q product (tpg_symm_apply_kernel): time per launch over TPG_EIG_S = 4 ... 32."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import tidypopgen_amd as tpg
n, m = 5000, 200000
ctx = tpg.default_context(); ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
for S in [int(x) for x in sys.argv[1:]] or (0, 4, 8, 16, 32):
    if S: os.environ["TPG_EIG_S"] = str(S)
    for rep in range(2):
        ctx.prof_reset()
        r = tpg.gt_pca_partialSVD(X, None, None, k=20, code256=tpg.CODE_IMPUTE_PRED)
        d = ctx.prof_dump()
    print("S", S or "default", "symm_apply %.3f ms / %d launches = %.1f us each; d[0]=%.6f" % (d["eig_symm_apply"][1], d["eig_symm_apply"][0], 1e3 * d["eig_symm_apply"][1] / d["eig_symm_apply"][0], r["d"][0]))
