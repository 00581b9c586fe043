#!/usr/bin/env python3
"""Class Gram with the mixed-precision fold (TPG_GRAM_FOLD64=0) against the FP64 fold (=1): time and agreement."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import tidypopgen_amd as tpg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
ctx = tpg.default_context()
ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
vi = tpg.View(X, None, None, code256=np.ascontiguousarray(tpg.CODE_IMPUTE_PRED))
cnt = tpg.loci_counts(vi)
alt = cnt[:, 1] + 2 * cnt[:, 2]
cols = (np.where((alt > 0) & (alt < 2 * n))[0] + 1).astype(np.int32)
v = tpg.View(X, None, cols, code256=np.ascontiguousarray(tpg.CODE_IMPUTE_PRED))
center, scale = tpg.pca_center_scale(v)
res = {}
for mode in ("mixed", "fp64"):
    os.environ["TPG_GRAM_FOLD64"] = "1" if mode == "fp64" else "0"
    for rep in range(3):
        ctx.prof_reset()
        K = tpg.pca_gram(v, center, scale)
        ctx.sync()
        d = ctx.prof_dump()
        print(mode, rep, " ".join(f"{k}={ms:.3f}" for k, (c, ms) in sorted(d.items()) if "gcls" in k or "gram" in k), flush=True)
    res[mode] = K
a, b = res["mixed"], res["fp64"]
print("max |diff| / max |K|:", float(np.abs(a - b).max() / np.abs(b).max()), " relative Frobenius:", float(np.linalg.norm(a - b) / np.linalg.norm(b)))
print("symmetric:", np.array_equal(a, a.T), " trace rel diff:", float(abs(np.trace(a) - np.trace(b)) / np.trace(b)))
