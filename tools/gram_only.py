#!/usr/bin/env python3
"""Time the PCA Gram kernel alone (HIP events inside the library) and check it against an FP64 numpy Gram of 64 rows."""
import sys
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
for rep in range(3):
    ctx.prof_reset()
    K = tpg.pca_gram(v, center, scale)
    ctx.sync()
    for name, (c, ms) in sorted(ctx.prof_dump().items()):
        if "gram_mfma" in name or "gram_classes" in name or "gcls" in name:
            print(f"rep{rep} {name}: {c} launches {ms:.3f} ms", flush=True)
print("trace", np.trace(K), "sym", np.array_equal(K, K.T), "checksum", float(np.abs(K).sum()))
