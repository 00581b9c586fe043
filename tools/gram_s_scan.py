import os, sys
import numpy as np
sys.path.insert(0, ".")
import tidypopgen_amd as tpg
n, m = 5000, 1000000
ctx = tpg.default_context(); ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
vi = tpg.View(X, None, None, code256=np.ascontiguousarray(tpg.CODE_IMPUTE_PRED))
cnt = tpg.loci_counts(vi); alt = cnt[:, 1] + 2 * cnt[:, 2]
cols = (np.where((alt > 0) & (alt < 2 * n))[0] + 1).astype(np.int32)
v = tpg.View(X, None, cols, code256=np.ascontiguousarray(tpg.CODE_IMPUTE_PRED))
center, scale = tpg.pca_center_scale(v)
os.environ["TPG_DEBUG"] = "1"
K = tpg.pca_gram(v, center, scale)
os.environ.pop("TPG_DEBUG")
for S in (0, 8, 10, 12, 14, 16, 18, 20, 24, 28):
    if S: os.environ["TPG_GRAM_S"] = str(S)
    best = (1e9, 0)
    for rep in range(3):
        ctx.prof_reset(); K = tpg.pca_gram(v, center, scale); ctx.sync(); d = ctx.prof_dump()
        best = min(best, (d["pca_gram_classes"][1] + d["gcls_assemble"][1], d["pca_gram_classes"][1]))
    print(S or "model", "gram+assemble %.3f gram %.3f" % best, flush=True)
