import sys, numpy as np
sys.path.insert(0, ".")
import tidypopgen_amd as tpg
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5000, 1000000)
X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
vi = tpg.View(X, None, None, code256=np.ascontiguousarray(tpg.CODE_IMPUTE_PRED))
cnt = tpg.loci_counts(vi); alt = cnt[:, 1] + 2 * cnt[:, 2]
cols = (np.where((alt > 0) & (alt < 2 * n))[0] + 1).astype(np.int32)
for rep in range(2):
    print("=== rep", rep, file=sys.stderr)
    r = tpg.gt_pca_partialSVD(X, None, cols, k=20)
