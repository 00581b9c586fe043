#!/usr/bin/env python3
"""Class Gram: the lean one-wave-per-SIMD kernel (TPG_GRAM_KERNEL=14, 34 = interleaved steps) against the two-waves-per-SIMD
kernel (2 = the default, 1 = its block table by scalar loads): time and agreement.  tools/gram3_ab.py [n] [m] [S values...]"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import tidypopgen_amd as tpg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
Ss = [int(x) for x in sys.argv[3:]] or [0]
ctx = tpg.default_context()
ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
vi = tpg.View(X, None, None, code256=np.ascontiguousarray(tpg.CODE_IMPUTE_PRED))
cnt = tpg.loci_counts(vi)
alt = cnt[:, 1] + 2 * cnt[:, 2]
cols = (np.where((alt > 0) & (alt < 2 * n))[0] + 1).astype(np.int32)
v = tpg.View(X, None, cols, code256=np.ascontiguousarray(tpg.CODE_IMPUTE_PRED))
center, scale = tpg.pca_center_scale(v)
os.environ["TPG_GRAM_FOLD64"] = "0"
ref = None
for kern in (os.environ.get("GRAM_AB_KERNELS") or "1,2,14,34").split(","):
    for S in Ss:
        if kern == "2":  # the default: two waves per SIMD, block table through vector loads
            os.environ.pop("TPG_GRAM_KERNEL", None)
        else:
            os.environ["TPG_GRAM_KERNEL"] = kern
        if S:
            os.environ["TPG_GRAM_S"] = str(S)
        else:
            os.environ.pop("TPG_GRAM_S", None)
        best = 1e9
        for rep in range(3):
            ctx.prof_reset()
            K = tpg.pca_gram(v, center, scale)
            ctx.sync()
            d = ctx.prof_dump()
            if "pca_gram_classes" not in d:
                sys.exit("this panel takes the digit kernel (TPG_GRAM_CLASSES=1 forces the class path)")
            best = min(best, d["pca_gram_classes"][1])
        if ref is None:
            ref = K
        print(f"kernel {kern} S={S or 'model'}: {best:.3f} ms   max |diff| / max |K| = {float(np.abs(K - ref).max() / np.abs(ref).max()):.2e}"
              f"  symmetric {np.array_equal(K, K.T)}", flush=True)
        if kern in ("1", "2") and len(Ss) > 1 and S == Ss[1]:
            break
