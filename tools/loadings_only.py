#!/usr/bin/env python3
"""The loadings kernel (v = Z'u / d, int8 MFMA on six u digits) alone on the bench panel: HIP-event time inside the library.
tools/loadings_only.py [n] [m] [k]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import tidypopgen_amd as tpg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = tpg.default_context()
ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
v = tpg.View(X, None, None, code256=np.ascontiguousarray(tpg.CODE_IMPUTE_PRED))
center, scale = tpg.pca_center_scale(v)
rng = np.random.default_rng(1)
U = np.linalg.qr(rng.standard_normal((n, k)))[0]
d = np.linspace(2e3, 1e3, k)
ref = None
best = 1e9
for rep in range(4):
    ctx.prof_reset()
    V = tpg.pca_loadings(v, center, scale, U, d)
    ctx.sync()
    p = ctx.prof_dump()
    best = min(best, p["loadings_mfma"][1])
print(f"loadings_mfma {best:.4f} ms   finalize {p['loadings_finalize'][1]:.4f}  digits {p['loadings_u_digits'][1]:.4f}   checksum {float(np.abs(V).sum()):.10e}", flush=True)
