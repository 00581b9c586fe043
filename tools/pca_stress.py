#!/usr/bin/env python3
"""PCA robustness sweep on the GPU against the numpy (LAPACK) oracle: shapes x k x population structure."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import tidypopgen_amd as tpg
from oracle import oracle as orc

bad = 0
for n, m in ((40, 300), (128, 1000), (333, 2500), (1000, 4000)):
    for G in (1, 2, 10):
        fbm = orc.synth_fbm(1000 + n + G, n, m, npop=G, miss=0.0)
        X = tpg.FBM.from_numpy(fbm)
        alt = orc.CODE_012[fbm].sum(axis=0)
        keep = np.where((alt > 0) & (alt < 2 * n))[0] + 1
        for k in (1, 2, 5, 14, 30, 52):
            if k >= min(n, len(keep)):
                continue
            t0 = time.time()
            try:
                r = tpg.gt_pca_partialSVD(X, None, keep.astype(np.int32), k=k, code256=tpg.CODE_012)
            except Exception as e:  # noqa: BLE001
                print(f"n={n} m={m} G={G} k={k}: FAILED {e}")
                bad += 1
                continue
            dt = time.time() - t0
            o = orc.gt_pca_partialSVD(fbm, None, keep.astype(np.int32), k=k, code256=orc.CODE_012)
            ed = np.max(np.abs(r["d"] - o["d"]) / o["d"][0])
            # subspace agreement for well separated values only
            gaps = np.abs(np.diff(np.append(o["d"], o["d"][-1] * 0.0)))
            sep = gaps / o["d"][0] > 1e-6
            eu = 0.0
            for j in range(k):
                if sep[j] and (j == 0 or sep[j - 1]):
                    eu = max(eu, min(np.max(np.abs(r["u"][:, j] - o["u"][:, j])), np.max(np.abs(r["u"][:, j] + o["u"][:, j]))))
            ev = 0.0
            for j in range(k):
                if sep[j] and (j == 0 or sep[j - 1]):
                    ev = max(ev, min(np.max(np.abs(r["v"][:, j] - o["v"][:, j])), np.max(np.abs(r["v"][:, j] + o["v"][:, j]))))
            flag = "" if (ed < 1e-6 and eu < 1e-5 and ev < 1e-5) else "  <-- CHECK"
            if flag:
                bad += 1
            print(f"n={n} m={m} G={G} k={k}: d err {ed:.2e}  u err {eu:.2e}  v err {ev:.2e}  {dt*1e3:.0f} ms{flag}")
print("bad:", bad)
