#!/usr/bin/env python3
"""Is the pairwise MFMA rate set by the DATA?  The same kernels, the same instruction stream, on panels whose operand planes
toggle differently: the synthetic panel (random genotypes), all genotypes 0 (v = 1, d = -1 everywhere: constant operand words),
all heterozygous (d = 0), all missing (every operand word zero).  If the chip's clock under the FP4 MFMAs is a power limit, the
constant panels run faster although nothing in the code path changes.   tools/pw_power_probe.py [n] [m]"""
import os
import sys
sys.path.insert(0, ".")
import numpy as np
import tidypopgen_amd as tpg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
ctx = tpg.default_context()
ctx.prof_enable(True)
ops = {"pairwise_mfma": 5.0, "pairwise_mfma_as": 2.0, "pairwise_mfma_ibs": 3.0, "pairwise_mfma_king": 4.0}
sets = ((None, "pairwise_mfma"), (tpg.PW_FOR_AS, "pairwise_mfma_as"), (tpg.PW_FOR_IBS, "pairwise_mfma_ibs"), (tpg.PW_FOR_KING, "pairwise_mfma_king"))


def panel(kind):
    if kind == "random":
        return tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
    val = {"all 0": 0, "all 1 (het)": 1, "all 2": 2, "all missing": 3}[kind]
    return tpg.FBM.from_numpy(np.full((n, m), val, dtype=np.uint8, order="F"))


for kind in ("random", "all 0", "all 1 (het)", "all missing", "random"):
    X = panel(kind)
    v = tpg.View(X, code256=None)
    pw = tpg.Pairwise(ctx, n)
    line = f"{kind:12s}"
    for products, key in sets:
        best = 1e9
        for _ in range(3):
            ctx.prof_reset()
            pw.zero(); pw.accumulate(v, products=products); ctx.sync()
            best = min(best, ctx.prof_dump()[key][1])
        line += f"  {key[9:] or 'all':9s} {best:7.3f} ms = {ops[key] * n * n * m / best / 1e13:.3f}"
    print(line, flush=True)
    pw.free(); v.free(); X.free()
