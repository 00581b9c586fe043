#!/usr/bin/env python3
"""Quick per-kernel timing probe (HIP events inside the library).  Not the bench contract."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import tidypopgen_amd as tpg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
G = 51
ctx = tpg.default_context()
ctx.prof_enable(True)
t0 = time.time()
X = tpg.FBM.synth(3, n, m, npop=G, imputed_bytes=True)
ctx.sync()
print("synth", time.time() - t0)
for rep in range(2):
    ctx.prof_reset()
    t0 = time.time()
    v = tpg.View(X, code256=None)
    ctx.sync()
    t1 = time.time()
    pw = tpg.Pairwise(ctx, n)
    pw.accumulate(v)
    ctx.sync()
    t2 = time.time()
    k = pw.king()
    g = pw.grm()
    t3 = time.time()
    gid = (np.arange(n) % G).astype(np.int32)
    f = tpg.loci_alt_freq(X)
    t4 = time.time()
    h = tpg.pairwise_pop_fst(X, None, None, gid, G, method="Hudson")
    w = tpg.pairwise_pop_fst(X, None, None, gid, G, method="WC84")
    t5 = time.time()
    print(f"rep{rep}: view {t1-t0:.3f}s pairwise {t2-t1:.3f}s epilogues {t3-t2:.3f}s alt_freq {t4-t3:.3f}s fst {t5-t4:.3f}s")
    for name, (cnt, ms) in sorted(ctx.prof_dump().items()):
        print(f"   {name:24s} x{cnt:3d} {ms:10.3f} ms")
    ms, _ = ctx.prof_get("pairwise_mfma")
    ops = 5.0 * n * n * m  # 2.5 N^2 M MACs
    print(f"   pairwise int8: {ops/ms/1e9:.1f} TOP/s algorithmic (5 N^2 M ops), {ops/ms/1e9/5000*100:.1f}% of 5 POP/s")
