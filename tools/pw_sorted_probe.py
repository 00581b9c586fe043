#!/usr/bin/env python3
"""Does the ORDER of the loci change the rate of the pairwise kernels?  The sums do not depend on it; the operand words the
matrix cores see one after the other do: with the loci sorted by allele count, neighbouring blocks carry statistically similar
genotypes (fewer bit flips between successive operands), and the clock under the FP4 MFMAs is a power limit
(tools/pw_power_probe.py).  Same panel, same loci, three orders: as generated, sorted by allele count, random permutation."""
import sys
sys.path.insert(0, ".")
import numpy as np
import tidypopgen_amd as tpg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
ctx = tpg.default_context()
ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
cnt = tpg.loci_counts(tpg.View(X, code256=None))
alt = cnt[:, 1] + 2 * cnt[:, 2]
orders = {"as generated": None, "sorted by allele count": (np.argsort(alt, kind="stable") + 1).astype(np.int32),
          "random permutation": (np.random.default_rng(1).permutation(m) + 1).astype(np.int32)}
ref = None
for name, cols in orders.items():
    v = tpg.View(X, None, cols, code256=None)
    pw = tpg.Pairwise(ctx, n)
    line = f"{name:24s}"
    for products, key, ops in ((None, "pairwise_mfma", 5.0), (tpg.PW_FOR_AS, "pairwise_mfma_as", 2.0)):
        best = 1e9
        for _ in range(3):
            ctx.prof_reset()
            pw.zero(); pw.accumulate(v, products=products); ctx.sync()
            best = min(best, ctx.prof_dump()[key][1])
        line += f"  {key[9:]:8s} {best:7.3f} ms = {ops * n * n * m / best / 1e13:.3f}"
    c = pw.counts(("as_num",))["as_num"]
    if ref is None:
        ref = c
    print(line, " same counts:", bool(np.array_equal(c, ref)), flush=True)
    pw.free(); v.free()
