#!/usr/bin/env python3
"""What bounds the {V, D} product-set kernel (pairwise_grm / allele sharing alone): the kernel as it is, with the loads of its
loop removed, with loads and plane masks removed (TPG_PW_VARIANT=21 / 22: timing only, wrong sums), the workgroup form with the
operands shared through LDS (14), and the K split (TPG_PW_KSPLIT), all inside one GPU job.   tools/pw_as_probe.py [n] [m]"""
import os
import sys
sys.path.insert(0, ".")
import tidypopgen_amd as tpg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
ctx = tpg.default_context()
ctx.prof_enable(True)
X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
v = tpg.View(X, code256=None)
pw = tpg.Pairwise(ctx, n)


def run(var, ksplit=None, products=tpg.PW_FOR_AS, key="pairwise_mfma_as"):
    os.environ["TPG_PW_VARIANT"] = str(var)
    if ksplit is None:
        os.environ.pop("TPG_PW_KSPLIT", None)
    else:
        os.environ["TPG_PW_KSPLIT"] = str(ksplit)
    best = 1e9
    for _ in range(3):
        ctx.prof_reset()
        pw.zero(); pw.accumulate(v, products=products); ctx.sync()
        best = min(best, ctx.prof_dump()[key][1])
    return best


for label, var in (("as it is", 0), ("loads of the loop removed", 21), ("loads and plane masks removed", 22), ("operands through LDS, 4 stages", 14),
                   ("LDS form without its barrier", 24), ("LDS form without barrier and LDS-DMA", 25), ("LDS form without its LDS-DMA", 26),
                   ("as it is", 0)):
    print(f"{label:38s} {run(var):7.3f} ms", flush=True)
if len(sys.argv) > 3:
    sys.exit(0)
for S in (8, 12, 16, 20, 24, 32, 48):
    print(f"variant 0, K split {S:3d}: {run(0, S):7.3f} ms    LDS form: {run(14, S):7.3f} ms", flush=True)
