#!/usr/bin/env python3
"""The shader clock under the pairwise kernels, per launch, from one `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES`
pass (the counter file carries every dispatch's start and end time):
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pwclk -- python3 tools/pw_clock_probe.py run
    python3 tools/pw_clock_probe.py report gpurun_out/pwclk
`run` launches, in this order and three times each: the five-product kernel and the {V, D} kernel on the synthetic panel, on a
panel of constant genotypes (all dosage 0) and on an all-missing panel, then the {V, D} kernel without its loads and without loads
and plane masks (TPG_PW_VARIANT=21 / 22) on the synthetic panel.  `report` prints duration, clock (GRBM_GUI_ACTIVE / 8 XCDs /
duration) and MFMA pipe occupancy per launch."""
import collections
import csv
import glob
import os
import sys

PLAN = [("synthetic panel", "random", 0), ("all dosage 0", "zero", 0), ("all missing", "miss", 0),
        ("synthetic, loads removed", "random", 21), ("synthetic, loads + masks removed", "random", 22)]
n, m = 5000, 400000

if sys.argv[1] == "run":
    sys.path.insert(0, ".")
    import numpy as np
    import tidypopgen_amd as tpg

    ctx = tpg.default_context()
    panels = {}
    for label, kind, var in PLAN:
        if kind not in panels:
            panels[kind] = (tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True) if kind == "random" else
                            tpg.FBM.from_numpy(np.full((n, m), 0 if kind == "zero" else 3, dtype=np.uint8, order="F")))
        v = tpg.View(panels[kind], code256=None)
        pw = tpg.Pairwise(ctx, n)
        os.environ["TPG_PW_VARIANT"] = str(var)
        for products in ((None, tpg.PW_FOR_AS) if var == 0 else (tpg.PW_FOR_AS,)):
            for _ in range(3):
                pw.zero(); pw.accumulate(v, products=products)
        ctx.sync()
        pw.free(); v.free()
else:
    rows = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(sys.argv[2], "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "pairwise_kernel" not in k and "pairwise_set_kernel" not in k:
                continue
            d = rows[int(r["Dispatch_Id"])]
            d["kernel"] = k
            d["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            d[r["Counter_Name"]] = float(r["Counter_Value"])
    seq = [rows[k] for k in sorted(rows)]
    # the plan's launches in order: (five products x 3, {V, D} x 3) for the first three entries, {V, D} x 3 for the timing variants
    labels = []
    for label, kind, var in PLAN:
        labels += [label + ", all five"] * 3 + [label + ", {V, D}"] * 3 if var == 0 else [label + ", {V, D}"] * 3
    for lab, d in zip(labels, seq):
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        print(f"{lab:46s} {d['ms']:7.3f} ms  {cyc / d['ms'] / 1e6:5.3f} GHz  MFMA pipe busy {d['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:5.3f}   {d['kernel'][:40]}")
