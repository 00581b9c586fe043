#!/usr/bin/env python3
"""K split of the five-product pairwise kernel against HBM-side traffic, held clock and time: does a K
window that fits the 256-MB Infinity Cache cut the kernel's 25 GB of fetches, and does the clock the chip holds rise with it?
    python3 tools/pw_ksplit_probe.py run                      (three launches per K split, in the order of SPLITS)
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/ks_fetch -- python3 tools/pw_ksplit_probe.py run
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/ks_clk -- python3 tools/pw_ksplit_probe.py run
    python3 tools/pw_ksplit_probe.py report gpurun_out/ks_fetch gpurun_out/ks_clk
A wave-unit's K range is M / S loci; the K-split index is the slow index of the unit order, so all ~4 200 units of one K window
(M / S loci x 157 row tiles x 1 KiB per 64 loci = 2.5 GB / S of operands) run before the next window starts."""
import collections
import csv
import glob
import os
import sys

SPLITS = [5, 7, 10, 14, 20, 28, 40, 64]
REPS = 3
n, m = 5000, 1_000_000

if sys.argv[1] == "run":
    sys.path.insert(0, ".")
    import tidypopgen_amd as tpg

    ctx = tpg.default_context()
    ctx.prof_enable(True)
    X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
    v = tpg.View(X, code256=None)
    pw = tpg.Pairwise(ctx, n)
    pw.zero(); pw.accumulate(v); ctx.sync()  # (warm-up: the T4 layout, the unit table)
    for S in SPLITS:
        os.environ["TPG_PW_KSPLIT"] = str(S)
        best = 1e9
        for _ in range(REPS):
            ctx.prof_reset(); pw.zero(); pw.accumulate(v); ctx.sync()
            best = min(best, ctx.prof_dump()["pairwise_mfma"][1])
        print(f"S = {S:3d}: window {m // S:7d} loci = {2.5e3 / S:6.1f} MB of operands, best of {REPS}: {best:.3f} ms = "
              f"{5.0 * n * n * m / best / 1e13:.3f} of the FP4 peak", flush=True)
else:
    def launches(d):
        rows = collections.defaultdict(dict)
        for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if "tpg_pairwise_kernel" not in r["Kernel_Name"]:
                    continue
                q = rows[int(r["Dispatch_Id"])]
                q["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                q[r["Counter_Name"]] = q.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        return [rows[k] for k in sorted(rows)][1:]  # (without the warm-up launch)

    fetch, clk = launches(sys.argv[2]), launches(sys.argv[3])
    print("S    window MB   ms (pmc passes)   L2-miss GB/launch (FETCH_SIZE)   clock GHz   MFMA pipe busy")
    for i, S in enumerate(SPLITS):
        f = fetch[REPS * i:REPS * i + REPS]
        c = clk[REPS * i:REPS * i + REPS]
        # FETCH_SIZE: units of 1 024 B, and on gfx950 half the bytes of 16-byte-per-lane reads (MI355X_MICROARCH.md "HBM";
        # tools/pmc_summary.py): x 2.  It counts the L2s' memory-side requests, Infinity-Cache hits INCLUDED: what it shows is
        # how much the 4-MB L2s of the XCDs miss, not what reaches HBM
        gb = sum(x["FETCH_SIZE"] for x in f) / len(f) * 1024 * 2 / 1e9
        ms = min(x["ms"] for x in c)
        cyc = [x["GRBM_GUI_ACTIVE"] / 8 for x in c]
        ghz = sum(cy / x["ms"] / 1e6 for cy, x in zip(cyc, c)) / len(c)
        busy = sum(x["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cy for cy, x in zip(cyc, c)) / len(c)
        print(f"{S:3d}  {2.5e3 / S:8.1f}   {ms:8.3f}          {gb:8.2f}              {ghz:5.3f}      {busy:5.3f}")
