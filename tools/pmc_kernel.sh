#!/bin/bash
# PMC passes (separate, never with a trace domain) over one of the single-kernel probes.
# usage (GPU box, repo root): bash tools/pmc_kernel.sh <tag> tools/<probe>.py [args]
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/p1" -- python3 "$@" > "$out/p1.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum WRITE_SIZE --output-format csv -d "$out/p2" -- python3 "$@" > "$out/p2.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$out/p3" -- python3 "$@" > "$out/p3.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d "$out/p4" -- python3 "$@" > "$out/p4.log" 2>&1
python3 - "$out" <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(k, f)].add(r["Dispatch_Id"])
for k, c in sorted(acc.items()):
    if not any(s in k for s in ("pairwise_kernel", "pairwise_set_kernel", "gram_kernel", "gram2_kernel", "gram3_kernel", "gram1w_kernel", "gcls", "wc84", "hudson", "pack_fast")): continue
    d = max(len(v) for (kk, f), v in nd.items() if kk == k)
    print(k, "dispatches", d)
    for n, v in sorted(c.items()): print("   %-28s %.4g per launch" % (n, v / d))
PY
