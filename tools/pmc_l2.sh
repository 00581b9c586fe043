#!/bin/bash
# L2 hit / miss and HBM-side bytes of the kernels of one probe (one --pmc pass each, never with a trace domain).
# usage (GPU box, repo root): bash tools/pmc_l2.sh <tag> tools/<probe>.py [args]
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
# (TCC_HIT / TCC_MISS / FETCH_SIZE in one pass exceed what the hardware collects at once: rocprofv3 aborts and hangs)
timeout -k 10 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum WRITE_SIZE --output-format csv -d "$out/p1" -- python3 "$@" > "$out/p1.log" 2>&1 || echo "pass failed: see $out/p1.log"
python3 - "$out" <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(k, f)].add(r["Dispatch_Id"])
for k, c in sorted(acc.items()):
    if not any(s in k for s in ("pairwise_kernel", "gram_kernel", "gram2_kernel")): continue
    d = max(len(v) for (kk, f), v in nd.items() if kk == k)
    h, m = c["TCC_HIT_sum"] / d, c["TCC_MISS_sum"] / d
    print(k, "hit %.3f  requests %.3g  miss bytes %.3g GB" % (h / (h + m), h + m, m * 128 / 1e9))
PY
