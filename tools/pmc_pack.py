#!/usr/bin/env python3
"""Per-launch means (in millions) of the counters rocprofv3 collected for the pack kernels.
usage: tools/pmc_pack.py <dir under gpurun_out> ...   (each from `rocprofv3 --pmc <counters> --output-format csv -d gpurun_out/<dir> -- python3 tools/pack_only.py`)"""
import csv, glob, collections, sys
for x in sys.argv[1:]:
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{x}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pack_fast" in r["Kernel_Name"]: acc[r["Kernel_Name"][:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,c in acc.items():
        print(x, k, {n:(round(sum(v)/len(v)/1e6,3), len(v)) for n,v in c.items()})
