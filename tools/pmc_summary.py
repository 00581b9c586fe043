#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of one bench step into the files committed under profiles/.

usage: pmc_summary.py <dir> <tag>
  <dir> holds one sub-directory per `rocprofv3 --pmc ...` pass (any names) and, optionally, `trace/` from a
  `--kernel-trace --stats` run of the same command.  Writes
    profiles/<tag>_pmc_one_step.json   per-kernel counter sums, dispatch counts and derived figures
    profiles/traffic.json              HBM-side bytes per launch of the two MFMA kernels (read by bench.py)
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are in KiB-like units of
1024 B; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B per lane) coalesced reads, so it is doubled
for the kernels whose loads are all global_load_dwordx4 (the two MFMA kernels) and left as is elsewhere.
"""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for f in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[(k, f)].add(r["Dispatch_Id"])
disp = collections.defaultdict(int)
for (k, f), ids in ndisp.items():
    disp[k] = max(disp[k], len(ids))
avg_ns = {}
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        avg_ns[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"])
# kernels whose global reads are all 16 B per lane (global_load_dwordx4 / LDS-DMA of 16 B): FETCH_SIZE x 2.  The fast pack
# kernel belongs here since round 4 (commit 6dde6ae: 16-byte loads on its 8-byte-aligned columns; round 4's file still had it
# at 1.0 and reported half of its 5.0 GB)
WIDE = ("tpg_pairwise_kernel", "tpg_pairwise_set_kernel", "tpg_pairwise_wg_kernel", "tpg_pca_gram_kernel", "tpg_gcls_gram_kernel",
        "tpg_gcls_gram2_kernel", "tpg_gcls_gram1w_kernel", "tpg_pack_fast_kernel")
out, traffic = {}, {}
for k, c in sorted(acc.items()):
    d = disp[k]
    o = {"dispatches_per_pass": d, "counters_sum": dict(c)}
    der = {}
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8 / d  # the counter sums the 8 XCDs
        der["shader_cycles_per_launch"] = cyc
        if k in avg_ns:
            der["avg_launch_ms_trace"] = avg_ns[k] / 1e6
            der["shader_clock_GHz"] = cyc / avg_ns[k]
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            der["mfma_pipe_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / d / cyc  # 1024 SIMDs
    if c.get("SQ_INSTS_MFMA"):
        der["valu_class_insts_per_mfma"] = c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"] - 1.0
    if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in c:
        der["wait_any_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in c:
        corr = 2.0 if k.startswith(WIDE) else 1.0
        der["hbm_read_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * corr / d
        der["fetch_size_correction"] = corr
    if "WRITE_SIZE" in c:
        der["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024 / d
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum"):
        der["l2_hit_frac"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    o["derived"] = der
    out[k] = o
    if "hbm_read_bytes_per_launch" in der and "hbm_write_bytes_per_launch" in der:
        if k.startswith("tpg_pairwise_kernel"):
            traffic["pairwise_mfma"] = der["hbm_read_bytes_per_launch"] + der["hbm_write_bytes_per_launch"]
        if k.startswith("tpg_pca_gram_kernel"):
            traffic["pca_gram_mfma"] = der["hbm_read_bytes_per_launch"] + der["hbm_write_bytes_per_launch"]
        if k.startswith("tpg_gcls_gram_kernel") or k.startswith("tpg_gcls_gram2_kernel"):
            traffic["pca_gram_classes"] = der["hbm_read_bytes_per_launch"] + der["hbm_write_bytes_per_launch"]
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_pmc_one_step.json"), "w"), indent=1)
import datetime
import subprocess

try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip() or "unknown"
except OSError:
    commit = "unknown"
json.dump({"workload": "5000 x 1000000 per GPU, 51 populations, k = 20", "source": f"profiles/{tag}_pmc_one_step.json",
           "date": datetime.date.today().isoformat(), "commit": commit + " (the tree the summary was made in; the passes ran on its build)",
           "hbm_bytes_per_launch": traffic}, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
for k, o in out.items():
    if any(s in k for s in ("pairwise_kernel", "gram_kernel", "gram2_kernel", "gcls", "t4_expand", "pack_fast", "fst_kernel", "wc84", "grouped_counts")):
        print(k, json.dumps({a: (round(b, 4) if isinstance(b, float) else b) for a, b in o["derived"].items()}))
