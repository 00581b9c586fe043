import json,sys
o=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", o["ms_per_step"])
e=o["end_to_end"]
keys=("seconds","seconds_all_runs","upload_s","step_s","download_s","step_and_download_s","upload_GBps","last_block_in_HBM_s","tail_s","agreement_with_resident_step","pca_d_max_rel_diff_vs_bk_route","cpu_quota_throttling_over_all_runs")
for k in ("serial","overlapped"):
    print(k, {a:b for a,b in e[k].items() if a in keys})
b=e["bed"]
print("bed best", b.get("seconds"), b.get("skipped"))
for k in ("serial","overlapped"):
    if k in b: print("bed",k, {a:c for a,c in b[k].items() if a in keys})
