#!/usr/bin/env python3
"""Idle time of the GPU between the kernels of a bench run, by the pair of kernels around each gap.
usage: tools/gaps.py <rocprofv3 output dir>  [per-step kernel, default pairwise]  [min gap us, default 5]
(the directory of `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 bench.py ...`)"""
import collections, csv, glob, re, sys

d = sys.argv[1]
step_kernel = sys.argv[2] if len(sys.argv) > 2 else "tpg_pairwise_kernel"
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?")))
ev.sort()
steps = sum(1 for e in ev if e[2].startswith(step_kernel)) or 1
# the steady part of the run: from the first to the last launch of the per-step kernel
idx = [i for i, e in enumerate(ev) if e[2].startswith(step_kernel)]
lo, hi = idx[len(idx) // 3], idx[-1]          # skip the warm-up third
sel = ev[lo:hi]
nsteps = sum(1 for e in sel if e[2].startswith(step_kernel))
busy = 0
gaps = collections.defaultdict(lambda: [0.0, 0])
end = sel[0][0]
for s, e, name in sel:
    if s > end:
        g = (s - end) / 1e3
        if g >= min_gap:
            k = (prev, name)
            gaps[k][0] += g
            gaps[k][1] += 1
        busy += e - s
    else:
        busy += max(0, e - max(s, end))
    if e > end:
        end, prev = e, name
span = (sel[-1][1] - sel[0][0]) / 1e6
print(f"{nsteps} steps, {span / nsteps:.3f} ms per step, busy {busy / 1e6 / nsteps:.3f} ms per step, "
      f"idle {(span - busy / 1e6) / nsteps:.3f} ms per step")
print(f"gaps of at least {min_gap} us, per step:")
for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"  {t / nsteps:8.1f} us  x{c / nsteps:5.1f}   {a[:44]:44s} -> {b[:44]}")
