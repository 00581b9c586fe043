#!/usr/bin/env python3
"""s_waitcnt / scratch traffic inside the innermost MFMA loop of every kernel of a HIP translation unit whose name matches.
usage: tools/kwaits.py pairwise.hip [name-filter] [extra hipcc flags...]   (cross-compiles on CPU)"""
import re, subprocess, sys
src, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
extra = sys.argv[3:]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I/root/repo/include",
                "-I/root/repo/tidypopgen_amd/csrc", "--cuda-device-only", "-S", "/root/repo/tidypopgen_amd/csrc/" + src, "-o", "/tmp/kw.s"] + extra,
               check=True, stderr=subprocess.DEVNULL)
lines = open("/tmp/kw.s").read().split("\n")
starts = [(i, l) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for k, (i, name) in enumerate(starts):
    j = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
    body = lines[i:j]
    mf = [x for x, l in enumerate(body) if "v_mfma" in l]
    if not mf or filt not in name:
        continue
    hdr = max(x for x in range(mf[0]) if re.match(r"^\.LBB\d+_\d+:", body[x]))
    end = next(x for x in range(mf[-1], len(body)) if "s_cbranch" in body[x])
    loop = body[hdr:end + 1]
    w = [re.sub(r"\s+", " ", l.strip())[:40] for l in loop if "s_waitcnt" in l or "scratch_" in l]
    nm = sum("v_mfma" in l for l in loop)
    nv = sum(bool(re.match(r"\s*v_(?!mfma)", l)) for l in loop)
    nl = sum("global_load" in l for l in loop)
    print(f"{name.split(':')[0][:70]}\n   loop: {nm} mfma, {nv} valu, {nl} loads; waits: {w}")
