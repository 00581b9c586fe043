#!/usr/bin/env python3
"""Which FP4 magnitude each operand plane of the pairwise kernels carries (TPG_T4_ENC, devfrag.h) changes no sum -- the block
scales undo it -- only the bit patterns the matrix cores multiply, and with them the clock the chip holds under the MFMAs.
Every encoding is a build of the library (tools/build_variants.sh encK "-DTPG_T4_ENC=K"); this script runs tools/pw_only.py
for each of them as a child process, round after round (the boxes drift), inside one GPU job.   tools/enc_ab.py [rounds]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
names = sorted(f[7:-3] for f in os.listdir(os.path.join(ROOT, "abvar")) if f.startswith("libtpg_") and f.endswith(".so"))
res = {}
for r in range(rounds):
    for nm in names:
        env = dict(os.environ, TPG_LIB_PATH=os.path.join(ROOT, "abvar", f"libtpg_{nm}.so"))
        out = subprocess.run([sys.executable, "tools/pw_only.py", "5000", "1000000", "0"], cwd=ROOT, env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print(nm, "FAILED", out.stderr[-500:], flush=True)
            continue
        for line in out.stdout.splitlines():
            mm = re.match(r"(\w+)\s+variant 0:\s+([\d.]+) ms", line)
            if mm:
                res.setdefault((nm, mm.group(1)), []).append(float(mm.group(2)))
        print(f"round {r} {nm}: " + "  ".join(f"{k[1]} {v[-1]:.3f}" for k, v in res.items() if k[0] == nm), flush=True)
print()
for nm in names:
    print(nm, "  ".join(f"{s} {min(res[(nm, s)]):.3f}" for s in ("all", "as", "ibs", "king") if (nm, s) in res))
