#!/usr/bin/env python3
"""Run one probe script under every library build in abvar/ (tools/build_variants.sh), round after round, inside one GPU job:
tools/lib_ab.py <rounds> <probe.py> [probe args...]   -- prints the probe's last output line per build."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds, probe, args = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
names = sorted(f[7:-3] for f in os.listdir(os.path.join(ROOT, "abvar")) if f.startswith("libtpg_") and f.endswith(".so"))
for r in range(rounds):
    for nm in names:
        env = dict(os.environ, TPG_LIB_PATH=os.path.join(ROOT, "abvar", f"libtpg_{nm}.so"))
        out = subprocess.run([sys.executable, probe] + args, cwd=ROOT, env=env, capture_output=True, text=True)
        last = (out.stdout.strip().splitlines() or ["(no output)"])[-1] if out.returncode == 0 else "FAILED " + out.stderr[-300:]
        print(f"round {r} {nm:12s} {last}", flush=True)
