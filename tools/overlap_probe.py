#!/usr/bin/env python3
"""Does a device -> host copy on a second stream (second host thread, second context) slow the kernels of the first, and is it
slowed by them?  The pairwise pass (18 ms of MFMA kernel) alone, a download of 0.83 GB alone, and both at once."""
import sys
import threading
import time
sys.path.insert(0, ".")
import ctypes as C
import numpy as np
import tidypopgen_amd as tpg
from tidypopgen_amd import api

lib, chk = tpg._lib.lib, tpg._lib.check
n, m = 5000, 1000000
ctx = tpg.default_context()
X = tpg.FBM.synth(3, n, m, npop=51, imputed_bytes=True)
v = tpg.View(X, code256=None)
pw = tpg.Pairwise(ctx, n)
ctx2 = tpg.Context(0)
nbytes = 832_000_000
d = ctx.dev_alloc(nbytes)
host = np.empty(nbytes, dtype=np.uint8)
host[:] = 0  # pages present


def kernels(reps=2):
    t = time.perf_counter()
    for _ in range(reps):
        pw.zero(); pw.accumulate(v)
    ctx.sync()
    return time.perf_counter() - t


def download():
    t = time.perf_counter()
    chk(lib.tpg_dev_to_host(ctx2.h, api._ptr(host), d, C.c_size_t(nbytes)))
    return time.perf_counter() - t


kernels(1); download()
for rep in range(3):
    tk, td = kernels(), download()
    res = {}
    th = threading.Thread(target=lambda: res.__setitem__("d", download()))
    t0 = time.perf_counter()
    th.start()
    tk2 = kernels()
    th.join()
    both = time.perf_counter() - t0
    print(f"kernels alone {tk*1e3:.1f} ms, download alone {td*1e3:.1f} ms ({nbytes/td/1e9:.1f} GB/s); together: kernels {tk2*1e3:.1f}, "
          f"download {res['d']*1e3:.1f} ({nbytes/res['d']/1e9:.1f} GB/s), wall {both*1e3:.1f}", flush=True)
