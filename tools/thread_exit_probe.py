#!/usr/bin/env python3
"""Where do 35 - 48 ms go after a helper thread's large device -> host copies?  (bench.py's end-to-end routes: the thread's last
statement returns, the join comes back that much later.)  Variants: the same copies from the main thread; a helper thread that
keeps its result arrays vs one that drops them; a helper thread that does nothing."""
import sys
import threading
import time
sys.path.insert(0, ".")
import ctypes as C
import numpy as np
import tidypopgen_amd as tpg
from tidypopgen_amd import api

lib, chk = tpg._lib.lib, tpg._lib.check
ctx = tpg.default_context()
ctx2 = tpg.Context(0)
nbytes = 816_000_000
d = ctx.dev_alloc(nbytes)
keep = {}


def download(c, tag, fresh=True):
    a = np.empty(nbytes, dtype=np.uint8) if fresh else keep.setdefault("reuse", np.zeros(nbytes, dtype=np.uint8))
    t = time.perf_counter()
    chk(lib.tpg_dev_to_host(c.h, api._ptr(a), d, C.c_size_t(nbytes)))
    dt = time.perf_counter() - t
    keep[tag] = a
    return dt


def in_thread(fn):
    out = {}
    def body():
        out["r"] = fn()
        out["t_end"] = time.perf_counter()
    th = threading.Thread(target=body)
    t0 = time.perf_counter()
    th.start(); th.join()
    t1 = time.perf_counter()
    return out["r"], (t1 - out["t_end"]) * 1e3, (t1 - t0) * 1e3


download(ctx, "warm")
for rep in range(2):
    t = time.perf_counter(); dt = download(ctx, f"main{rep}"); tot = time.perf_counter() - t
    print(f"main thread, fresh array:    copy {dt*1e3:6.1f} ms, call {tot*1e3:6.1f} ms")
    r, gap, tot = in_thread(lambda: download(ctx2, f"thr{rep}"))
    print(f"helper thread, fresh array:  copy {r*1e3:6.1f} ms, thread end -> join {gap:6.1f} ms, total {tot:6.1f} ms")
    r, gap, tot = in_thread(lambda: download(ctx2, "reuse", fresh=False))
    print(f"helper thread, reused array: copy {r*1e3:6.1f} ms, thread end -> join {gap:6.1f} ms, total {tot:6.1f} ms")
    r, gap, tot = in_thread(lambda: 0.0)
    print(f"helper thread, nothing:      thread end -> join {gap:6.1f} ms")
    t = time.perf_counter(); a = np.empty(nbytes, dtype=np.uint8); a[::4096] = 1; print(f"numpy touch of a fresh array: {(time.perf_counter()-t)*1e3:.1f} ms")
    t = time.perf_counter(); del a; print(f"freeing it: {(time.perf_counter()-t)*1e3:.1f} ms")
