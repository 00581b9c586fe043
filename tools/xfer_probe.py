#!/usr/bin/env python3
"""Raw host <-> HBM rates on this box (torch pinned / pageable copies) beside the library's bulk paths."""
import sys, time, os
import numpy as np
import torch
sys.path.insert(0, ".")
import tidypopgen_amd as tpg

n, m = 5000, int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
nbytes = n * m
dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
pin = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
page = torch.empty(nbytes, dtype=torch.uint8)
page.fill_(1); pin.fill_(1)
for name, src in (("pinned", pin), ("pageable", page)):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); dev.copy_(src, non_blocking=False); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"torch H2D {name}: {nbytes/dt/1e9:.1f} GB/s")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); src.copy_(dev); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"torch D2H {name}: {nbytes/dt/1e9:.1f} GB/s")
ctx = tpg.default_context()
a = np.asfortranarray(page.numpy().reshape(m, n).T)
for rep in range(2):
    t0 = time.perf_counter(); X = tpg.FBM.from_numpy(a); ctx.sync(); dt = time.perf_counter() - t0
    print(f"tpg from_numpy: {nbytes/dt/1e9:.1f} GB/s")
    del X
for base in ("/dev/shm", os.environ.get("TMPDIR", "/tmp")):
    path = os.path.join(base, f"xfer_probe_{os.getpid()}.bk")
    try:
        a.T.tofile(path)
    except OSError as e:
        print(base, "skipped", e); continue
    for rep in range(2):
        t0 = time.perf_counter(); X = tpg.FBM.open_bk(path, n, m); ctx.sync(); dt = time.perf_counter() - t0
        print(f"tpg open_bk {base}: {nbytes/dt/1e9:.1f} GB/s")
        del X
    # plain threaded pread into pageable memory: what the file system itself delivers
    import concurrent.futures as cf
    buf = np.empty(nbytes, dtype=np.uint8)
    def rd(k, T=16):
        lo, hi = nbytes * k // T, nbytes * (k + 1) // T
        fd = os.open(path, os.O_RDONLY)
        got = os.preadv(fd, [memoryview(buf)[lo:hi]], lo)
        os.close(fd)
        return got
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(16) as ex:
        list(ex.map(rd, range(16)))
    dt = time.perf_counter() - t0
    print(f"16-thread pread {base} -> pageable: {nbytes/dt/1e9:.1f} GB/s")
    os.remove(path)
print("cpus", len(os.sched_getaffinity(0)))
