#!/usr/bin/env python3
"""What a plain device-to-device copy gets from HBM on this GPU (bytes read + bytes written per second): the yardstick for the
kernels that read and write HBM in equal parts (two-view pack, locus-major copy, gather)."""
import torch
for gb in (1.25, 5.0):
    n = int(gb * 1e9)
    a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    a.fill_(1); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"copy of {gb} GB: {best:.3f} ms = {2 * n / best / 1e9:.2f} TB/s (read + write)")
    best = 1e9
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); s = a.view(torch.int64).sum(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"read of {gb} GB (sum): {best:.3f} ms = {n / best / 1e9:.2f} TB/s")
    best = 1e9
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); b.fill_(3); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"write of {gb} GB (fill): {best:.3f} ms = {n / best / 1e9:.2f} TB/s")
    del a, b
