#!/usr/bin/env python3
"""Wall-clock per phase of the bench step (sync after each phase) next to the kernel time inside it."""
import sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, ".")
import bench

args = bench.parse()
st = bench.Step(args, 0, 1, 0)
tpg, api, lib, ctx = st.tpg, st.api, st.lib, st.ctx
chk = tpg._lib.check
n, m, G, P = args.n, args.m, args.pops, st.P
st.run(); ctx.sync()
ctx.prof_enable(True)

def phase(name, fn):
    ctx.sync(); ctx.prof_reset(); t0 = time.perf_counter(); r = fn(); ctx.sync(); t1 = time.perf_counter()
    kms = sum(ms for _, ms in ctx.prof_dump().values())
    print(f"{name:22s} wall {1e3*(t1-t0):8.3f} ms   kernels {kms:8.3f} ms   gap {1e3*(t1-t0)-kms:7.3f}")
    return r

for rep in range(2):
    print("rep", rep)
    v = phase("view raw", lambda: api.View(st.X, None, None, code256=st.code_012))
    phase("alt_freq", lambda: chk(lib.tpg_alt_freq_dip_pseudo(ctx.h, v.h, api._ptr(st.ploidy), C.c_int(0), st.d_freq)))
    phase("grouped_alt_freq", lambda: chk(lib.tpg_grouped_alt_freq_dip_pseudo(ctx.h, v.h, api._ptr(st.gid), C.c_int(G), api._ptr(st.ploidy), C.c_int(0), st.d_gfreq)))
    for name, code in (("Hudson", 0), ("WC84", 2)):
        sn, sd = np.zeros(P), np.zeros(P)
        phase("fst " + name, lambda: chk(lib.tpg_pairwise_pop_fst_sums(ctx.h, v.h, api._ptr(st.gid), C.c_int(G), api._ptr(st.ploidy), C.c_int(code), api._ptr(st.pairs), C.c_int(P), api._ptr(sn), api._ptr(sd))))
    phase("pairwise zero+acc", lambda: (st.pw.zero(), st.pw.accumulate(v)))
    phase("epilogues", lambda: chk(lib.tpg_pairwise_epilogues(ctx.h, st.pw.h, C.c_int(0), C.c_int64(m), st.d_nn[0], st.d_nn[1], C.c_void_p(None), st.d_nn[2])))
    phase("view free", lambda: v.free())
    vi = phase("view imputed", lambda: api.View(st.X, None, st.pca_cols, code256=st.code_imp))
    phase("pca_partial_svd", lambda: chk(lib.tpg_pca_partial_svd(ctx.h, vi.h, C.c_int(args.k), api._ptr(st.pca_d), st.d_pca["u"], st.d_pca["v"], st.d_pca["center"], st.d_pca["scale"], C.byref(st.pca_fro))))
    vi.free()
    ctx.prof_reset()
    ctx.sync(); t0 = time.perf_counter(); st.run(); ctx.sync(); t1 = time.perf_counter()
    print(f"whole step wall {1e3*(t1-t0):.3f} ms, kernels {sum(ms for _, ms in ctx.prof_dump().values()):.3f}")
