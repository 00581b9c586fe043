"""Shared bodies of the multi-GPU tests (tests/test_gpu_multirank.py): run in the pytest process on distinct GPUs, or as a child
process (`python -m tests.multi_cases <ndev>`) whose environment makes the library load tests/host/mock_rccl.cpp instead of
librccl.so, so that ncclCommInitAll and every nccl* call site run with ndev ranks on ONE GPU."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK_SRC = os.path.join(ROOT, "tests", "host", "mock_rccl.cpp")
MOCK_SO = os.path.join(ROOT, "tests", "host", "libmock_rccl.so")


def build_mock_rccl() -> str:
    """one translation unit, host code + the one-thread wait kernel of its stream-ordered mode: hipcc for gfx950"""
    if not os.path.exists(MOCK_SO) or os.path.getmtime(MOCK_SO) < os.path.getmtime(MOCK_SRC):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-x", "hip",
                               "--offload-arch=gfx950", MOCK_SRC, "-o", MOCK_SO, "-lpthread"])
    return MOCK_SO


def align_sign(a, b):
    s = np.sign((a * b).sum(axis=0))
    s[s == 0] = 1
    return a * s


def multi_against_oracle(ndev, devices=None, setenv=None):
    """tpg.Multi(ndev) against the oracle: pairwise (five products and a product set behind one reduce-scatter), grouped
    frequencies, Fst, the PCA with every rank's own loci and with whole weight classes per rank (all-to-all)."""
    import tidypopgen_amd as tpg
    from oracle import oracle as orc

    n, m, G, k = 260, 12000, 5, 6
    fbm = orc.synth_fbm(37, n, m, npop=G, miss=0.04, imputed_bytes=True)
    gid = (np.arange(n) % G).astype(np.int32)
    mg = tpg.Multi(ndev, devices=devices)
    out = mg.pairwise(fbm)
    assert np.array_equal(out["ibs"], orc.snp_ibs(fbm), equal_nan=True)
    assert np.array_equal(out["king"], orc.snp_king(fbm), equal_nan=True)
    as_ = orc.snp_allele_sharing(fbm)
    assert np.array_equal(out["allele_sharing"], as_, equal_nan=True)
    assert np.allclose(out["grm"], orc.pairwise_grm(as_), rtol=1e-12, atol=1e-14)
    ibs_alone = mg.pairwise(fbm, which=("ibs",))  # V and D + H in one sum (TPG_PW_DH) on every device: the "lacks a product" words of the reduction carry it
    assert np.array_equal(ibs_alone["ibs"], out["ibs"], equal_nan=True)
    only = mg.pairwise(fbm, which=("king", "grm"))  # the {V, D, A} kernel on every device, one reduce-scatter
    assert np.array_equal(only["king"], out["king"], equal_nan=True) and np.array_equal(only["grm"], out["grm"], equal_nan=True)
    assert np.array_equal(mg.loci_alt_freq(fbm, None, None, gid, G),
                          orc.grouped_alt_freq_dip_pseudo_cpp(fbm, None, None, gid, G, np.full(n, 2.0)))
    for method in ("Hudson", "WC84"):
        o = orc.pairwise_pop_fst(fbm, None, None, gid, G, method=method, by_locus=True)
        t = mg.pairwise_pop_fst(fbm, None, None, gid, G, method=method, by_locus=True)
        assert np.array_equal(t["fst_locus"], o["fst_locus"], equal_nan=True), method
        assert np.allclose(t["fst_tot"], o["fst_tot"], rtol=1e-12, atol=0), method
    dec = np.where(fbm > 3, fbm - 4, fbm)
    pc = (np.where((dec.sum(axis=0) > 0) & (dec.sum(axis=0) < 2 * n))[0] + 1).astype(np.int32)
    o = orc.gt_pca_partialSVD(fbm, None, pc, k=k)
    for exchange in (False, True):
        if exchange:
            (setenv or os.environ.__setitem__)("TPG_GRAM_EXCHANGE", "1")
        t = mg.gt_pca_partialSVD(fbm, None, pc, k=k)
        assert np.array_equal(t["center"], o["center"]) and np.array_equal(t["scale"], o["scale"])
        assert np.allclose(t["d"], o["d"], rtol=1e-6, atol=0)
        so = o["u"] * o["d"]
        assert np.max(np.abs(align_sign(t["u"] * t["d"], so) - so)) <= 1e-6 * np.max(np.abs(so))
    # the streamed form (tpg_multi_stream_run): the store stays on the host, every device sweeps ITS share of colInd in blocks
    # under a budget (two block buffers, an uploader and a downloader thread per device), then the same exchanges -- one
    # reduce-scatter of the pairwise slabs, all-reduces of the Fst sums, of the Gram matrix and of its Frobenius norm
    for budget in (0, 1 << 20):
        S = tpg.Stream.from_numpy(fbm, budget_bytes=budget)
        s = S.run(None, pc, pairwise=("ibs", "king", "allele_sharing", "grm"), groupIds=gid, ngroups=G, alt_freq=True,
                  grouped_alt_freq=True, loci_counts=True, fst=("Hudson", "WC84"), fst_by_locus=True, k=k, multi=mg)
        sub = np.asfortranarray(fbm[:, pc - 1])
        assert np.array_equal(s["ibs"], orc.snp_ibs(sub), equal_nan=True)
        assert np.array_equal(s["king"], orc.snp_king(sub), equal_nan=True)
        as_s = orc.snp_allele_sharing(sub)
        assert np.array_equal(s["allele_sharing"], as_s, equal_nan=True)
        assert np.allclose(s["grm"], orc.pairwise_grm(as_s), rtol=1e-12, atol=1e-14)
        assert np.array_equal(s["alt_freq"], orc.alt_freq_dip_pseudo_cpp(sub, None, None, np.full(n, 2.0)), equal_nan=True)
        assert np.array_equal(s["grouped_alt_freq"], orc.grouped_alt_freq_dip_pseudo_cpp(sub, None, None, gid, G, np.full(n, 2.0)))
        assert np.array_equal(s["loci_counts"][:, 3], (sub > 2).sum(axis=0))
        for method in ("Hudson", "WC84"):
            of = orc.pairwise_pop_fst(sub, None, None, gid, G, method=method, by_locus=True)
            assert np.array_equal(s["fst_locus"][method], of["fst_locus"], equal_nan=True), method
            assert np.allclose(s["fst_tot"][method], of["fst_tot"], rtol=1e-12, atol=0), method
        assert np.array_equal(s["center"], o["center"]) and np.array_equal(s["scale"], o["scale"])
        assert np.allclose(s["d"], o["d"], rtol=1e-8, atol=0)
        so = o["u"] * o["d"]
        assert np.max(np.abs(align_sign(s["u"] * s["d"], so) - so)) <= 1e-6 * np.max(np.abs(so))
        assert np.max(np.abs(align_sign(s["v"], o["v"]) - o["v"])) <= 1e-6 * np.max(np.abs(o["v"]))
        assert abs(s["square_frobenius"] / o["square_frobenius"] - 1) < 1e-12
        if budget:
            assert s["report"]["blocks"] > 1
        S.close()
    # ... and from a PLINK .bed payload: a store whose run is paced by its kernels takes the Gram ONCE over the kept views
    # laid end to end (StreamRun::batch_gram) -- here on every device over its own share, then the same all-reduce
    import shutil
    import tempfile

    dec = np.where(fbm > 3, fbm - 4, fbm).astype(np.uint8)  # the imputed genotypes as genotypes: nothing missing
    tmp = tempfile.mkdtemp(prefix="tpg_multi_bed_")
    try:
        code = np.array([3, 2, 0, 1], dtype=np.uint8)[dec]  # PLINK: 00 hom A1 (dosage 2), 01 missing, 10 het, 11 hom A2
        code = np.vstack([code, np.zeros(((-n) % 4, m), dtype=np.uint8)])
        q = code.T.reshape(m, -1, 4)
        with open(os.path.join(tmp, "s.bed"), "wb") as f:
            f.write(bytes([0x6C, 0x1B, 0x01]))
            f.write((q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8).tobytes())
        S = tpg.Stream.open_bed(os.path.join(tmp, "s.bed"), n, m, budget_bytes=0)
        s = S.run(None, pc, pairwise=("ibs", "king"), k=k, multi=mg)
        sub = np.asfortranarray(dec[:, pc - 1])
        assert s["report"]["blocks"] > 2 and s["report"]["views_kept"]
        assert np.array_equal(s["ibs"], orc.snp_ibs(sub), equal_nan=True)
        assert np.array_equal(s["king"], orc.snp_king(sub), equal_nan=True)
        assert np.array_equal(s["center"], o["center"]) and np.array_equal(s["scale"], o["scale"])
        assert np.allclose(s["d"], o["d"], rtol=1e-8, atol=0)
        so = o["u"] * o["d"]
        assert np.max(np.abs(align_sign(s["u"] * s["d"], so) - so)) <= 1e-6 * np.max(np.abs(so))
        assert np.max(np.abs(align_sign(s["v"], o["v"]) - o["v"])) <= 1e-6 * np.max(np.abs(o["v"]))
        S.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    transport = mg.transport()
    mg.close()
    return transport


def mock_stats():
    """collectives this process completed over the mock: {all, all-reduce, reduce-scatter, all-to-all}"""
    lib = C.CDLL(MOCK_SO)
    out = (C.c_uint64 * 4)()
    lib.mock_rccl_stats(out)
    return dict(zip(("all", "allreduce", "reducescatter", "alltoallv"), [int(x) for x in out]))


def async_mock_selftest():
    """MOCK_RCCL_ASYNC=1, one rank, driven directly: the call returns before the collective has run; a reader that is NOT ordered
    behind it (a kernel on another stream: the library's mailbox fetch through a second context) sees the poison pattern in the
    receive buffer; a reader behind it on the stream sees the sum.  This is the property the stream-ordered mode exists for,
    shown on the mock itself."""
    import time

    import tidypopgen_amd as tpg
    from tidypopgen_amd import api

    hip = C.CDLL("libamdhip64.so")
    lib = C.CDLL(MOCK_SO)
    tlib = tpg._lib.lib

    def chk(rc, what):
        assert rc == 0, (what, rc)

    n = 1 << 12  # 16 KiB: what the contexts read back goes through their mailbox kernel, not a copy engine
    ctx_a, ctx_b = tpg.Context(0), tpg.Context(0)
    sa, sb = C.c_void_p(), C.c_void_p()
    chk(hip.hipStreamCreateWithFlags(C.byref(sa), C.c_uint(1)), "stream")  # non-blocking streams: no implicit ordering between them
    chk(hip.hipStreamCreateWithFlags(C.byref(sb), C.c_uint(1)), "stream")
    ctx_a.set_stream(sa.value)
    ctx_b.set_stream(sb.value)
    d_send, d_recv = ctx_a.dev_alloc(4 * n), ctx_a.dev_alloc(4 * n)
    src = np.arange(n, dtype=np.int32)
    zero = np.zeros(n, dtype=np.int32)
    tpg._lib.check(tlib.tpg_dev_from_host(ctx_a.h, d_send, api._ptr(src), C.c_size_t(4 * n)))
    tpg._lib.check(tlib.tpg_dev_from_host(ctx_a.h, d_recv, api._ptr(zero), C.c_size_t(4 * n)))
    warm = np.zeros(n, dtype=np.int32)  # (the contexts' mailboxes are set up by their first use)
    tpg._lib.check(tlib.tpg_dev_to_host(ctx_b.h, api._ptr(warm), d_recv, C.c_size_t(4 * n)))
    tpg._lib.check(tlib.tpg_dev_to_host(ctx_a.h, api._ptr(warm), d_recv, C.c_size_t(4 * n)))
    cs = (C.c_void_p * 1)()
    chk(lib.ncclCommInitAll(cs, C.c_int(1), (C.c_int * 1)(0)), "init")
    comm = C.c_void_p(cs[0])
    t0 = time.perf_counter()
    chk(lib.ncclAllReduce(d_send, d_recv, C.c_size_t(n), C.c_int(2), C.c_int(0), comm, sa), "allreduce")
    t_call = time.perf_counter() - t0
    early, late = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    tpg._lib.check(tlib.tpg_dev_to_host(ctx_b.h, api._ptr(early), d_recv, C.c_size_t(4 * n)))  # the other stream
    t_early = time.perf_counter() - t0
    tpg._lib.check(tlib.tpg_dev_to_host(ctx_a.h, api._ptr(late), d_recv, C.c_size_t(4 * n)))   # behind the collective
    t_late = time.perf_counter() - t0
    print(f"call {t_call * 1e3:.2f} ms, other-stream read done at {t_early * 1e3:.2f} ms, ordered read at {t_late * 1e3:.2f} ms", flush=True)
    poison = np.uint32(0xA5A5A5A5).astype(np.int32)
    delay = int(os.environ.get("MOCK_RCCL_DELAY_MS", "3")) * 1e-3
    assert t_call < 0.5 * delay, t_call                  # the call did not wait for the (delayed) combine
    assert t_early < 0.8 * delay and t_late >= delay, (t_early, t_late)
    assert np.all(early == poison), early[:4]            # not ordered behind the collective: poison
    assert np.array_equal(late, src), late[:4]           # ordered behind it on its stream: the sum over the one rank
    chk(lib.ncclCommDestroy(comm), "destroy")
    a = (C.c_uint64 * 4)()
    lib.mock_rccl_async_stats(a)
    assert list(a) == [1, 1, 0, 0], list(a)
    print("SELFTEST_OK", f"call returned after {t_call * 1e3:.2f} ms")


if __name__ == "__main__":
    if sys.argv[1] == "selftest":
        async_mock_selftest()
        sys.exit(0)
    ndev = int(sys.argv[1])
    assert os.environ.get("TPG_RCCL_LIBRARY") == MOCK_SO and os.environ.get("TPG_MULTI_FORCE_RCCL") == "1"
    tr = multi_against_oracle(ndev, devices=[0] * ndev)
    st = mock_stats()
    print("MOCK_OK", tr, st)
    assert "mock_rccl" in tr, tr
    # per rank: two pairwise calls = two reduce-scatters; the forced class exchange = at least one real all-to-all + its self-test
    assert st["reducescatter"] >= 2 * ndev and st["alltoallv"] >= 2 * ndev and st["allreduce"] >= 10 * ndev, st
    a = (C.c_uint64 * 4)()
    C.CDLL(MOCK_SO).mock_rccl_async_stats(a)
    if os.environ.get("MOCK_RCCL_ASYNC") == "1":  # every collective completed BEHIND its call, none failed there, no wait kernel gave up
        assert a[0] == 1 and a[1] >= st["all"] and a[2] == 0 and a[3] == 0, list(a)
        print("ASYNC_OK", list(a))
    else:
        assert a[0] == 0 and a[1] == 0, list(a)
