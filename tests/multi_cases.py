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
    """host-only translation unit (no device code): g++ against the HIP runtime API"""
    if not os.path.exists(MOCK_SO) or os.path.getmtime(MOCK_SO) < os.path.getmtime(MOCK_SRC):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__",
                               "-I/opt/rocm/include", MOCK_SRC, "-o", MOCK_SO, "-L/opt/rocm/lib", "-lamdhip64",
                               "-Wl,-rpath,/opt/rocm/lib"])
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
    transport = mg.transport()
    mg.close()
    return transport


def mock_stats():
    """collectives this process completed over the mock: {all, all-reduce, reduce-scatter, all-to-all}"""
    lib = C.CDLL(MOCK_SO)
    out = (C.c_uint64 * 4)()
    lib.mock_rccl_stats(out)
    return dict(zip(("all", "allreduce", "reducescatter", "alltoallv"), [int(x) for x in out]))


if __name__ == "__main__":
    ndev = int(sys.argv[1])
    assert os.environ.get("TPG_RCCL_LIBRARY") == MOCK_SO and os.environ.get("TPG_MULTI_FORCE_RCCL") == "1"
    tr = multi_against_oracle(ndev, devices=[0] * ndev)
    st = mock_stats()
    print("MOCK_OK", tr, st)
    assert "mock_rccl" in tr, tr
    # per rank: two pairwise calls = two reduce-scatters; the forced class exchange = at least one real all-to-all + its self-test
    assert st["reducescatter"] >= 2 * ndev and st["alltoallv"] >= 2 * ndev and st["allreduce"] >= 10 * ndev, st
