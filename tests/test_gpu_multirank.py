"""GPU: the sharded (N > 1) paths give the same answers as the unsharded ones on the same panel.

RCCL refuses two ranks on one device, and the test box has one GPU, so:
  * the two-process test runs bench.py's step with two ranks that share the GPU and exchange through the library's
    host-callback transport over gloo (tpg_comm_init_host) -- everything except the RCCL calls themselves is the code
    the RCCL run executes: shard ranges, slabs laid out in bands, reduce-scatter semantics, band epilogues, the GRM
    mean over ranks, Fst sums, the Gram all-reduce inside tpg_pca_partial_svd_sharded;
  * the one-process tests drive tpg_multi_* (ncclCommInitAll is skipped for one device) and the sharded entry points
    on one rank, which must equal the unsharded entry points bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env_extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **env_extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return r


_GLOO_DIGEST_CACHE = {}


@pytest.mark.timeout(900)
@pytest.mark.parametrize("scaling,exchange", [("strong", False), ("weak", False), ("strong", True)])
def test_two_shards_equal_one(tmp_path, scaling, exchange):
    """exchange: the PCA Gram with whole weight classes per rank (the packed columns go round by an all-to-all,
    gramcls.hip) instead of every rank's own loci -- forced here, the cost model keeps it for long panels"""
    d1, d2 = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    common = ["--steps", "1", "--warmup", "0", "--indiv", "700", "--pops", "9", "--k", "8", "--no-cpu-baseline",
              "--no-end-to-end", "--no-standalone", "--scaling", scaling]
    if "one" not in _GLOO_DIGEST_CACHE:  # (the 1-GPU digest of the 60 000-locus panel is the same for the three cases)
        _run([sys.executable, "bench.py", "--gpus", "1", "--snps", "60000", "--digest", d1] + common, {})
        _GLOO_DIGEST_CACHE["one"] = json.load(open(d1))
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
          "127.0.0.1", "--master-port", "29537" if exchange else ("29533" if scaling == "strong" else "29535"), "bench.py", "--gpus", "2", "--snps",
          "60000" if scaling == "strong" else "30000", "--digest", d2] + common,
         {"TPG_BENCH_SHARE_GPU": "1", **({"TPG_GRAM_EXCHANGE": "1"} if exchange else {})})
    a, b = _GLOO_DIGEST_CACHE["one"], json.load(open(d2))
    # integer cross-products are exact, so every epilogue value is identical (the bands of the two ranks tile the
    # matrices: the digest sums what each rank wrote); the GRM mean is summed in another order
    for name in ("ibs", "king", "grm"):
        assert a[name + "_nan"] == b[name + "_nan"]
        tol = 1e-12 if name == "grm" else 0  # GRM = 2 (M - mb) / (1 - mb): values near 0 carry the rounding of mb
        assert np.allclose(a[name + "_corner"], b[name + "_corner"], rtol=tol, atol=tol, equal_nan=True), name
        assert np.allclose(a[name + "_last"], b[name + "_last"], rtol=tol, atol=tol, equal_nan=True), name
        assert a[name + "_sum"] == pytest.approx(b[name + "_sum"], rel=1e-12)
    for name in ("fst_hudson", "fst_wc84"):
        assert np.allclose(a[name], b[name], rtol=1e-12, atol=0)
    # the digit Gram of a shard rounds its weights per shard (1e-7); exchanged classes are exact integer matrices
    assert np.allclose(a["pca_d"], b["pca_d"], rtol=1e-9 if exchange else 1e-7)
    assert a["pca_fro"] == pytest.approx(b["pca_fro"], rel=1e-12)
    assert np.allclose(a["pca_u_abs_colsum"], b["pca_u_abs_colsum"], rtol=1e-5)


def test_one_rank_sharded_entry_points_equal_unsharded():
    import tidypopgen_amd as tpg

    n, m = 333, 5000
    X = tpg.FBM.synth(13, n, m, npop=5, miss=0.04, imputed_bytes=True)
    v = tpg.View(X, code256=None)
    comm = tpg.Comm.init_rank(X.ctx, 1, 0, None)
    assert comm.shard_loci(m) == (0, m)
    ref = tpg.Pairwise(X.ctx, n)
    ref.accumulate(v)
    sh = tpg.ShardedPairwise(comm, n)
    sh.accumulate(v)
    sh.reduce()
    assert sh.band() == (0, n)
    with pytest.raises(tpg._lib.TpgError):
        sh.accumulate(v)  # reduced accumulators must be zeroed first
    e_ref, e_sh = ref.epilogues(m=m), sh.epilogues(m=m)
    for k in e_ref:
        assert np.array_equal(e_ref[k], e_sh[k], equal_nan=True), k
    c_ref, c_sh = ref.counts(), sh.counts()
    for k in c_ref:
        assert np.array_equal(c_ref[k], c_sh[k]), k
    sh.zero()
    sh.accumulate(v)
    assert np.array_equal(sh.counts(("ibs",))["ibs"], c_ref["ibs"])
    arr = np.arange(7.0)
    assert np.array_equal(comm.allreduce_f64(arr.copy()), arr)


def test_multi_one_device_against_oracle():
    """tpg_multi_* with one device: the single-process path an R session takes (upload of the device's share of
    colInd, pack, accumulate, reduce, band epilogue written straight into the host matrices)."""
    import tidypopgen_amd as tpg
    from oracle import oracle as orc

    n, m = 150, 3000
    fbm = orc.synth_fbm(17, n, m, npop=4, miss=0.05)
    mg = tpg.Multi(1)
    out = mg.pairwise(fbm)
    assert np.array_equal(out["ibs"], orc.snp_ibs(fbm), equal_nan=True)
    assert np.array_equal(out["king"], orc.snp_king(fbm), equal_nan=True)
    as_ = orc.snp_allele_sharing(fbm)
    assert np.array_equal(out["allele_sharing"], as_, equal_nan=True)
    assert np.allclose(out["grm"], orc.pairwise_grm(as_), rtol=1e-12, atol=1e-14)
    rows = (np.random.default_rng(1).permutation(n)[:77] + 1).astype(np.int32)
    cols = (np.random.default_rng(2).permutation(m)[:999] + 1).astype(np.int32)
    sub = mg.pairwise(fbm, rows, cols, which=("ibs", "king"), ibs_type="adjusted_counts")
    assert np.array_equal(sub["ibs"], orc.snp_ibs(fbm, rows, cols, type="adjusted_counts"), equal_nan=True)
    assert np.array_equal(sub["king"], orc.snp_king(fbm, rows, cols), equal_nan=True)
    with pytest.raises(tpg._lib.TpgError):
        mg.pairwise(fbm, None, np.array([m + 1], dtype=np.int32))
    # only the cross-products the requested matrices need: {V, D} for the GRM alone, {V, D, A} for KING + GRM, {V, D, H} for IBS
    assert np.array_equal(mg.pairwise(fbm, which=("grm",))["grm"], out["grm"], equal_nan=True)
    kg = mg.pairwise(fbm, which=("king", "grm"))
    assert np.array_equal(kg["king"], out["king"], equal_nan=True) and np.array_equal(kg["grm"], out["grm"], equal_nan=True)
    assert np.array_equal(mg.pairwise(fbm, which=("ibs", "allele_sharing"))["ibs"], out["ibs"], equal_nan=True)
    mg.close()


def test_band_layout_of_many_ranks_on_one_gpu():
    """The slab layout of an 8-rank run (bands padded to equal chunks) with the exchange emulated on one GPU: eight
    host-transport communicators whose callback adds nothing (every "rank" sees only its own partials).  Rank r
    accumulating ALL loci and finishing band r must reproduce band r of the unsharded result, and the eight bands
    must tile the matrix."""
    import tidypopgen_amd as tpg
    from tidypopgen_amd import sharding

    n, m, W = 700, 4096, 8
    X = tpg.FBM.synth(19, n, m, npop=6, miss=0.03)
    v = tpg.View(X, code256=None)
    ref = tpg.Pairwise(X.ctx, n)
    ref.accumulate(v)
    want = ref.epilogues(("ibs", "king", "allele_sharing"), m=m)
    cover = np.zeros((n, n), dtype=int)
    for r in range(W):
        comm = tpg.Comm.host(X.ctx, W, r, lambda a: a)  # identity "all-reduce": this rank's partials ARE the totals
        mask = sharding.band_mask(n, W, r)
        cover += mask
        # the five-product kernel on odd ranks; on even ranks the product-set kernels, which fill the same band-padded slabs
        runs = [(None, ("ibs", "king", "allele_sharing"))] if r % 2 else [(tpg.PW_FOR_KING, ("king", "allele_sharing")),
                                                                         (tpg.PW_FOR_IBS, ("ibs",)), (tpg.PW_FOR_AS, ("allele_sharing",))]
        for products, which in runs:
            sh = tpg.ShardedPairwise(comm, n)
            sh.accumulate(v, products=products)
            sh.reduce()
            assert sh.band() == sharding.band_rows(n, W, r)
            got = sh.epilogues(which, m=m)
            for k in which:
                assert np.array_equal(got[k][mask], want[k][mask], equal_nan=True), (r, k)
                assert np.isnan(got[k][~mask]).all(), (r, k)  # nothing written outside the band
            sh.free()
        comm.close()
    assert cover.min() == 1 and cover.max() == 1


def test_rccl_calls_on_a_one_rank_communicator(monkeypatch):
    """What a one-GPU box can rehearse of the RCCL transport: librccl is loaded at run time (dlopen), a real
    communicator is created (TPG_COMM_FORCE_RCCL=1 makes tpg_comm_init_rank do so for one rank too), and the
    reduce-scatter of the int32 slabs, the N x N FP64 all-reduce inside the sharded PCA, the GRM mean and the host-staged
    all-reduce of Fst sums all go through ncclReduceScatter / ncclAllReduce on the context's stream.  With one rank the
    sums are the inputs, so the results must equal the unsharded entry points bit for bit."""
    import tidypopgen_amd as tpg

    monkeypatch.setenv("TPG_COMM_FORCE_RCCL", "1")
    n, m, k = 300, 6000, 6
    X = tpg.FBM.synth(23, n, m, npop=5, miss=0.0, imputed_bytes=True)
    comm = tpg.Comm.init_rank(X.ctx, 1, 0, None)
    v = tpg.View(X, code256=None)
    ref = tpg.Pairwise(X.ctx, n)
    ref.accumulate(v)
    sh = tpg.ShardedPairwise(comm, n)
    sh.accumulate(v)
    sh.reduce()                                    # ncclReduceScatter, in place
    e_ref, e_sh = ref.epilogues(m=m), sh.epilogues(m=m)   # GRM mean: ncclAllReduce of two doubles
    for key in e_ref:
        assert np.array_equal(e_ref[key], e_sh[key], equal_nan=True), key
    arr = np.linspace(0, 1, 1000)
    assert np.array_equal(comm.allreduce_f64(arr.copy()), arr)      # host array staged through the device
    import ctypes as C
    from tidypopgen_amd import api

    vi = tpg.View(X, code256=tpg.CODE_IMPUTE_PRED)
    exact = tpg.gt_pca_partialSVD(X, k=k)
    d, u = np.zeros(k), np.zeros((n, k), order="F")
    vl, ce, sc = np.zeros((m, k), order="F"), np.zeros(m), np.zeros(m)
    fro = C.c_double()
    tpg._lib.check(tpg._lib.lib.tpg_pca_partial_svd_sharded(X.ctx.h, comm.h, vi.h, C.c_int(k), api._ptr(d), api._ptr(u),
                                                            api._ptr(vl), api._ptr(ce), api._ptr(sc), C.byref(fro)))
    assert np.array_equal(d, exact["d"]) and np.array_equal(u, exact["u"]) and fro.value == exact["square_frobenius"]
    comm.close()


def _align_sign(a, b):
    s = np.sign((a * b).sum(axis=0))
    s[s == 0] = 1
    return a * s


@pytest.mark.parametrize("stream_budget", [None, 1 << 20])
@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_multi_fst_freq_pca_against_oracle(ndev, stream_budget, monkeypatch):
    """tpg_multi_pop_fst / tpg_multi_grouped_alt_freq / tpg_multi_pca_partial_svd: what ONE R process calls for
    BASELINE configs 4 and 5.  ndev = 1 is the plain single-device path; ndev = 2, 3 list device 0 several times, so the
    device threads exchange through the in-process host transport (RCCL refuses one GPU twice): thread teams, phases,
    status agreement and the row-range writes into the caller's matrices are the code an 8-GPU run executes.
    stream_budget: TPG_STREAM_BUDGET makes a device's share "too large to hold", so that the same entry points take the
    streamed form (csrc/stream.hip: the share swept in blocks, two block buffers per device) -- same results asked."""
    import tidypopgen_amd as tpg
    from oracle import oracle as orc

    if stream_budget:
        monkeypatch.setenv("TPG_STREAM_BUDGET", str(stream_budget))
    n, m, G, k = 180, 2900, 5, 6
    fbm = orc.synth_fbm(29, n, m, npop=G, miss=0.04, imputed_bytes=True)
    gid = (np.arange(n) % G).astype(np.int32)
    mg = tpg.Multi(ndev, devices=[0] * ndev)
    rows = (np.random.default_rng(3).permutation(n)[:150] + 1).astype(np.int32)
    cols = (np.sort(np.random.default_rng(4).permutation(m)[:2500]) + 1).astype(np.int32)
    for r, c in ((None, None), (rows, cols)):
        g = gid if r is None else gid[r - 1]
        # grouped and ungrouped allele frequencies: bit exact
        assert np.array_equal(mg.loci_alt_freq(fbm, r, c, g, G),
                              orc.grouped_alt_freq_dip_pseudo_cpp(fbm, r, c, g, G, np.full(len(g), 2.0)))
        assert np.array_equal(mg.loci_alt_freq(fbm, r, c, as_counts=True), orc.loci_alt_freq(fbm, r, c, as_counts=True))
        for method in ("Hudson", "WC84", "Nei87"):
            o = orc.pairwise_pop_fst(fbm, r, c, g, G, method=method, by_locus=True)
            t = mg.pairwise_pop_fst(fbm, r, c, g, G, method=method, by_locus=True)
            assert np.array_equal(t["fst_locus"], o["fst_locus"], equal_nan=True), method  # per locus: bit identical
            assert np.allclose(t["fst_tot"], o["fst_tot"], rtol=1e-12, atol=0), method
            t2 = mg.pairwise_pop_fst(fbm, r, c, g, G, method=method)
            assert np.allclose(t2["fst_tot"], o["fst_tot"], rtol=1e-12, atol=0), method
        o = orc.pairwise_pop_fst(fbm, r, c, g, G, method="Hudson", return_num_dem=True)
        t = mg.pairwise_pop_fst(fbm, r, c, g, G, method="Hudson", return_num_dem=True)
        for key in ("Fst_by_locus_num", "Fst_by_locus_den"):
            assert np.array_equal(t[key], o[key], equal_nan=True), key
    # PCA on the polymorphic loci (big_SVD stops on a zero scale)
    dec = np.where(fbm > 3, fbm - 4, fbm)
    pc = (np.where((dec.sum(axis=0) > 0) & (dec.sum(axis=0) < 2 * n))[0] + 1).astype(np.int32)
    o = orc.gt_pca_partialSVD(fbm, None, pc, k=k)
    t = mg.gt_pca_partialSVD(fbm, None, pc, k=k)
    assert np.array_equal(t["center"], o["center"]) and np.array_equal(t["scale"], o["scale"])
    assert t["square_frobenius"] == pytest.approx(o["square_frobenius"], rel=1e-12)
    assert np.allclose(t["d"], o["d"], rtol=1e-6, atol=0)
    so, st = o["u"] * o["d"], _align_sign(t["u"] * t["d"], o["u"] * o["d"])
    assert np.max(np.abs(st - so)) <= 1e-6 * np.max(np.abs(so))
    assert np.max(np.abs(_align_sign(t["v"], o["v"]) - o["v"])) <= 1e-6 * np.max(np.abs(o["v"]))
    # a shard with a zero scale fails on ONE device thread only: every device gives up together (no thread is left in
    # the Gram all-reduce), and the error is big_SVD's
    mono = fbm.copy(order="F")
    mono[:, m - 5] = 0
    with pytest.raises(tpg._lib.TpgError) as e:
        mg.gt_pca_partialSVD(mono, None, None, k=k)
    assert e.value.code == 4
    # a panel too short for every device to hold k loci runs on one device
    short = mg.gt_pca_partialSVD(fbm, None, pc[:100], k=k)
    assert np.allclose(short["d"], orc.gt_pca_partialSVD(fbm, None, pc[:100], k=k)["d"], rtol=1e-6)
    mg.close()


@pytest.mark.timeout(900)
def test_bench_gpus_n_launches_n_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it starts two ranks itself (torch.distributed.run children) and
    reports n_gpus = 2 and the transport; asking for more GPUs than there are is refused instead of reported as N."""
    common = ["--steps", "1", "--warmup", "0", "--indiv", "500", "--snps", "40000", "--pops", "7", "--k", "6",
              "--no-cpu-baseline", "--no-end-to-end"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + common, cwd=ROOT, capture_output=True, text=True,
                       timeout=600, env=dict(env, TPG_BENCH_SHARE_GPU="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and "transport" in out["config"]["collectives"]
    import tidypopgen_amd as tpg

    have = tpg.device_count()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(have + 1)] + common, cwd=ROOT, capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode != 0 and "refusing" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("ndev", [2, 8])
def test_pca_with_whole_classes_per_rank(ndev, monkeypatch):
    """tpg_gram_classes_exchanged among the device threads of one process (in-process host transport, one GPU listed ndev
    times): key histogram all-reduce, equal-cost key ranges, records packed by destination, the all-to-all, the class Gram
    of the records received, double centering per rank, the triangle all-reduce.  Against one device and the oracle; with 8
    ranks some own only a few classes (and on a tiny panel none)."""
    import tidypopgen_amd as tpg
    from oracle import oracle as orc

    monkeypatch.setenv("TPG_GRAM_EXCHANGE", "1")
    n, m, k = 230, 9000, 7
    fbm = orc.synth_fbm(31, n, m, npop=4, miss=0.03, imputed_bytes=True)
    dec = np.where(fbm > 3, fbm - 4, fbm)
    pc = (np.where((dec.sum(axis=0) > 0) & (dec.sum(axis=0) < 2 * n))[0] + 1).astype(np.int32)
    one = tpg.gt_pca_partialSVD(tpg.FBM.from_numpy(fbm), None, pc, k=k)
    o = orc.gt_pca_partialSVD(fbm, None, pc, k=k)
    mg = tpg.Multi(ndev, devices=[0] * ndev)
    t = mg.gt_pca_partialSVD(fbm, None, pc, k=k)
    assert np.array_equal(t["center"], one["center"]) and np.array_equal(t["scale"], one["scale"])
    assert np.allclose(t["d"], one["d"], rtol=1e-10, atol=0)  # exact class matrices, double weights: only the sum order differs
    assert np.allclose(t["d"], o["d"], rtol=1e-6, atol=0)
    assert np.max(np.abs(_align_sign(t["u"], one["u"]) - one["u"])) <= 1e-8
    assert np.max(np.abs(_align_sign(t["v"], one["v"]) - one["v"])) <= 1e-8 * np.abs(one["v"]).max()
    assert t["square_frobenius"] == pytest.approx(one["square_frobenius"], rel=1e-12)
    # a short panel: fewer classes than ranks can own, some ranks receive nothing
    few = pc[:max(8 * ndev, 2 * k * ndev)]
    t2 = mg.gt_pca_partialSVD(fbm, None, few, k=k)
    o2 = orc.gt_pca_partialSVD(fbm, None, few, k=k)
    assert np.allclose(t2["d"], o2["d"], rtol=1e-6, atol=0)
    mg.close()


# ---------------------------------------------------------------------------------------------------------------------
# Real multi-GPU hardware.  Everything above rehearses the sharded paths on ONE GPU (gloo / in-process transports).  The
# tests below run the same comparisons over RCCL on DISTINCT devices -- the first real ncclReduceScatter with N > 1, the
# first real ncclAllToAllv, ncclCommInitAll on several devices -- and are skipped on a box with fewer GPUs than they need,
# so they cost nothing today and verify the transport the first time 2 or 8 GPUs are there (axis being sharded:
# R/snp_ibs.R:59-82).
def _ngpu():
    import torch  # counting devices does not initialise HIP

    return torch.cuda.device_count()


def _need_gpus(k):
    return pytest.mark.skipif(_ngpu() < k, reason=f"needs {k} GPUs, this box has {_ngpu()}")


def _digests_match(a, b, exchange):
    for name in ("ibs", "king", "grm"):
        assert a[name + "_nan"] == b[name + "_nan"]
        tol = 1e-12 if name == "grm" else 0
        assert np.allclose(a[name + "_corner"], b[name + "_corner"], rtol=tol, atol=tol, equal_nan=True), name
        assert np.allclose(a[name + "_last"], b[name + "_last"], rtol=tol, atol=tol, equal_nan=True), name
        assert a[name + "_sum"] == pytest.approx(b[name + "_sum"], rel=1e-12)
    for name in ("fst_hudson", "fst_wc84"):
        assert np.allclose(a[name], b[name], rtol=1e-12, atol=0)
    assert np.allclose(a["pca_d"], b["pca_d"], rtol=1e-9 if exchange else 1e-7)
    assert a["pca_fro"] == pytest.approx(b["pca_fro"], rel=1e-12)
    assert np.allclose(a["pca_u_abs_colsum"], b["pca_u_abs_colsum"], rtol=1e-5)


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("ngpu,exchange", [pytest.param(2, False, marks=_need_gpus(2)), pytest.param(2, True, marks=_need_gpus(2)),
                                           pytest.param(8, False, marks=_need_gpus(8)), pytest.param(8, True, marks=_need_gpus(8))])
def test_rccl_ranks_on_distinct_gpus_equal_one_gpu(tmp_path, ngpu, exchange):
    """`bench.py --gpus N --digest` over RCCL (no TPG_BENCH_SHARE_GPU: one process per GPU, ncclCommInitRank, the real
    reduce-scatter / all-reduces; exchange: the first real ncclAllToAllv) must reproduce the 1-GPU digest with the
    tolerances of test_two_shards_equal_one, and the line must say the collectives ran over rccl."""
    d1, dn = str(tmp_path / "one.json"), str(tmp_path / "n.json")
    common = ["--steps", "1", "--warmup", "0", "--indiv", "700", "--pops", "9", "--k", "8", "--no-cpu-baseline",
              "--no-end-to-end", "--no-standalone", "--snps", "120000"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TPG_BENCH_SHARE_GPU")}
    _run([sys.executable, "bench.py", "--gpus", "1", "--digest", d1] + common, {})
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(ngpu), "--digest", dn] + common, cwd=ROOT, capture_output=True,
                       text=True, timeout=900, env=dict(env, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0",
                                                        **({"TPG_GRAM_EXCHANGE": "1"} if exchange else {})))
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == ngpu
    assert "transport rccl" in line["config"]["collectives"], line["config"]["collectives"]
    if exchange:
        assert "all-to-all" in line["config"]["pca_gram_path"], line["config"]["pca_gram_path"]
    _digests_match(json.load(open(d1)), json.load(open(dn)), exchange)


@pytest.mark.parametrize("stream_budget", [None, 1 << 20])
@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_multi_and_multi_stream_device_threads_against_oracle(ndev, stream_budget, monkeypatch):
    """tests/multi_cases.py::multi_against_oracle -- tpg_multi_* and the streamed form, tpg_multi_stream_run (every device
    sweeps its share of colInd in blocks under a budget, then the exchanges) -- with device 0 listed ndev times: the device
    threads exchange through the in-process transport.  The same body runs over the mock RCCL (below) and, on a box that has
    them, on distinct GPUs (next test)."""
    from tests import multi_cases

    if stream_budget:  # tpg_multi_pairwise / _grouped_alt_freq / _pop_fst / _pca_partial_svd themselves take the streamed form
        monkeypatch.setenv("TPG_STREAM_BUDGET", str(stream_budget))
    tr = multi_cases.multi_against_oracle(ndev, devices=[0] * ndev, setenv=monkeypatch.setenv)
    assert tr == ("none" if ndev == 1 else "host callback"), tr


@pytest.mark.parametrize("ndev", [pytest.param(2, marks=_need_gpus(2)), pytest.param(8, marks=_need_gpus(8))])
def test_multi_on_distinct_gpus_against_oracle(ndev, monkeypatch):
    """tpg.Multi(ndev) on devices 0 .. ndev-1 (ncclCommInitAll, one host thread per device, RCCL between them) against the
    oracle, as test_multi_fst_freq_pca_against_oracle does with one device listed several times; once more with the class
    exchange forced (ncclAllToAllv inside tpg_multi_pca_partial_svd)."""
    from tests import multi_cases

    tr = multi_cases.multi_against_oracle(ndev, setenv=monkeypatch.setenv)
    assert tr.startswith("rccl: librccl"), tr


# ---------------------------------------------------------------------------------------------------------------------
# The nccl* call sites with N > 1 ranks on ONE GPU: tests/host/mock_rccl.cpp stands in for librccl.so (TPG_RCCL_LIBRARY), with
# rccl.h's semantics -- counts and displacements in elements, the in-place reduce-scatter at sendbuff + rank * recvcount, send
# and receive counts of the all-to-all checked against each other, every buffer range checked against its allocation, every
# rank checked to be in the SAME collective.  What the skipped tests above would be the first to run on hardware runs here
# today; what the mock cannot show is in its header (stream ordering, RCCL's own kernels, xGMI).
_MOCK_DIGEST_CACHE = {}


@pytest.fixture(scope="module")
def mock_rccl():
    from tests import multi_cases

    return multi_cases.build_mock_rccl()


def test_mock_rccl_semantics_and_misuse_detection(mock_rccl):
    """The mock itself, driven directly (two rank threads, device buffers from the library's allocator): sums, the in-place
    reduce-scatter, an all-to-all with ragged counts -- and the mistakes it exists to catch at comm.hip's call sites:
    displacements given in bytes, a receive count that differs from the peer's send count, a reduce-scatter whose recvbuff
    overlaps sendbuff anywhere else than at rank * recvcount, a rank in another collective than its peers."""
    import ctypes as C
    import threading

    import tidypopgen_amd as tpg
    from tidypopgen_amd import api

    lib = C.CDLL(mock_rccl)
    lib.ncclGetErrorString.restype = C.c_char_p
    ctx = tpg.Context(0)
    R, W = 2, 1000
    tlib = tpg._lib.lib

    def dalloc(nbytes):
        p = C.c_void_p()
        tpg._lib.check(tlib.tpg_dev_alloc(ctx.h, C.c_size_t(nbytes), C.byref(p)))
        return p

    def put(p, a):
        tpg._lib.check(tlib.tpg_dev_from_host(ctx.h, p, api._ptr(a), C.c_size_t(a.nbytes)))
        tpg._lib.check(tlib.tpg_ctx_sync(ctx.h))

    def get(p, a):
        tpg._lib.check(tlib.tpg_dev_to_host(ctx.h, api._ptr(a), p, C.c_size_t(a.nbytes)))
        return a

    def two_ranks(body):
        cs = (C.c_void_p * R)()
        assert lib.ncclCommInitAll(cs, C.c_int(R), (C.c_int * R)(0, 0)) == 0
        out, err = [None] * R, []

        def run(r):
            try:
                out[r] = body(r, C.c_void_p(cs[r]))
            except Exception as e:  # noqa: BLE001
                err.append(repr(e))

        th = [threading.Thread(target=run, args=(r,)) for r in range(R)]
        [t.start() for t in th]
        [t.join() for t in th]
        for c in cs:
            lib.ncclCommDestroy(C.c_void_p(c))
        assert not err, err
        return out

    lock = threading.Lock()  # the context's staging helpers are used by one host thread at a time

    def good(r, comm):
        rng = np.random.default_rng(100 + r)
        mine = rng.integers(-1000, 1000, R * W).astype(np.int32)
        x = rng.standard_normal(77)
        # ragged: rank r sends B + 3 + r + 2 d words to d (B large enough that displacements in BYTES leave the allocation,
        # which the library's pool rounds up to whole MiB)
        B = 70_000
        scnt = np.array([B + 3 + r + 2 * dd for dd in range(R)], dtype=np.uint64)
        soff = np.concatenate([[0], np.cumsum(scnt)[:-1]]).astype(np.uint64)
        rcnt = np.array([B + 3 + s_ + 2 * r for s_ in range(R)], dtype=np.uint64)
        roff = np.concatenate([[0], np.cumsum(rcnt)[:-1]]).astype(np.uint64)
        send = np.arange(int(scnt.sum()), dtype=np.uint64) + np.uint64(10_000_000 * r)
        with lock:
            d, dx, ds_, dr_ = dalloc(mine.nbytes), dalloc(x.nbytes), dalloc(send.nbytes), dalloc(8 * int(rcnt.sum()))
            put(d, mine); put(dx, x); put(ds_, send)
        rc = lib.ncclReduceScatter(d, C.c_void_p(d.value + 4 * W * r), C.c_size_t(W), C.c_int(2), C.c_int(0), comm, None)
        assert rc == 0, lib.ncclGetErrorString(rc)
        assert lib.ncclAllReduce(dx, dx, C.c_size_t(77), C.c_int(8), C.c_int(0), comm, None) == 0
        rc = lib.ncclAllToAllv(ds_, api._ptr(scnt), api._ptr(soff), dr_, api._ptr(rcnt), api._ptr(roff), C.c_int(5), comm, None)
        assert rc == 0, lib.ncclGetErrorString(rc)
        with lock:
            res = dict(mine=mine, after=get(d, np.zeros(R * W, dtype=np.int32)), x=x, xsum=get(dx, np.zeros(77)), send=send, scnt=scnt,
                       soff=soff, got=get(dr_, np.zeros(int(rcnt.sum()), dtype=np.uint64)), rcnt=rcnt, roff=roff)
        # misuse 1: displacements in bytes (every rank makes the same mistake) -> the pieces leave the allocation
        res["bytes_rc"] = lib.ncclAllToAllv(ds_, api._ptr(scnt), api._ptr(soff * np.uint64(8)), dr_, api._ptr(rcnt),
                                            api._ptr(roff * np.uint64(8)), C.c_int(5), comm, None)
        return res

    res = two_ranks(good)
    tot = sum(res[r]["mine"].astype(np.int64) for r in range(R))
    for r in range(R):
        a = res[r]
        assert np.array_equal(a["after"][W * r:W * (r + 1)], tot[W * r:W * (r + 1)])      # my chunk: the sum
        other = np.ones(R * W, dtype=bool)
        other[W * r:W * (r + 1)] = False
        assert np.array_equal(a["after"][other], a["mine"][other])                          # the rest: untouched
        assert np.array_equal(a["xsum"], res[0]["x"] + res[1]["x"])
        for s_ in range(R):
            b = res[s_]
            piece = b["send"][int(b["soff"][r]):int(b["soff"][r]) + int(b["scnt"][r])]
            assert np.array_equal(a["got"][int(a["roff"][s_]):int(a["roff"][s_]) + int(a["rcnt"][s_])], piece)
        assert a["bytes_rc"] != 0

    def mismatch(r, comm):  # misuse 2: rank 1 expects one word more from rank 0 than rank 0 sends
        scnt, soff = np.array([4, 4], dtype=np.uint64), np.array([0, 4], dtype=np.uint64)
        rcnt, roff = np.array([4 + (r == 1), 4], dtype=np.uint64), np.array([0, 5], dtype=np.uint64)
        with lock:
            ds_, dr_ = dalloc(8 * 8), dalloc(8 * 10)
        return lib.ncclAllToAllv(ds_, api._ptr(scnt), api._ptr(soff), dr_, api._ptr(rcnt), api._ptr(roff), C.c_int(5), comm, None)

    assert all(rc != 0 for rc in two_ranks(mismatch))

    def overlap(r, comm):  # misuse 3: in place at the WRONG chunk (rank r writes chunk 1 - r)
        with lock:
            d = dalloc(4 * R * W)
        return lib.ncclReduceScatter(d, C.c_void_p(d.value + 4 * W * (1 - r)), C.c_size_t(W), C.c_int(2), C.c_int(0), comm, None)

    assert all(rc != 0 for rc in two_ranks(overlap))

    def different_ops(r, comm):  # a rank in another collective than its peers: on hardware, the classic hang
        with lock:
            d = dalloc(4 * R * W)
        if r == 0:
            return lib.ncclAllReduce(d, d, C.c_size_t(W), C.c_int(2), C.c_int(0), comm, None)
        return lib.ncclReduceScatter(d, C.c_void_p(d.value + 4 * W), C.c_size_t(W), C.c_int(2), C.c_int(0), comm, None)

    assert all(rc != 0 for rc in two_ranks(different_ops))
    ctx.close()


# stream-ordered, adversarial mock (MOCK_RCCL_ASYNC=1): a collective only ENQUEUES; until a delayed combine has run its receive
# buffer holds poison, so whatever consumes a collective's output without being ordered behind it on the stream fails parity
_MOCK_ASYNC = dict(MOCK_RCCL_ASYNC="1", MOCK_RCCL_DELAY_MS="3", MOCK_RCCL_KERNEL_TIMEOUT_S="30")


def test_mock_rccl_stream_ordered_mode_shows_an_unordered_reader_poison(mock_rccl):
    """tests/multi_cases.py::async_mock_selftest in a child process (the mode is fixed per process)"""
    r = subprocess.run([sys.executable, "-m", "tests.multi_cases", "selftest"], cwd=ROOT, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MOCK_RCCL_ASYNC="1", MOCK_RCCL_DELAY_MS="50"))
    assert r.returncode == 0 and "SELFTEST_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


# (rank THREADS stay on the synchronous mock: with all ranks on ONE device and in one process, any runtime call that waits for
# the device -- a hipMalloc of the library's pool, a pinned allocation -- waits for the OTHER rank's wait kernel, which waits
# for this rank: measured, the first collective sits out its 30 s and hands back poison.  Ranks that are processes have a
# device context each, and so do threads on distinct GPUs.)
@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["sync"])
@pytest.mark.parametrize("ndev", [2, 8])
def test_mock_rccl_multi_against_oracle(mock_rccl, ndev, mode):
    """tpg.Multi with device 0 listed ndev times and TPG_MULTI_FORCE_RCCL=1: ncclCommInitAll, then ndev rank threads through
    ncclReduceScatter (in place at rank offsets, band-padded slabs), the FP64 / int32 ncclAllReduce calls, and -- with the class
    exchange forced -- ncclAllToAllv with the library's element counts and displacements; results against the oracle.  A child
    process, because the library loads its RCCL once per process."""
    env = dict(os.environ, TPG_RCCL_LIBRARY=mock_rccl, TPG_MULTI_FORCE_RCCL="1", **(_MOCK_ASYNC if mode == "async" else {}))
    r = subprocess.run([sys.executable, "-m", "tests.multi_cases", str(ndev)], cwd=ROOT, capture_output=True, text=True, timeout=800, env=env)
    assert r.returncode == 0 and "MOCK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert ("ASYNC_OK" in r.stdout) == (mode == "async"), r.stdout[-2000:]


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("mode", ["sync", "async", "async+overlap"])
@pytest.mark.parametrize("nproc,exchange,scaling", [(2, False, "strong"), (2, True, "strong"), (4, True, "strong")])  # (weak scaling: test_two_shards_equal_one)
def test_mock_rccl_rank_processes_equal_one_gpu(tmp_path, mock_rccl, nproc, exchange, scaling, mode):
    """`bench.py --gpus N --digest` with one PROCESS per rank, all on device 0, over the mock (TPG_BENCH_SHARE_GPU=rccl):
    tpg_comm_unique_id -> broadcast -> ncclCommInitRank, the reduce-scatter of the pairwise slabs, the Fst / Gram / GRM-mean
    all-reduces, tpg_comm_agree's status words and (exchange) ncclAllToAllv -- the digest must equal the 1-GPU digest with the
    tolerances of the gloo rehearsal, and the line must say which library the collectives ran over."""
    d1, dn = str(tmp_path / "one.json"), str(tmp_path / "n.json")
    snps = 60000
    common = ["--steps", "1", "--warmup", "0", "--indiv", "700", "--pops", "9", "--k", "8", "--no-cpu-baseline",
              "--no-end-to-end", "--no-standalone", "--scaling", scaling]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if "one" not in _MOCK_DIGEST_CACHE:  # the 1-GPU digest of this panel does not depend on the case (one rank: weak = strong)
        _run([sys.executable, "bench.py", "--gpus", "1", "--digest", d1, "--snps", str(snps)] + common, {})
        _MOCK_DIGEST_CACHE["one"] = json.load(open(d1))
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(nproc), "--digest", dn, "--snps",
                        str(snps if scaling == "strong" else snps // nproc)] + common, cwd=ROOT, capture_output=True,
                       text=True, timeout=900, env=dict(env, MASTER_ADDR="127.0.0.1", TPG_BENCH_SHARE_GPU="rccl", TPG_RCCL_LIBRARY=mock_rccl,
                                                        **({"TPG_GRAM_EXCHANGE": "1"} if exchange else {}),
                                                        **(_MOCK_ASYNC if mode.startswith("async") else {}),
                                                        # the reduce-scatter of the pair counts on a second communicator / stream
                                                        # beside the PCA (tpg_pairwise_reduce_begin / _end), under the adversarial mock
                                                        **({"TPG_OVERLAP_REDUCE": "1"} if mode.endswith("overlap") else {})))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "[mock_rccl]" not in r.stderr, r.stderr[-3000:]  # (a collective that failed behind its call says so there)
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == nproc
    assert "libmock_rccl.so" in line["config"]["collectives"] and "ONE GPU" in line["config"]["collectives"], line["config"]["collectives"]
    if exchange:
        assert "all-to-all" in line["config"]["pca_gram_path"], line["config"]["pca_gram_path"]
    assert ("second communicator" in line["config"]["collectives"]) == mode.endswith("overlap"), line["config"]["collectives"]
    _digests_match(_MOCK_DIGEST_CACHE["one"], json.load(open(dn)), exchange)
