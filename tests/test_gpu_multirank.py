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


@pytest.mark.timeout(900)
@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_two_shards_equal_one(tmp_path, scaling):
    d1, d2 = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    common = ["--steps", "1", "--warmup", "0", "--indiv", "700", "--pops", "9", "--k", "8", "--no-cpu-baseline",
              "--no-end-to-end", "--scaling", scaling]
    _run([sys.executable, "bench.py", "--gpus", "1", "--snps", "60000", "--digest", d1] + common, {})
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
          "127.0.0.1", "--master-port", "29533" if scaling == "strong" else "29535", "bench.py", "--gpus", "2", "--snps",
          "60000" if scaling == "strong" else "30000", "--digest", d2] + common, {"TPG_BENCH_SHARE_GPU": "1"})
    a, b = json.load(open(d1)), json.load(open(d2))
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
    assert np.allclose(a["pca_d"], b["pca_d"], rtol=1e-7)  # Gram partials use per-shard weight scaling
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
        sh = tpg.ShardedPairwise(comm, n)
        sh.accumulate(v)
        sh.reduce()
        assert sh.band() == sharding.band_rows(n, W, r)
        got = sh.epilogues(("ibs", "king", "allele_sharing"), m=m)
        mask = sharding.band_mask(n, W, r)
        cover += mask
        for k in want:
            assert np.array_equal(got[k][mask], want[k][mask], equal_nan=True), (r, k)
            assert np.isnan(got[k][~mask]).all(), (r, k)  # nothing written outside the band
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
