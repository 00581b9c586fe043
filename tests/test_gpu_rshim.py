"""GPU: shim/tpg_rshim.c driven the way R drives it -- through tests/rmock (NOT R: a minimal stand-in for the R C API,
see its header) -- on FBM objects whose fields the shim reads with Rf_eval, with the block loops of the reference's R
drivers restated here (R/snp_ibs.R:59-82, R/snp_king.R:51-77, R/snp_allele_sharing.R:49-70).  What is pinned: an
UNMODIFIED driver gets correct matrices by default; the deferred mode is opt-in; the HBM copy of the genotype FBM is
dropped when the backing file changes in place (R/gt_impute_simple.R:86); the whole-analysis entry points."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests import rmock

pytestmark = pytest.mark.gpu


R = rmock.Session


def _double_fbm(tmp_path, name, n, fill=0.0):
    """bigstatsr::FBM(n, n, init = fill): a backing file of n x n doubles"""
    p = tmp_path / name
    np.full(n * n, fill).tofile(p)
    return p


def _read_double_fbm(p, n):
    return np.fromfile(p, dtype=np.float64).reshape((n, n), order="F")


@pytest.fixture()
def shim(tmp_path, monkeypatch):
    monkeypatch.delenv("TPG_RSHIM_DEFERRED", raising=False)
    lib = rmock.build(tmp_path)
    yield R(lib)
    lib.R_unload_tpgshim(None)
    lib.rmock_reset()


def _driver_loop(r, which, BM, K, K2, rows, cols, block):
    lo, up = orc.cut_by_size(len(cols), block)
    rmock.driver_loop(r, which, BM, K, K2, rows, cols, lo, up)


def _oracle_counts(which, fbm, rows, cols):
    n = len(rows)
    A, B = np.zeros((n, n), order="F"), np.zeros((n, n), order="F")
    {"ibs": orc.increment_ibs_counts, "king": orc.increment_king_numerator, "as": orc.increment_as_counts}[which](A, B, fbm, rows, cols)
    return A, B


def test_unmodified_drivers_are_correct_by_default(shim, tmp_path):
    n_all, m_all = 90, 2600
    fbm = orc.synth_fbm(41, n_all, m_all, npop=3, miss=0.05)
    bk = tmp_path / "geno.bk"
    fbm.T.tofile(bk)  # column-major bytes
    BM = shim.fbm(bk, n_all, m_all, orc.CODE_012)
    rows = (np.random.default_rng(5).permutation(n_all)[:64] + 1).astype(np.int32)
    cols = np.arange(1, m_all + 1, dtype=np.int32)
    n = len(rows)
    for which in ("ibs", "king", "as"):
        kp, k2p = _double_fbm(tmp_path, which + "_k.bk", n), _double_fbm(tmp_path, which + "_k2.bk", n)
        K, K2 = shim.fbm(kp, n, n), shim.fbm(k2p, n, n)
        _driver_loop(shim, which, BM, K, K2, rows, cols, 700)
        # the driver reads its FBMs right after the loop (R/snp_ibs.R:84-95): no flush was called
        A, B = _oracle_counts(which, fbm, rows, cols)
        assert np.array_equal(_read_double_fbm(kp, n), A), which
        assert np.array_equal(_read_double_fbm(k2p, n), B), which
    # a scattered colInd (every 7th locus) goes through the gathered upload
    sub = cols[::7].copy()
    kp, k2p = _double_fbm(tmp_path, "s_k.bk", n), _double_fbm(tmp_path, "s_k2.bk", n)
    _driver_loop(shim, "ibs", BM, shim.fbm(kp, n, n), shim.fbm(k2p, n, n), rows, sub, 100)
    A, B = _oracle_counts("ibs", fbm, rows, sub)
    assert np.array_equal(_read_double_fbm(kp, n), A) and np.array_equal(_read_double_fbm(k2p, n), B)
    # the accumulators of an analysis stay mapped while its block loop runs and go when the next analysis starts (or at
    # tpg_release / unload): a session never holds more than one analysis' pair of temp files
    maps = open("/proc/self/maps").read()
    assert "ibs_k.bk" not in maps and "king_k2.bk" not in maps and "as_k.bk" not in maps
    shim.call("tpg_release")
    assert "s_k2.bk" not in open("/proc/self/maps").read()


def test_deferred_mode_is_opt_in(shim, tmp_path, monkeypatch):
    monkeypatch.setenv("TPG_RSHIM_DEFERRED", "1")  # read when the shim creates its context
    n_all, m_all = 50, 1500
    fbm = orc.synth_fbm(43, n_all, m_all, npop=2, miss=0.03)
    bk = tmp_path / "geno.bk"
    fbm.T.tofile(bk)
    BM = shim.fbm(bk, n_all, m_all, orc.CODE_012)
    rows = np.arange(1, n_all + 1, dtype=np.int32)
    cols = np.arange(1, m_all + 1, dtype=np.int32)
    kp, k2p = _double_fbm(tmp_path, "k.bk", n_all, 2.0), _double_fbm(tmp_path, "k2.bk", n_all, 0.0)
    _driver_loop(shim, "king", BM, shim.fbm(kp, n_all, n_all), shim.fbm(k2p, n_all, n_all), rows, cols, 400)
    assert np.all(_read_double_fbm(kp, n_all) == 2.0)  # nothing yet
    shim.call("tpg_flush")
    A, B = _oracle_counts("king", fbm, rows, cols)
    assert np.array_equal(_read_double_fbm(kp, n_all), A + 2.0) and np.array_equal(_read_double_fbm(k2p, n_all), B)
    assert "k2.bk" not in open("/proc/self/maps").read()  # unmapped at the flush


@pytest.mark.parametrize("cache", [False, True])
def test_backing_file_rewritten_in_place_is_reuploaded(shim, tmp_path, monkeypatch, cache):
    """gt_impute_simple rewrites the .bk in place (R/gt_impute_simple.R:86); the next call must see the new bytes.  Default:
    nothing of the FBM outlives a call.  TPG_RSHIM_CACHE=1 (opt-in): an HBM copy guarded by size + mtime + page fingerprint,
    and tpg_invalidate(BM) for edits the heuristic could miss."""
    if cache:
        monkeypatch.setenv("TPG_RSHIM_CACHE", "1")
    else:
        monkeypatch.delenv("TPG_RSHIM_CACHE", raising=False)
    n, m = 120, 4000
    fbm = orc.synth_fbm(47, n, m, npop=4, miss=0.06)
    bk = tmp_path / "geno.bk"
    fbm.T.tofile(bk)
    BM = shim.fbm(bk, n, m, orc.CODE_IMPUTE_PRED)
    rows, cols = np.arange(1, n + 1, dtype=np.int32), np.arange(1, m + 1, dtype=np.int32)
    ploidy = np.full(n, 2.0)

    def alt_freq():
        out = shim.call("alt_freq_dip_pseudo_cpp", BM, shim.int(rows), shim.int(cols), shim.real(ploidy), shim.int([1]),
                        shim.lib.rmock_lgl(1))
        return shim.as_numpy(out, (m, 2))

    before = alt_freq()
    assert np.array_equal(before, orc.alt_freq_dip_pseudo_cpp(fbm, rows, cols, ploidy, True, orc.CODE_IMPUTE_PRED))
    assert np.array_equal(alt_freq(), before)  # second call (cache: served from the HBM copy)
    imputed = np.where(fbm == 3, np.uint8(4 + 1), fbm)  # every missing genotype imputed as heterozygous (bytes 5)
    mm = np.memmap(bk, dtype=np.uint8, mode="r+")
    st0 = os.stat(bk)
    mm[:] = imputed.T.reshape(-1)
    mm.flush()
    del mm
    os.utime(bk, ns=(st0.st_atime_ns, st0.st_mtime_ns))  # worst case: the modification time did not move
    after = alt_freq()
    want = orc.alt_freq_dip_pseudo_cpp(imputed, rows, cols, ploidy, True, orc.CODE_IMPUTE_PRED)
    assert np.array_equal(after, want) and not np.array_equal(after, before)
    # a sparse edit (one genotype in a page the fingerprint may not sample) with the modification time put back: always seen
    # by default; under the cache only tpg_invalidate(BM) guarantees it
    mm = np.memmap(bk, dtype=np.uint8, mode="r+")
    j_edit = m // 2 + 7
    old_byte = int(mm[j_edit * n + 3])
    mm[j_edit * n + 3] = (old_byte + 1) % 3
    mm.flush()
    del mm
    os.utime(bk, ns=(st0.st_atime_ns, st0.st_mtime_ns))
    imputed2 = imputed.copy()
    imputed2[3, j_edit] = (old_byte + 1) % 3
    if cache:
        shim.call("tpg_invalidate", BM)
    assert np.array_equal(alt_freq(), orc.alt_freq_dip_pseudo_cpp(imputed2, rows, cols, ploidy, True, orc.CODE_IMPUTE_PRED))
    # a scattered colInd goes through the gathered upload (default) / the cached copy
    sub = cols[::9].copy()
    out = shim.call("alt_freq_dip_pseudo_cpp", BM, shim.int(rows), shim.int(sub), shim.real(ploidy), shim.int([1]),
                    shim.lib.rmock_lgl(1))
    assert np.array_equal(shim.as_numpy(out, (len(sub), 2)),
                          orc.alt_freq_dip_pseudo_cpp(imputed2, rows, sub, ploidy, True, orc.CODE_IMPUTE_PRED))


@pytest.mark.parametrize("stream_budget", [None, 512 << 10])
def test_whole_analysis_entry_points(shim, tmp_path, monkeypatch, stream_budget):
    """stream_budget: the backing file is "larger than the GPU may hold" (TPG_STREAM_BUDGET), so the same .Call entry points
    sweep it in blocks from the file mapping (csrc/stream.hip) -- what an R session with a 400-GB .bk gets"""
    monkeypatch.setenv("TPG_DEVICES", "1")
    if stream_budget:
        monkeypatch.setenv("TPG_STREAM_BUDGET", str(stream_budget))
    n, m, G, k = 160, 3000, 4, 5
    fbm = orc.synth_fbm(53, n, m, npop=G, miss=0.0)
    fbm[:, :2] = np.array([0, 1] * (n // 2), dtype=np.uint8)[:, None]  # no monomorphic loci by accident
    dec = fbm
    poly = (dec.sum(axis=0) > 0) & (dec.sum(axis=0) < 2 * n)
    cols = (np.where(poly)[0] + 1).astype(np.int32)
    bk = tmp_path / "geno.bk"
    fbm.T.tofile(bk)
    BM = shim.fbm(bk, n, m, orc.CODE_IMPUTE_PRED)
    rows = np.arange(1, n + 1, dtype=np.int32)
    gid = (np.arange(n) % G).astype(np.int32)
    mc = len(cols)
    out = shim.call("tpg_snp_pairwise", BM, shim.int(rows), shim.int(cols), shim.lib.rmock_lgl(0), shim.lib.rmock_nil())
    assert np.array_equal(shim.list_elt(out, 0, (n, n)), orc.snp_ibs(fbm, rows, cols), equal_nan=True)
    assert np.array_equal(shim.list_elt(out, 1, (n, n)), orc.snp_king(fbm, rows, cols), equal_nan=True)
    # which = king + grm (BASELINE config 2): four of the five cross-products; the others come back NULL
    out = shim.call("tpg_snp_pairwise", BM, shim.int(rows), shim.int(cols), shim.lib.rmock_lgl(0), shim.int([2 | 8]))
    assert shim.lib.TYPEOF(shim.lib.VECTOR_ELT(out, 0)) == 0 and shim.lib.TYPEOF(shim.lib.VECTOR_ELT(out, 2)) == 0  # NILSXP
    assert np.array_equal(shim.list_elt(out, 1, (n, n)), orc.snp_king(fbm, rows, cols), equal_nan=True)
    assert np.allclose(shim.list_elt(out, 3, (n, n)), orc.pairwise_grm(orc.snp_allele_sharing(fbm, rows, cols)), rtol=1e-12,
                       atol=1e-13, equal_nan=True)
    gf = shim.call("tpg_grouped_alt_freq", BM, shim.int(rows), shim.int(cols), shim.int(gid), shim.int([G]),
                   shim.real(np.full(n, 2.0)), shim.lib.rmock_lgl(0))
    assert np.array_equal(shim.as_numpy(gf, (mc, 2 * G)),
                          orc.grouped_alt_freq_dip_pseudo_cpp(fbm, rows, cols, gid, G, np.full(n, 2.0), code256=orc.CODE_IMPUTE_PRED))
    pairs = orc.combn2(G).astype(np.float64)  # utils::combn gives doubles
    fst = shim.call("tpg_pairwise_pop_fst", BM, shim.int(rows), shim.int(cols), shim.int(gid), shim.int([G]),
                    shim.real(np.full(n, 2.0)), shim.int([2]), shim.matrix(pairs), shim.lib.rmock_lgl(1), shim.lib.rmock_lgl(0))
    o = orc.pairwise_pop_fst(fbm, rows, cols, gid, G, method="WC84", by_locus=True, code256=orc.CODE_IMPUTE_PRED)
    P = pairs.shape[1]
    assert np.array_equal(shim.list_elt(fst, 0, (mc, P)), o["fst_locus"], equal_nan=True)
    assert np.allclose(shim.list_elt(fst, 1), o["fst_tot"], rtol=1e-12, atol=0)
    pca = shim.call("tpg_pca_partial_svd", BM, shim.int(rows), shim.int(cols), shim.int([k]))
    op = orc.gt_pca_partialSVD(fbm, rows, cols, k=k)
    assert np.allclose(shim.list_elt(pca, 0), op["d"], rtol=1e-6)
    assert np.array_equal(shim.list_elt(pca, 3), op["center"]) and np.array_equal(shim.list_elt(pca, 4), op["scale"])
    assert shim.list_elt(pca, 5)[0] == pytest.approx(op["square_frobenius"], rel=1e-12)
    # errors of the library arrive as R errors (Rf_error), as BEGIN_RCPP / END_RCPP deliver C++ exceptions
    with pytest.raises(RuntimeError, match="tidypopgen"):
        shim.call("tpg_pca_partial_svd", BM, shim.int(rows), shim.int(cols), shim.int([n + 1]))  # k > n
