"""GPU vs the CPU oracle AT the BASELINE.json sizes (configs 2 and 3-5), on samples the oracle finishes in seconds.

The panels are the bench panels (synthetic, seed 2: 1 000 x 650 000; seed 3: 5 000 x 1 000 000; 51 populations, 2 %
missing stored as imputed bytes).  The synthetic generator is a pure function of (seed, individual, locus), so the
oracle regenerates on the host exactly the rows / loci it needs (oracle.synth_rows / synth_fbm -- bit-identical to
the device generator, tests/test_gpu_parity.py::test_synth_matches_host) and restates the reference on them:

  * pairwise: 64 individuals spread over the first, middle and last (partial) row tiles x ALL loci.  The 64 x 64
    sub-blocks of all six count matrices of increment_{ibs,king,as}_counts must be bit-exact (this is where K-split
    > 1, multi-round unit tables and the 10^6-deep int32 accumulation of the full-size launch would show), and IBS /
    KING / allele sharing identical to the oracle's R-order epilogues;
  * Hudson / WC84 by locus for 5 population pairs x ALL loci bit-identical, totals <= 1e-12;
  * alt_freq, missingness and grouped_alt_freq (all 51 groups) on 30 000 sampled loci bit-exact, read out of the
    whole-panel device result;
  * PCA: numpy.linalg.eigh (LAPACK) of the device Gram matrix vs d^2 (<= 1e-9) and vs u (sign-aligned, 1e-6 where the
    spectral gap allows), four 64 x 64 sub-blocks of the Gram matrix (spread sample, the partial last row tile, a middle
    tile, the first tile edges) vs an FP64 numpy Gram over all loci (<= 1e-6 of its scale), center / scale vs the counts,
    the loadings v on 30 000 sampled loci vs FP64 Z'u / d from oracle-generated columns (<= 1e-9), and
    fbm256_prod_and_rowSumsSq at full size vs a numpy restatement for the 64 sampled individuals over all loci (<= 1e-9).
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = 51


def _sample_rows(n):
    """64 individuals: both ends of the first 64-row super-tile, a run across a tile edge in the middle, scattered
    ones (stride coprime with the 51 populations), and the last rows (the partial last tile)."""
    mid = (n // 2) // 32 * 32
    rows = [0, 1, 30, 31, 32, 33, 62, 63, 64, 65, mid - 2, mid - 1, mid, mid + 1, mid + 31, mid + 32]
    rows += list(range(n - 12, n))
    rng = np.random.default_rng(n)
    extra = [int(x) for x in rng.permutation(n) if x not in rows][: 64 - len(rows)]
    return np.array(sorted(rows + extra), dtype=np.int64)


def _oracle_pair_counts(orc, seed, rows, m):
    """the six count matrices of increment_{ibs,king,as}_counts on the sampled rows x ALL loci (the oracle's restatement)"""
    sub = orc.synth_rows(seed, rows, m, npop=G, miss=0.02, imputed_bytes=True)  # 64 x m FBM bytes of those rows
    o = {k: np.zeros((64, 64), order="F") for k in ("ibs", "ibs_valid", "king_num", "n_Aa_i", "as_num", "as_den")}
    orc.increment_ibs_counts(o["ibs"], o["ibs_valid"], sub, None, None)        # src/snp_ibs.cpp:45-72
    orc.increment_king_numerator(o["king_num"], o["n_Aa_i"], sub, None, None)  # src/snp_king.cpp:45-72
    orc.increment_as_counts(o["as_num"], o["as_den"], sub, None, None)         # src/snp_as.cpp:44-65
    return o


# what each product set's accumulators can give (include/tpg.h: IBS = V + D + H, IBS_valid = 2 V, KING = D - V + A + A',
# N_Aa_i = A, AS = D / V): the count matrices and the epilogues that need nothing else
_SET_OUTPUTS = {
    "all": (("ibs", "ibs_valid", "king_num", "n_Aa_i", "as_num", "as_den"), ("ibs", "king", "allele_sharing", "grm")),
    "as": (("ibs_valid", "as_num", "as_den"), ("allele_sharing", "grm")),
    "ibs": (("ibs", "ibs_valid", "as_num", "as_den"), ("ibs", "allele_sharing", "grm")),
    "ibs1": (("ibs", "ibs_valid", "as_den"), ("ibs",)),  # V and D + H in one sum (TPG_PW_DH): what snp_ibs alone runs on
    "king": (("ibs_valid", "king_num", "n_Aa_i", "as_num", "as_den"), ("king", "allele_sharing", "grm")),
}


def _check_counts_and_epilogues(tpg, orc, pw, o, rows, m, which):
    ix = np.ix_(rows, rows)
    counts, eps = _SET_OUTPUTS[which]
    for names in (counts[:3], counts[3:]):
        if not names:
            continue
        c = pw.counts(names)
        for k in names:
            assert np.array_equal(c[k][ix], o[k]), (which, k)
        del c
    ep = pw.epilogues(eps, m=m)
    with np.errstate(invalid="ignore", divide="ignore"):
        if "ibs" in eps:
            assert np.array_equal(ep["ibs"][ix], o["ibs"] / o["ibs_valid"], equal_nan=True)   # R/snp_ibs.R:88-95
    if "king" in eps:
        assert np.array_equal(ep["king"][ix], orc.king_epilogue(o["king_num"], o["n_Aa_i"]), equal_nan=True)
    if "allele_sharing" in eps:
        assert np.array_equal(ep["allele_sharing"][ix], orc.as_epilogue(o["as_num"], o["as_den"]), equal_nan=True)
        # GRM = the reference's formula on the (sample-verified) allele-sharing matrix: the mean is over all N (N - 1) pairs
        assert np.allclose(ep["grm"], orc.pairwise_grm(ep["allele_sharing"]), rtol=1e-12, atol=1e-14)
    return ep


def _check_pairwise_sample(tpg, orc, X, seed, n, m, monkeypatch=None):
    """The five-product kernel of the fused pass AND the three product-set kernels a stand-alone snp_ibs / snp_king /
    pairwise_grm (and every block of an R driver loop) runs on -- tpg_pairwise_set_kernel<RA, RB, MASK, NS>: K split > 1,
    multi-round unit tables, the 128 x 64 / 64 x 64 wave tiles' diagonal handling and partial last tiles at this N -- each
    against the same oracle sample; then the {V, D} workgroup form (LDS-DMA ring, TPG_PW_VARIANT=14) once."""
    rows = _sample_rows(n)
    assert len(set(rows.tolist())) == 64
    o = _oracle_pair_counts(orc, seed, rows, m)
    v = tpg.View(X, code256=None)
    pw = tpg.Pairwise(X.ctx, n)
    sets = {"all": None, "as": tpg.PW_FOR_AS, "ibs": tpg.PW_FOR_IBS, "ibs1": tpg.PW_FOR_IBS_ALONE, "king": tpg.PW_FOR_KING}
    for which, products in sets.items():
        pw.zero()
        pw.accumulate(v, products=products)
        assert pw.products() == (tpg.PW_ALL if products is None else products)
        ep = _check_counts_and_epilogues(tpg, orc, pw, o, rows, m, which)
        if which == "king":
            # BASELINE config 2's literal workload -- pairwise_king + pairwise_grm -- is this kernel + these two epilogues
            kg = pw.epilogues(which=("king", "grm"))
            assert np.array_equal(kg["king"], ep["king"], equal_nan=True) and np.array_equal(kg["grm"], ep["grm"])
            del kg
        del ep
    if monkeypatch is not None:
        monkeypatch.setenv("TPG_PW_VARIANT", "14")
        pw.zero()
        pw.accumulate(v, products=tpg.PW_FOR_AS)
        monkeypatch.delenv("TPG_PW_VARIANT")
        _check_counts_and_epilogues(tpg, orc, pw, o, rows, m, "as")
    pw.free()
    return v, o


def _check_increment_block_loop(tpg, orc, X, o, n, m):
    """The reference's own block loop over the literal mirrors (R/snp_ibs.R:59-82, R/snp_king.R:51-77,
    R/snp_allele_sharing.R:49-69): blocks of bigstatsr::block_size(n) loci cut by CutBySize, one increment_* call per
    block on the host FBM, the caller's N x N double matrices incremented in place -- against the same oracle sample."""
    rows = _sample_rows(n)
    ix = np.ix_(rows, rows)
    fbm = X.to_numpy()
    lo, up = tpg.cut_by_size(m, tpg.block_size(n))  # CutBySize, R/local_reimplementations.R:13-15
    nb = len(lo)
    rowInd = np.arange(1, n + 1, dtype=np.int32)
    for fn, names in ((tpg.increment_ibs_counts, ("ibs", "ibs_valid")), (tpg.increment_king_numerator, ("king_num", "n_Aa_i")),
                      (tpg.increment_as_counts, ("as_num", "as_den"))):
        K, K2 = np.zeros((n, n), order="F"), np.zeros((n, n), order="F")
        for b in range(nb):
            fn(K, K2, fbm, rowInd, np.arange(lo[b], up[b] + 1, dtype=np.int32), ctx=X.ctx)
        assert np.array_equal(K[ix], o[names[0]]), names[0]
        assert np.array_equal(K2[ix], o[names[1]]), names[1]
        assert np.array_equal(K, K.T) and (names[1] == "n_Aa_i" or np.array_equal(K2, K2.T))
        del K, K2
    tpg.resident_drop(X.ctx)
    return nb


def _check_loci_sample(tpg, orc, X, v012, seed, n, m, nblocks=30, width=1000):
    gid = (np.arange(n) % G).astype(np.int32)
    ploidy = np.full(n, 2.0)
    cnt = tpg.loci_counts(v012)
    f = tpg.alt_freq_dip_pseudo_cpp(v012, None, as_counts=False)
    ga = tpg.grouped_alt_freq_dip_pseudo_cpp(v012, gid, G, None, as_counts=False)
    gm = tpg.grouped_missingness_cpp(v012, gid, G)
    starts = np.linspace(0, m - width, nblocks).astype(np.int64)
    starts[1:-1] += 37  # off the 128-locus group grid, but keep the very first and very last loci
    for j0 in starts:
        blk = orc.synth_fbm(seed, n, width, j0=int(j0), npop=G, miss=0.02, imputed_bytes=True)
        sl = slice(int(j0), int(j0) + width)
        of = orc.alt_freq_dip_pseudo_cpp(blk, None, None, ploidy)                     # src/alt_freq_dip_pseudo_cpp.cpp:21-57
        assert np.array_equal(f[sl], of, equal_nan=True)
        og = orc.grouped_alt_freq_dip_pseudo_cpp(blk, None, None, gid, G, ploidy)     # src/grouped_alt_freq_dip_pseudo_cpp.cpp:24-57
        assert np.array_equal(ga[sl], og, equal_nan=True)
        om = orc.grouped_missingness_cpp(blk, None, None, gid, G)                     # src/grouped_missingness_cpp.cpp:21-32
        assert np.array_equal(gm[sl], om)
        codes = np.minimum(blk, 3)
        for k in range(4):
            assert np.array_equal(cnt[sl, k], (codes == k).sum(axis=0))
    return cnt


def _check_fst_sample(tpg, orc, X, seed, n, m):
    gid = (np.arange(n) % G).astype(np.int32)
    pairs_full = np.array([[1, 2], [1, 51], [26, 27], [50, 51], [11, 41]], dtype=np.int32).T  # 2 x 5, 1-based
    pops = sorted(set(pairs_full.ravel().tolist()))
    remap = {p: k + 1 for k, p in enumerate(pops)}
    pairs_sub = np.vectorize(remap.get)(pairs_full).astype(np.int32)
    rows = np.array([i for i in range(n) if (i % G) + 1 in pops], dtype=np.int64)
    gid_sub = np.array([remap[(i % G) + 1] - 1 for i in rows], dtype=np.int32)
    sub = orc.synth_rows(seed, rows, m, npop=G, miss=0.02, imputed_bytes=True)
    pf = orc.grouped_summaries_dip_pseudo_cpp(sub, None, None, gid_sub, len(pops), np.full(len(rows), 2.0))
    del sub
    with np.errstate(invalid="ignore", divide="ignore"):
        for method in ("Hudson", "WC84", "Nei87"):
            if method == "Hudson":  # src/pairwise_fst_hudson_loop.cpp:23-62
                o = orc.pairwise_fst_hudson_loop(pairs_sub, pf["n"], pf["freq_alt"], pf["freq_ref"], by_locus=True)
                ond = orc.pairwise_fst_hudson_loop(pairs_sub, pf["n"], pf["freq_alt"], pf["freq_ref"], by_locus=True,
                                                   return_num_dem=True)
            elif method == "Nei87":  # src/pairwise_fst_nei87_loop.cpp:23-114 (not in BASELINE's list: run beside the two that are)
                o = orc.pairwise_fst_nei87_loop(pairs_sub, pf["n"], pf["het_obs"], pf["freq_alt"], pf["freq_ref"], by_locus=True)
                ond = orc.pairwise_fst_nei87_loop(pairs_sub, pf["n"], pf["het_obs"], pf["freq_alt"], pf["freq_ref"],
                                                  by_locus=True, return_num_dem=True)
            else:                   # src/pairwise_fst_wc84_loop.cpp:22-120
                o = orc.pairwise_fst_wc84_loop(pairs_sub, pf["n"], pf["freq_alt"], pf["het_obs"], by_locus=True)
                ond = orc.pairwise_fst_wc84_loop(pairs_sub, pf["n"], pf["freq_alt"], pf["het_obs"], by_locus=True,
                                                 return_num_dem=True)
            d = tpg.pairwise_pop_fst(X, None, None, gid, G, method=method, by_locus=True, pairwise_combn=pairs_full)
            assert np.array_equal(d["fst_locus"], o["fst_locus"], equal_nan=True), method
            dnd = tpg.pairwise_pop_fst(X, None, None, gid, G, method=method, return_num_dem=True,
                                       pairwise_combn=pairs_full)
            for key in ("Fst_by_locus_num", "Fst_by_locus_den"):
                assert np.array_equal(dnd[key], ond[key], equal_nan=True), (method, key)
            # Totals: the per-locus terms are bit-identical (above), so the totals differ only by the ORDER of a sum
            # of ~10^6 mixed-sign terms (the reference adds them one after the other, :43-52; the device adds chunk
            # partials).  Both are therefore compared with the correctly rounded sums (math.fsum) of those terms: the
            # device must be within 1e-12 of it, and the reference's own sequential sum is no closer than that either.
            num, den = ond["Fst_by_locus_num"], ond["Fst_by_locus_den"]
            exact = np.zeros(num.shape[1])
            for c in range(num.shape[1]):
                ok = ~(np.isnan(num[:, c]) | np.isnan(den[:, c]))
                exact[c] = math.fsum(num[ok, c]) / math.fsum(den[ok, c])
            assert np.allclose(d["fst_tot"], exact, rtol=1e-12, atol=0), (method, d["fst_tot"] / exact - 1)
            assert np.allclose(o["fst_tot"], exact, rtol=1e-10, atol=0), (method, o["fst_tot"] / exact - 1)
            assert np.allclose(d["fst_tot"], o["fst_tot"], rtol=1e-10, atol=0), method
            # the fused totals-only path (what bench.py times; WC84 in its hand-reduced form with fast reciprocals)
            s = tpg.pairwise_pop_fst(X, None, None, gid, G, method=method, pairwise_combn=pairs_full, sums=True)
            assert np.allclose(s["fst_tot"], exact, rtol=1e-11, atol=0), (method, s["fst_tot"] / exact - 1)


def _fp64_gram_of_rows(orc, seed, rows, m, keep, center, scale):
    """Z Z' of a few individuals over all kept loci in FP64 numpy, from oracle-generated rows"""
    sub = orc.CODE_IMPUTE_PRED[orc.synth_rows(seed, rows, m, npop=G, miss=0.02, imputed_bytes=True)][:, keep]
    Ks = np.zeros((len(rows), len(rows)))
    for a in range(0, sub.shape[1], 65536):
        Z = (sub[:, a:a + 65536] - center[a:a + 65536]) / scale[a:a + 65536]
        Ks += Z @ Z.T
    return Ks, sub


def _check_pca_sample(tpg, orc, X, seed, n, m, k):
    rows = _sample_rows(n)
    vi = tpg.View(X, code256=tpg.CODE_IMPUTE_PRED)
    cnt = tpg.loci_counts(vi).astype(np.int64)
    assert cnt[:, 3].sum() == 0
    alt = cnt[:, 1] + 2 * cnt[:, 2]
    keep = (alt > 0) & (alt < 2 * n)
    cols = (np.where(keep)[0] + 1).astype(np.int32)
    r = tpg.gt_pca_partialSVD(X, None, cols, k=k)
    # center / scale of bigsnpr::snp_scaleBinom from the (oracle-checked) counts
    center = alt[keep] / n
    p = center / 2
    scale = np.sqrt(2 * p * (1 - p))
    assert np.array_equal(r["center"], center)
    assert np.allclose(r["scale"], scale, rtol=1e-15, atol=0)
    vv = tpg.View(X, None, cols, code256=tpg.CODE_IMPUTE_PRED)
    K = tpg.pca_gram(vv, r["center"], r["scale"])
    # eigenvalues AND eigenvectors of the device Gram matrix by LAPACK (numpy.linalg.eigh) against d^2 and u
    lam_all, U_all = np.linalg.eigh(K)
    lam, U = lam_all[::-1][:k], U_all[:, ::-1][:, :k]
    assert np.allclose(r["d"] ** 2, lam, rtol=1e-9, atol=0)
    gaps = np.minimum(np.abs(np.diff(lam_all[::-1][:k + 1])), np.abs(np.diff(np.concatenate([[np.inf], lam]))))
    sign = np.sign((U * r["u"]).sum(axis=0))
    du = np.abs(r["u"] - U * sign).max(axis=0)
    # an eigenvector is determined up to (residual) / (gap to its neighbours): 1e-6 wherever the gap allows it
    tol_u = np.maximum(1e-6, 1e-10 * lam[0] / gaps)
    assert np.all(du <= tol_u), (du, tol_u)
    # the Gram matrix itself against FP64 numpy over ALL kept loci on four 64 x 64 sub-blocks: the spread sample, the
    # last 64 rows (the partial last row tile), a tile-aligned run in the middle, and rows around the first tile edges
    mid = (n // 2) // 64 * 64
    row_sets = [rows, np.arange(n - 64, n), np.arange(mid, mid + 64), np.arange(17, 17 + 64)]
    sub0 = None
    for rs in row_sets:
        rs = np.asarray(rs, dtype=np.int64)
        Ks, sub = _fp64_gram_of_rows(orc, seed, rs, m, keep, center, scale)
        if sub0 is None:
            sub0 = sub
        Kd = K[np.ix_(rs, rs)]
        # every entry relative to the scale of ITS pair, sqrt(K_ii K_jj) (the diagonal is ~10^6, an off-diagonal entry 10^3 -
        # 10^4: a bound relative to max |K| would let one be wrong by 1 in 10^3).  1e-9 of that scale is 3e-7 .. 1e-5 relative
        # for an off-diagonal entry: the class path's weights are within 2^-47 and its fold within 3e-10 of max |K|
        dg = np.sqrt(np.abs(np.diag(Ks)))
        assert np.all(np.abs(Kd - Ks) <= 1e-9 * np.outer(dg, dg)), float((np.abs(Kd - Ks) / np.outer(dg, dg)).max())
        assert np.allclose(np.diag(Kd), np.diag(Ks), rtol=1e-9)
    # u spans eigenvectors of K
    assert np.abs(K @ r["u"] - r["u"] * r["d"] ** 2).max() <= 1e-9 * r["d"][0] ** 2
    del K, U_all
    # loadings v = Z'u / d (R/gt_pca_partialSVD.R:82-89 via big_SVD) on 30 blocks of 1 000 loci, from oracle-generated
    # columns in FP64 numpy with the (LAPACK-checked) device u
    kept_index = np.cumsum(keep) - 1  # locus -> row of v
    width = 1000
    starts = np.linspace(0, m - width, 30).astype(np.int64)
    starts[1:-1] += 37
    vmax = np.abs(r["v"]).max()
    for j0 in starts:
        blk = orc.CODE_IMPUTE_PRED[orc.synth_fbm(seed, n, width, j0=int(j0), npop=G, miss=0.02, imputed_bytes=True)]
        kb = keep[j0:j0 + width]
        ji = kept_index[j0:j0 + width][kb]
        Z = (blk[:, kb] - center[ji]) / scale[ji]
        v_ref = (Z.T @ r["u"]) / r["d"]
        assert np.abs(r["v"][ji] - v_ref).max() <= 1e-9 * vmax, int(j0)
    # fbm256_prod_and_rowSumsSq at full size (src/fbm_prod_and_rowSumSq.cpp:30-44: XV[i,k] = sum_j z_ij V[j,k],
    # rss[i] = sum_j z_ij^2, missing -> 0) against numpy on the 64 sampled individuals over ALL kept loci
    XV, rss = tpg.fbm256_prod_and_rowSumsSq(X, None, cols, center, scale, r["v"], code256=tpg.CODE_IMPUTE_PRED)
    XV_ref = np.zeros((64, k))
    rss_ref = np.zeros(64)
    for a in range(0, sub0.shape[1], 65536):
        Z = (sub0[:, a:a + 65536] - center[a:a + 65536]) / scale[a:a + 65536]
        XV_ref += Z @ r["v"][a:a + 65536]
        rss_ref += (Z * Z).sum(axis=1)
    assert np.abs(XV[rows] - XV_ref).max() <= 1e-9 * np.abs(XV_ref).max()
    assert np.allclose(rss[rows], rss_ref, rtol=1e-10)
    # and X V = u d (the SVD relation) for everybody
    assert np.abs(XV - r["u"] * r["d"]).max() <= 1e-8 * r["d"][0]
    return r


def test_config2_hgdp_shape_against_oracle(monkeypatch):
    """BASELINE config 2: 1 000 x 650 000, pairwise_king + pairwise_grm (+ the per-locus sweeps and PCA)"""
    import tidypopgen_amd as tpg
    from oracle import oracle as orc

    n, m, seed = 1000, 650_000, 2
    X = tpg.FBM.synth(seed, n, m, npop=G, miss=0.02, imputed_bytes=True)
    _, o = _check_pairwise_sample(tpg, orc, X, seed, n, m, monkeypatch)
    assert _check_increment_block_loop(tpg, orc, X, o, n, m) == 5  # block_size(1000) = 134 217 loci
    v012 = tpg.View(X)
    _check_loci_sample(tpg, orc, X, v012, seed, n, m, nblocks=12)
    _check_fst_sample(tpg, orc, X, seed, n, m)
    _check_pca_sample(tpg, orc, X, seed, n, m, k=10)


def test_config3_to_5_bench_panel_against_oracle(monkeypatch):
    """BASELINE configs 3-5: 5 000 x 1 000 000, 51 populations, k = 20 -- the bench.py panel (seed 3)"""
    import tidypopgen_amd as tpg
    from oracle import oracle as orc

    n, m, seed = 5000, 1_000_000, 3
    X = tpg.FBM.synth(seed, n, m, npop=G, miss=0.02, imputed_bytes=True)
    _, o = _check_pairwise_sample(tpg, orc, X, seed, n, m, monkeypatch)
    assert _check_increment_block_loop(tpg, orc, X, o, n, m) == 38  # the 38 blocks of 26 315 / 26 316 loci of the R drivers
    v012 = tpg.View(X)
    _check_loci_sample(tpg, orc, X, v012, seed, n, m, nblocks=30)
    del v012
    _check_fst_sample(tpg, orc, X, seed, n, m)
    _check_pca_sample(tpg, orc, X, seed, n, m, k=20)
