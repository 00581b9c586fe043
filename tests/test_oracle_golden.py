"""Pins the CPU oracle against the golden vectors / known answers that the
reference's own tests hold for the hot path.  CPU only."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import fixtures as fx


def test_split_len_matches_survey_example():
    lo, up = orc.cut_by_size(961, 300)
    assert list(up - lo + 1) == [240, 240, 241, 240]
    lo, up = orc.cut_by_size(6, 3)
    assert list(up - lo + 1) == [3, 3]


def test_ibs_hand_counts_and_block_invariance():
    # tests/testthat/test_snp_ibs.R:28-66
    fbm = orc.fbm_from_genotypes(fx.IBS_3x6)
    raw = orc.snp_ibs(fbm, type="raw_counts")
    assert raw["ibs"][0, 1] == sum([1, 2, 2, 1, 1, 2])
    raw2 = orc.snp_ibs(fbm, block_size=3, type="raw_counts")
    assert np.array_equal(raw2["ibs"], raw["ibs"]) and np.array_equal(raw2["valid_n"], raw["valid_n"])
    sub = orc.snp_ibs(fbm, ind_row=[1, 3], ind_col=[2, 3, 5, 6], type="raw_counts")
    assert sub["ibs"][0, 1] == sum(np.array([1, 1, 2, 1, 2, 1])[[1, 2, 4, 5]])


def test_ibs_plink_golden():
    # tests/testthat/test_snp_ibs.R:69-105
    prop = orc.snp_ibs(fx.families_fbm())
    assert np.array_equal(np.round(prop, 6), fx.plink_mibs())
    # unequal blocks give the same answer
    prop_b = orc.snp_ibs(fx.families_fbm(), block_size=300)
    assert np.array_equal(prop_b, prop)


def test_king_r_restatement_and_blocks():
    # tests/testthat/test_snp_king.R:155-186
    fbm = orc.fbm_from_genotypes(fx.IBS_3x6)
    k = orc.snp_king(fbm)
    assert np.array_equal(k, fx.king_r(fx.IBS_3x6), equal_nan=True)
    assert np.array_equal(orc.snp_king(fbm, block_size=3), k, equal_nan=True)
    fam = fx.families_fbm()
    X = fam.astype(float)
    X[X == 3] = np.nan
    kf = orc.snp_king(fam)
    assert np.array_equal(kf, fx.king_r(X), equal_nan=True)
    assert np.array_equal(orc.snp_king(fam, block_size=300), kf, equal_nan=True)


def test_king_golden_kin0():
    # tests/testthat/test_snp_king.R:189-236 (tolerance 0.06 there; 4-dp file here)
    kf = orc.snp_king(fx.families_fbm())
    gold = fx.king_kin0_matrix()
    assert np.nanmax(np.abs(kf - gold)) < 1e-4


def test_allele_sharing_definition_and_blocks():
    # tests/testthat/test_pairwise_allele_sharing.R:29-47 (hierfstat::matching)
    fbm = orc.fbm_from_genotypes(fx.AS_3x6)
    a = orc.snp_allele_sharing(fbm)
    assert np.allclose(a, fx.matching(fx.AS_3x6), rtol=0, atol=1e-15, equal_nan=True)
    assert np.array_equal(orc.snp_allele_sharing(fbm, block_size=3), a, equal_nan=True)
    fam = fx.families_fbm()
    X = fam.astype(float)
    X[X == 3] = np.nan
    assert np.allclose(orc.snp_allele_sharing(fam), fx.matching(X), rtol=0, atol=1e-14)


def test_allele_sharing_pad_quirk_switch():
    # unequal blocks add +1 to every numerator element per narrow block
    fam = fx.families_fbm()
    n = 12
    num = np.zeros((n, n), order="F"); den = np.zeros((n, n), order="F")
    numq = np.zeros((n, n), order="F"); denq = np.zeros((n, n), order="F")
    lo, up = orc.cut_by_size(961, 300)
    widest = max(up - lo + 1)
    cols = np.arange(1, 962, dtype=np.int32)
    narrow = 0
    for a, b in zip(lo, up):
        orc.increment_as_counts(num, den, fam, None, cols[a - 1:b])
        q = (b - a + 1) < widest
        narrow += q
        orc.increment_as_counts(numq, denq, fam, None, cols[a - 1:b], pad_quirk=q)
    assert narrow == 3
    assert np.array_equal(numq - num, np.full((n, n), 3.0)) and np.array_equal(den, denq)


def test_grm_definition():
    # tests/testthat/test_pairwise_grm.R:33-41: 2 * beta.dosage(inb=FALSE), i.e.
    # (M - mb)/(1 - mb) * 2 with M = matching and mb the mean off-diagonal M
    fbm = orc.fbm_from_genotypes(fx.FST_7x6)
    M = fx.matching(fx.FST_7x6)
    off = M[~np.eye(7, dtype=bool)]
    expect = 2 * (M - off.mean()) / (1 - off.mean())
    assert np.allclose(orc.pairwise_grm(orc.snp_allele_sharing(fbm)), expect, rtol=0, atol=1e-14)


def test_alt_freq_known_answers():
    # tests/testthat/test_loci_freq.R:1-96
    g = fx.FREQ_3x6
    fbm = orc.fbm_from_genotypes(g)
    freq = np.nansum(g, axis=0) / (np.array([3, 3, 3, 2, 3, 1]) * 2)
    assert np.array_equal(orc.loci_alt_freq(fbm), freq)
    counts = orc.loci_alt_freq(fbm, as_counts=True)
    assert np.array_equal(counts[:, 0] / counts[:, 1], freq)
    # subset: remove 2nd individual and 3rd, 5th snp
    f1 = orc.loci_alt_freq(fbm, ind_row=[1, 3], ind_col=[1, 2, 4, 6])
    assert np.array_equal(f1, np.nansum(g[[0, 2]][:, [0, 1, 3, 5]], axis=0) / (np.array([2, 2, 2, 1]) * 2))
    # all-missing locus -> NaN  (:50-63)
    f2 = orc.loci_alt_freq(fbm, ind_row=[2, 3], ind_col=[1, 2, 5, 6])
    assert np.array_equal(f2[:3], np.nansum(g[1:][:, [0, 1, 4]], axis=0) / 4) and np.isnan(f2[3])
    f3 = orc.loci_alt_freq(orc.fbm_from_genotypes(fx.FREQ2_3x6), ind_row=[2, 3])
    assert np.isnan(f3[3]) and np.isnan(f3[5])
    # block invariance (:93-95)
    assert np.array_equal(orc.loci_alt_freq(fbm, block_size=2), orc.loci_alt_freq(fbm))


def test_missingness_known_answers():
    # tests/testthat/test_loci_missingness.R:27-75
    g = fx.FREQ_3x6
    fbm = orc.fbm_from_genotypes(g)
    n_na = np.isnan(g).sum(axis=0)
    assert np.array_equal(orc.loci_missingness(fbm, as_counts=True), n_na)
    assert np.array_equal(orc.loci_missingness(fbm), n_na / 3)
    assert np.array_equal(orc.loci_missingness(fbm, ind_row=[2, 3], as_counts=True), np.isnan(g[1:]).sum(axis=0))


def test_grouped_counts_against_group_map():
    # tests/testthat/test_loci_freq.R:98-257 / test_loci_missingness.R:77-160 compare the grouped
    # kernels with per-group calls of the ungrouped ones
    g = fx.FST_7x6
    fbm = orc.fbm_from_genotypes(g)
    gid = fx.FST_GROUPS_3
    ploidy = np.full(7, 2.0)
    ga = orc.grouped_alt_freq_dip_pseudo_cpp(fbm, None, None, gid, 3, ploidy)
    gm = orc.grouped_missingness_cpp(fbm, None, None, gid, 3)
    for k in range(3):
        rows = np.where(gid == k)[0] + 1
        with np.errstate(invalid="ignore"):
            f = orc.alt_freq_dip_pseudo_cpp(fbm, rows, None, np.full(len(rows), 2.0), as_counts=True)
        assert np.array_equal(ga[:, k], f[:, 0] / f[:, 1], equal_nan=True)
        assert np.array_equal(ga[:, 3 + k], f[:, 1])
        assert np.array_equal(gm[:, k], orc.loci_missingness(fbm, ind_row=rows, as_counts=True))


def _fst(g, gid, G, method, **kw):
    return orc.pairwise_pop_fst(orc.fbm_from_genotypes(g), None, None, gid, G, method=method, **kw)


def test_fst_scikit_allel_golden():
    # tests/testthat/test_pairwise_pop_fst.R:55-343
    gid = fx.FST_GROUPS_2
    for method, tag in (("Hudson", "fst_hudson"), ("WC84", "fst_wc")):
        tot = _fst(fx.FST_7x6, gid, 2, method)["fst_tot"]
        assert tot[0] == pytest.approx(float(fx.scikit(tag)), rel=0, abs=2e-16)
        per = _fst(fx.FST_7x6, gid, 2, method, by_locus=True)["fst_locus"][:, 0]
        assert np.allclose(per, fx.scikit(tag + "_per_loc"), rtol=0, atol=3e-16)
        mono = _fst(fx.FST_MONO_7x6, gid, 2, method)["fst_tot"]
        assert mono[0] == pytest.approx(float(fx.scikit(tag + "_monomorphic")), rel=0, abs=2e-16)
        # locus missing in a whole population is ignored (:144-194, :290-342)
        a = _fst(fx.FST_MISSPOP_7x6, gid, 2, method)["fst_tot"]
        b = _fst(fx.FST_MISSPOP_7x5, gid, 2, method)["fst_tot"]
        assert a[0] == b[0]


def test_fst_num_dem_consistency_and_three_pops():
    gid = fx.FST_GROUPS_3
    for method in ("Hudson", "WC84", "Nei87"):
        nd = _fst(fx.FST_7x6, gid, 3, method, return_num_dem=True)
        ratio = _fst(fx.FST_7x6, gid, 3, method, by_locus=True)
        with np.errstate(invalid="ignore", divide="ignore"):
            assert np.array_equal(nd["Fst_by_locus_num"] / nd["Fst_by_locus_den"], ratio["fst_locus"], equal_nan=True)
        num, den = nd["Fst_by_locus_num"], nd["Fst_by_locus_den"]
        ok = ~np.isnan(num) & ~np.isnan(den)
        tot = np.array([num[ok[:, c], c].sum() / den[ok[:, c], c].sum() for c in range(3)])
        assert np.allclose(tot, ratio["fst_tot"], rtol=1e-15, atol=0)
    assert np.array_equal(orc.combn2(4), np.array([[1, 1, 1, 2, 2, 3], [2, 3, 4, 3, 4, 4]]))


def test_pca_against_plain_pca_definition():
    # tests/testthat/test_gt_pca.R:320-374 pins gt_pca_partialSVD to prcomp at 1e-4: std.dev = d/sqrt(n-1),
    # percent = d^2/||Z||_F^2.  Same check here with numpy's SVD on the scaled matrix.
    fam = fx.families_fbm()
    X = fam.astype(float)
    keep = np.where(((X == 3).sum(axis=0) == 0))[0]
    maf = X[:, keep].sum(axis=0) / 24
    maf = np.minimum(maf, 1 - maf)
    keep = keep[maf > 0.01]
    assert len(keep) == 609  # tests/testthat/test_gt_pca.R:337
    cols = keep + 1
    res = orc.gt_pca_partialSVD(fam, None, cols, k=10, code256=orc.CODE_012)
    Z = (X[:, keep] - res["center"]) / res["scale"]
    s = np.linalg.svd(Z, compute_uv=False)
    assert np.allclose(res["d"], s[:10], rtol=1e-10)
    assert res["square_frobenius"] == pytest.approx((Z ** 2).sum(), rel=1e-12)
    # scores u*d == Z v (sign-consistent by construction)
    assert np.allclose(res["u"] * res["d"], Z @ res["v"], atol=1e-9)
    XV, rss = orc.fbm256_prod_and_rowSumsSq(fam, None, cols, res["center"], res["scale"], res["v"])
    assert np.allclose(XV, Z @ res["v"], atol=1e-10) and np.allclose(rss, (Z ** 2).sum(axis=1), rtol=1e-12)


def test_blas_path_matches_integer_oracle():
    fbm = orc.synth_fbm(7, 40, 300, npop=5)
    r = np.arange(1, 41, dtype=np.int32); c = np.arange(1, 301, dtype=np.int32)
    for inc_o, inc_b in ((orc.increment_ibs_counts, orc.blas_increment_ibs),
                         (orc.increment_king_numerator, orc.blas_increment_king),
                         (orc.increment_as_counts, orc.blas_increment_as)):
        A = np.zeros((40, 40), order="F"); B = np.zeros((40, 40), order="F")
        A2 = np.zeros((40, 40), order="F"); B2 = np.zeros((40, 40), order="F")
        inc_o(A, B, fbm, r, c)
        inc_b(A2, B2, fbm, r, c)
        assert np.array_equal(A, A2) and np.array_equal(B, B2)


def test_synth_is_deterministic_and_shardable():
    a = orc.synth_fbm(3, 50, 200, npop=7)
    b = orc.synth_fbm(3, 50, 80, j0=120, npop=7)
    assert np.array_equal(a[:, 120:], b)
    assert set(np.unique(a)) <= {0, 1, 2, 3} and 0.005 < (a == 3).mean() < 0.05
    c = orc.synth_fbm(3, 50, 200, npop=7, imputed_bytes=True)
    assert set(np.unique(c)) <= {0, 1, 2, 4, 5, 6}
    assert np.array_equal(np.where(a == 3, 9, a), np.where(c > 3, 9, c))


def test_windows_stats_generic_reference_expectations():
    # tests/testthat/test_window_stats_generic.R:1-86: the values the reference asserts (runner's window rule)
    x = np.array([1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 14, 15, 16], dtype=float)
    chrom = np.array(["chr1"] * 6 + ["chr2"] * 7)
    pos = np.array([50, 120, 150, 180, 230, 390, 110, 120, 150, 180, 230, 280, 350])
    w = orc.windows_stats_generic(x, chrom, pos, "sum", 4, 3, "snp", 1)
    assert list(w["n_loci"]) == [4, 3, 4, 4]
    assert w["stat"][0] == x[0:4].sum() and w["stat"][3] == x[9:13].sum()
    wc = orc.windows_stats_generic(x, chrom, pos, "sum", 4, 3, "snp", 1, complete=True)
    assert np.isnan(wc["stat"][1])
    wb = orc.windows_stats_generic(x, chrom, pos, "sum", 100, 50, "bp", 1)
    c2 = wb["chromosome"] == "chr2"
    assert wb["start"][c2].min() == 101
    assert wb["n_loci"][c2 & (wb["start"] == 101)][0] == 4
    assert wb["stat"][c2 & (wb["start"] == 251)][0] == 31
    c1 = wb["chromosome"] == "chr1"
    assert np.isnan(wb["stat"][c1 & (wb["start"] == 251)][0])  # chr1 window 251-350 is empty


def test_loci_pi_by_hand():
    # tests/testthat/test_loci_pi.R:28-55: the test's own `pi` function, on the full matrix and on the subset
    # without individual 2 and loci 3, 5; a single individual gives NA
    def pi_by_hand(x):
        n = (~np.isnan(x)).sum(axis=0) * 2
        c0 = np.nansum(x, axis=0)
        c1 = n - c0
        with np.errstate(invalid="ignore", divide="ignore"):
            return c0 * c1 / (n * (n - 1) / 2)

    g = fx.FREQ_3x6
    fbm = orc.fbm_from_genotypes(g)
    assert np.array_equal(orc.gt_pi_diploid(fbm), pi_by_hand(g))
    sub = g[np.ix_([0, 2], [0, 1, 3, 5])]
    assert np.array_equal(orc.gt_pi_diploid(fbm, [1, 3], [1, 2, 4, 6]), pi_by_hand(sub), equal_nan=True)
    # grouped: column g of the grouped result is the ungrouped result of that group's individuals (:98-101)
    gid = fx.FST_GROUPS_3
    f7 = orc.fbm_from_genotypes(fx.FST_7x6)
    with np.errstate(invalid="ignore", divide="ignore"):
        grp = orc.gt_grouped_pi_diploid(f7, None, None, gid, 3)["pi"]
        for k in range(3):
            rows = np.where(gid == k)[0] + 1
            assert np.array_equal(grp[:, k], orc.gt_pi_diploid(f7, rows, None), equal_nan=True)
            assert np.array_equal(grp[:, k], pi_by_hand(fx.FST_7x6[gid == k]), equal_nan=True)


def test_indiv_het_obs_by_hand():
    # tests/testthat/test_indiv_het_obs.R:35-51 (rowMeans(x == 1, na.rm = TRUE); counts het_n / na_n) and :56-95
    g = fx.FREQ_3x6
    fbm = orc.fbm_from_genotypes(g)
    by_hand = np.array([np.nanmean(np.where(np.isnan(r), np.nan, (r == 1).astype(float))) for r in g])
    assert np.array_equal(orc.indiv_het_obs(fbm), by_hand)
    counts = orc.indiv_het_obs(fbm, as_counts=True)
    assert np.array_equal(counts[:, 0], np.nansum(g == 1, axis=1)) and np.array_equal(counts[:, 1], np.isnan(g).sum(axis=1))
    homo = np.array([[2, 2, 0, 0, 2, 0]] * 3, dtype=float)
    assert np.array_equal(orc.indiv_het_obs(orc.fbm_from_genotypes(homo)), np.zeros(3))


def test_filter_high_relatedness_families():
    # tests/testthat/test_filter_high_relatedness.R:8-60: KING of the families data, threshold 0.2.  The test's
    # comments name the two pairs over it (individuals 11-12 and 9-10; the KING golden file test_king.kin0 holds their
    # coefficients, 0.2248 and 0.2741); filtering the kept sub-matrix again keeps everybody.
    king = orc.snp_king(fx.families_fbm())
    gold = fx.king_kin0_matrix()
    assert round(king[10, 11], 4) == gold[10, 11] == 0.2248 and round(king[8, 9], 4) == gold[8, 9] == 0.2741
    off = king.copy()
    np.fill_diagonal(off, 0)
    assert sorted(map(tuple, np.argwhere(np.triu(off > 0.2)))) == [(8, 9), (10, 11)]
    passed, removed, keep = orc.filter_high_relatedness(king, 0.2)
    assert len(removed) == 2 and keep.sum() == 10
    assert len({9, 10} & set(removed.tolist())) == 1 and len({11, 12} & set(removed.tolist())) == 1
    sub = king[np.ix_(passed - 1, passed - 1)]
    assert orc.filter_high_relatedness(sub, 0.2)[2].all()
    assert orc.filter_high_relatedness(np.array([[0.5]]), 0.2)[2].all()      # a single individual passes (:46-52)
    assert orc.filter_high_relatedness(king, 0.6)[2].all()                   # nothing over the threshold


def test_r_mean_two_pass():
    rng = np.random.default_rng(0)
    x = rng.random(1000)
    assert orc.r_mean(x) == pytest.approx(np.mean(x), rel=1e-15)
    assert np.isnan(orc.r_mean([np.nan]))
    assert orc.r_mean([1.0, np.nan, 3.0]) == 2.0


def _slow_r_fst(pf, p1, p2, method):
    """The reference's own vectorised R implementations (data-raw/reference_r_implementations/pairwise_pop_fst.R:
    Hudson :99-117, Nei87 :179-215, WC84 :277-313), restated with numpy: a second, independently written form of the
    three estimators in the reference tree.  -> (by-locus Fst, total)"""
    fa, fr, n, ho = pf["freq_alt"], pf["freq_ref"], pf["n"], pf["het_obs"]
    cols = [p1, p2]
    with np.errstate(invalid="ignore", divide="ignore"):
        if method == "Hudson":
            num = (fa[:, p1] - fa[:, p2]) ** 2 - fa[:, p1] * fr[:, p1] / (n[:, p1] - 1) - fa[:, p2] * fr[:, p2] / (n[:, p2] - 1)
            den = fa[:, p1] * fr[:, p2] + fa[:, p2] * fr[:, p1]
            return num / den, np.nanmean(num) / np.nanmean(den)
        if method == "Nei87":
            nn = n[:, cols] / 2
            sHo = ho[:, cols]
            mHo = np.nanmean(sHo, axis=1)
            sp2 = fa[:, cols] ** 2 + fr[:, cols] ** 2
            np_ = (~np.isnan(nn)).sum(axis=1)
            mn = np_ / np.nansum(1 / nn, axis=1)
            msp2 = np.nanmean(sp2, axis=1)
            mp2 = fa[:, cols].mean(axis=1) ** 2 + fr[:, cols].mean(axis=1) ** 2
            mHs = mn / (mn - 1) * (1 - msp2 - mHo / 2 / mn)
            Ht = 1 - mp2 + mHs / mn / np_ - mHo / 2 / mn / np_
            Dst = Ht - mHs
            Dstp = np_ / (np_ - 1) * Dst
            Htp = mHs + Dstp
            return Dstp / Htp, np.mean(Dstp) / np.mean(Htp)
        r = 2
        n_ind = n[:, cols] / 2
        n_total = n_ind.sum(axis=1)
        n_bar = n_ind.mean(axis=1)
        n_c = (n_total - (n_ind ** 2).sum(axis=1) / n_total) / (r - 1)
        p = fa[:, cols]
        p_bar = (p * n_ind).sum(axis=1) / n_total
        s2 = ((p - p_bar[:, None]) ** 2 * n_ind).sum(axis=1) / n_bar / (r - 1)
        h_bar = (ho[:, cols] * n_ind).sum(axis=1) / n_total
        a = n_bar / n_c * (s2 - 1 / (n_bar - 1) * (p_bar * (1 - p_bar) - (r - 1) / r * s2 - h_bar / 4))
        b = n_bar / (n_bar - 1) * (p_bar * (1 - p_bar) - (r - 1) / r * s2 - (2 * n_bar - 1) / (4 * n_bar) * h_bar)
        c = h_bar / 2
        return a / (a + b + c), np.nanmean(a) / np.nanmean(a + b + c)


@pytest.mark.parametrize("method", ["Hudson", "Nei87", "WC84"])
def test_fst_loops_against_the_references_vectorised_r_versions(method):
    # The reference tree holds every estimator twice: the C++ loops the package calls (src/pairwise_fst_*_loop.cpp,
    # what the oracle's C restates) and the vectorised R versions they replaced (data-raw/reference_r_implementations/).
    # Two restatements written from two different sources must agree; for Nei87 this is the only cross-check the
    # reference offers without hierfstat.
    n, m, G = 90, 400, 4
    fbm = orc.synth_fbm(91, n, m, npop=G, miss=0.03)
    gid = (np.arange(n) % G).astype(np.int32)
    with np.errstate(invalid="ignore", divide="ignore"):
        pf = orc.grouped_summaries_dip_pseudo_cpp(fbm, None, None, gid, G, np.full(n, 2.0))
        got = orc.pairwise_pop_fst(fbm, None, None, gid, G, method=method, by_locus=True)
    pairs = orc.combn2(G)
    for c in range(pairs.shape[1]):
        loc, tot = _slow_r_fst(pf, pairs[0, c] - 1, pairs[1, c] - 1, method)
        # Nei87's Dst = Ht - mHs cancels: the two operation orders differ by ~1e-14 absolute on by-locus values
        assert np.allclose(got["fst_locus"][:, c], loc, rtol=1e-11, atol=1e-12, equal_nan=True), (method, c)
        assert got["fst_tot"][c] == pytest.approx(tot, rel=1e-12), (method, c)
