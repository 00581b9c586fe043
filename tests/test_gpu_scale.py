"""GPU, BASELINE.json config sizes: properties that pin the results without an O(N^2 M) CPU oracle.
C2 = 1 000 individuals x 650 000 SNPs (pairwise_king + pairwise_grm on one MI355X)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, M, G = 1000, 650_000, 51


@pytest.fixture(scope="module")
def panel():
    import tidypopgen_amd as tpg

    X = tpg.FBM.synth(2, N, M, npop=G, miss=0.02, imputed_bytes=True)
    v = tpg.View(X, code256=None)
    pw = tpg.Pairwise(X.ctx, N)
    pw.accumulate(v)
    return tpg, X, v, pw


def test_pairwise_counts_tie_to_per_locus_counts(panel):
    tpg, X, v, pw = panel
    c = pw.counts()
    cnt = tpg.loci_counts(v).astype(np.int64)  # n0, n1, n2, nNA per locus
    typed_i = np.diag(c["as_den"])             # V_ii = loci typed in i
    assert typed_i.sum() == cnt[:, :3].sum()   # sum over individuals == sum over loci
    assert np.diag(c["as_num"]).sum() == (cnt[:, 0] + cnt[:, 2]).sum()       # D_ii = homozygous loci of i
    assert np.diag(c["n_Aa_i"]).sum() == cnt[:, 1].sum()                     # A_ii = heterozygous loci of i
    assert np.array_equal(np.diag(c["ibs"]), np.diag(c["ibs_valid"]))        # a genotype is IBS 2 with itself
    for k in ("ibs", "ibs_valid", "king_num", "as_num", "as_den"):
        assert np.array_equal(c[k], c[k].T), k
    assert np.all(c["ibs"] <= c["ibs_valid"]) and np.all(c["ibs"] >= 0)
    assert np.all(c["as_den"] <= np.minimum.outer(typed_i, typed_i))
    assert np.all(np.abs(c["as_num"]) <= c["as_den"])
    # N_Aa_i[i, j] counts loci where i is het and j typed: bounded by row het count and column typed count
    assert np.all(c["n_Aa_i"] <= np.diag(c["n_Aa_i"])[:, None]) and np.all(c["n_Aa_i"] <= typed_i[None, :])


def test_block_invariance_and_idempotence_at_scale(panel):
    tpg, X, v, pw = panel
    whole = pw.counts(("ibs", "king_num", "n_Aa_i", "as_num"))
    pw2 = tpg.Pairwise(X.ctx, N)
    edges = [0, 128 * 1000, 128 * 3500, M]
    for a, b in zip(edges, edges[1:]):
        pw2.accumulate(v, a, b)
    parts = pw2.counts(("ibs", "king_num", "n_Aa_i", "as_num"))
    for k in whole:
        assert np.array_equal(whole[k], parts[k]), k
    pw2.zero()
    pw2.accumulate(v)
    again = pw2.counts(("ibs",))
    assert np.array_equal(again["ibs"], whole["ibs"])  # integer atomics: order independent, run-to-run identical


def test_epilogue_semantics_at_scale(panel):
    tpg, X, v, pw = panel
    ep = pw.epilogues()
    c = pw.counts()
    assert np.allclose(np.diag(ep["king"]), 0.5)  # KING diagonal = 0.5 where the individual has a het locus
    assert np.array_equal(ep["ibs"], c["ibs"] / c["ibs_valid"])
    as_ = 0.5 * (1 + c["as_num"] / c["as_den"])
    assert np.array_equal(ep["allele_sharing"], as_)
    off = as_[~np.eye(N, dtype=bool)]
    assert np.allclose(ep["grm"], 2 * (as_ - off.mean()) / (1 - off.mean()), rtol=1e-12, atol=1e-14)


def test_per_locus_and_fst_at_scale(panel):
    tpg, X, v, pw = panel
    vv = tpg.View(X)  # CODE_012: imputed bytes stay missing
    gid = (np.arange(N) % G).astype(np.int32)
    cnt = tpg.loci_counts(vv).astype(np.int64)
    f = tpg.alt_freq_dip_pseudo_cpp(vv, None, as_counts=True)
    assert np.array_equal(f[:, 0], cnt[:, 1] + 2 * cnt[:, 2]) and np.array_equal(f[:, 1], 2 * cnt[:, :3].sum(axis=1))
    ga = tpg.grouped_alt_freq_dip_pseudo_cpp(vv, gid, G, None, as_counts=True)
    assert np.array_equal(ga[:, :G].sum(axis=1), f[:, 0]) and np.array_equal(ga[:, G:].sum(axis=1), f[:, 1])
    gm = tpg.grouped_missingness_cpp(vv, gid, G)
    assert np.array_equal(gm.sum(axis=1), cnt[:, 3])
    # Fst: whole range == ratio of the sums of two halves (the quantity SNP shards exchange); swapping the
    # populations of a pair does not change Hudson / WC84
    import ctypes as C
    from tidypopgen_amd import api

    pairs = np.ascontiguousarray(tpg.combn2(G).T)
    P = pairs.shape[0]
    ploidy = np.full(N, 2.0)
    for method, code in (("Hudson", 0), ("WC84", 2)):
        tot = tpg.pairwise_pop_fst(X, None, None, gid, G, method=method)["fst_tot"]
        sn, sd = np.zeros(P), np.zeros(P)
        tpg._lib.check(tpg._lib.lib.tpg_pairwise_pop_fst_sums(vv.ctx.h, vv.h, api._ptr(gid), C.c_int(G), api._ptr(ploidy),
                                                              C.c_int(code), api._ptr(pairs), C.c_int(P), api._ptr(sn),
                                                              api._ptr(sd)))
        assert np.allclose(sn / sd, tot, rtol=1e-13)
        swapped = tpg.pairwise_pop_fst(X, None, None, gid, G, method=method, pairwise_combn=tpg.combn2(G)[::-1])["fst_tot"]
        assert np.allclose(swapped, tot, rtol=1e-12)
        assert np.all(np.isfinite(tot)) and np.all(tot > -0.01) and np.all(tot < 1)


def test_pca_properties_at_scale(panel):
    tpg, X, v, pw = panel
    vi = tpg.View(X, code256=tpg.CODE_IMPUTE_PRED)
    cnt = tpg.loci_counts(vi).astype(np.int64)
    alt = cnt[:, 1] + 2 * cnt[:, 2]
    cols = (np.where((alt > 0) & (alt < 2 * N))[0] + 1).astype(np.int32)
    k = 10
    r = tpg.gt_pca_partialSVD(X, None, cols, k=k)
    vv = tpg.View(X, None, cols, code256=tpg.CODE_IMPUTE_PRED)
    K = tpg.pca_gram(vv, r["center"], r["scale"])
    assert np.array_equal(K, K.T)
    # trace(ZZ') = ||Z||_F^2; the Gram matrix uses per-locus weights rounded to 2^-24 relative
    assert np.trace(K) == pytest.approx(r["square_frobenius"], rel=1e-7)
    assert np.allclose(K.sum(axis=0), 0, atol=1e-6 * np.abs(K).max())          # centered columns
    u, d = r["u"], r["d"]
    assert np.allclose(u.T @ u, np.eye(k), atol=1e-10)
    assert np.all(np.diff(d) <= 0) and (d ** 2).sum() < r["square_frobenius"]
    res = K @ u - u * d ** 2
    assert np.abs(res).max() <= 1e-9 * d[0] ** 2                                # eigenpairs of the Gram matrix
    lam = np.linalg.eigvalsh(K)[::-1][:k]
    assert np.allclose(d ** 2, lam, rtol=1e-9)
    # v = Z'u/d has orthonormal columns, and X v reproduces the scores u d (projection kernel, a12)
    assert np.allclose(r["v"].T @ r["v"], np.eye(k), atol=1e-8)
    XV, rss = tpg.fbm256_prod_and_rowSumsSq(X, None, cols, r["center"], r["scale"], r["v"], code256=tpg.CODE_IMPUTE_PRED)
    assert np.allclose(XV, u * d, atol=1e-7 * d[0])
    assert np.allclose(rss, np.diag(K), rtol=1e-7)


def test_many_individuals_indexing():
    """N = 12 000 (375 row tiles, 35 000 pairwise units, 1.4 GB of int32 accumulators, 1.2 GB Gram matrix): the tables,
    slabs and epilogues index correctly far from the bench shape.  The full results must contain, as a sub-matrix, the
    results of the same analysis run on a subset of the individuals alone."""
    import tidypopgen_amd as tpg

    n, m, G = 12_000, 2_048, 40
    X = tpg.FBM.synth(11, n, m, npop=G, miss=0.03, imputed_bytes=True)
    rows = np.arange(1, n + 1, 7, dtype=np.int32)
    full = tpg.Pairwise(X.ctx, n)
    full.accumulate(tpg.View(X, code256=None))
    sub = tpg.Pairwise(X.ctx, len(rows))
    sub.accumulate(tpg.View(X, rows, None, code256=None))
    ix = np.ix_(rows - 1, rows - 1)
    for names in (("ibs", "ibs_valid", "king_num"), ("n_Aa_i", "as_num", "as_den")):
        cf, cs = full.counts(names), sub.counts(names)
        for k in cs:
            assert np.array_equal(cf[k][ix], cs[k]), k
        del cf, cs
    which = ("ibs", "king", "allele_sharing")
    ef, es = full.epilogues(which, m=m), sub.epilogues(which, m=m)
    for k in which:
        assert np.array_equal(ef[k][ix], es[k], equal_nan=True), k
    del ef, es, full, sub
    # PCA at this N: orthonormal scores, descending singular values, and orthonormal loadings (v = Z'u / d, so
    # V'V = U'KU / d^2 is the identity only if (u, d^2) are eigenpairs of K = ZZ')
    r = tpg.gt_pca_partialSVD(X, k=5)
    assert np.all(np.diff(r["d"]) < 0) and np.all(r["d"] > 0)
    assert np.allclose(r["u"].T @ r["u"], np.eye(5), atol=1e-9)
    assert np.allclose(r["v"].T @ r["v"], np.eye(5), atol=1e-7)


def test_forty_thousand_individuals_indexing():
    """N = 40 000 (1 250 row tiles, 261 000 pairwise units, 16 GB of int32 slabs, 12.8 GB per N x N double output): 32-bit lane
    offsets, unit tables and epilogue indexing far past anything the bench reaches.  One output matrix is on the host at a
    time.  (a) the full results contain, as a sub-matrix, the results of the same analysis on every 7th individual alone;
    (b) 64 individuals spread over the first, a middle and the last (partial) row tile against the CPU oracle."""
    import tidypopgen_amd as tpg
    from oracle import oracle as orc

    n, m, G = 40_000, 1_024, 40
    X = tpg.FBM.synth(17, n, m, npop=G, miss=0.03)
    rows = np.arange(1, n + 1, 7, dtype=np.int32)
    pick = np.unique(np.concatenate([np.arange(0, 20), np.arange(20_000, 20_022), np.arange(n - 22, n)]))
    fb = X.to_numpy()
    sub_fbm = np.ascontiguousarray(fb[pick])
    assert np.array_equal(sub_fbm, orc.synth_fbm(17, n, m, npop=G, miss=0.03)[pick])
    del fb
    full = tpg.Pairwise(X.ctx, n)
    full.accumulate(tpg.View(X, code256=None))
    sub = tpg.Pairwise(X.ctx, len(rows))
    sub.accumulate(tpg.View(X, rows, None, code256=None))
    small = tpg.Pairwise(X.ctx, len(pick))
    small.accumulate(tpg.View(X, (pick + 1).astype(np.int32), None, code256=None))
    o_ibs = orc.snp_ibs(sub_fbm, type="raw_counts")
    ix, px = np.ix_(rows - 1, rows - 1), np.ix_(pick, pick)
    cs, cp = sub.counts(), small.counts()
    assert np.array_equal(cp["ibs"], o_ibs["ibs"]) and np.array_equal(cp["ibs_valid"], o_ibs["valid_n"])
    for name in ("ibs", "ibs_valid", "king_num", "n_Aa_i", "as_num", "as_den"):
        cf = full.counts((name,))[name]
        assert np.array_equal(cf[ix], cs[name]), name
        assert np.array_equal(cf[px], cp[name]), name
        if name != "n_Aa_i":
            i = np.arange(0, n, 997)
            assert np.array_equal(cf[i, :], cf[:, i].T), name  # both triangles written
        del cf
    del cs, cp
    o_king = orc.snp_king(sub_fbm)
    o_as = orc.snp_allele_sharing(sub_fbm)
    for name, ref in (("king", o_king), ("allele_sharing", o_as)):
        ef = full.epilogues((name,), m=m)[name]
        es = sub.epilogues((name,), m=m)[name]
        assert np.array_equal(ef[ix], es, equal_nan=True), name
        assert np.allclose(ef[px], ref, rtol=1e-12, atol=0, equal_nan=True), name
        del ef, es
    full.free(); sub.free(); small.free()
    # the per-locus and grouped sweeps at this N (157 chunks of 256 individuals per locus)
    gid = (np.arange(n) % G).astype(np.int32)
    v = tpg.View(X)
    cnt = tpg.loci_counts(v)
    fb = X.to_numpy()
    for c in range(3):
        assert np.array_equal(cnt[:, c], (fb == c).sum(axis=0))
    ga = tpg.grouped_alt_freq_dip_pseudo_cpp(v, gid, G, np.full(n, 2.0), True)
    gb = orc.grouped_alt_freq_dip_pseudo_cpp(fb, None, None, gid, G, np.full(n, 2.0), True)
    assert np.array_equal(ga, gb, equal_nan=True)
    del fb
    X.free()


def test_bench_size_properties():
    """BASELINE configs 3-5 (5 000 x 1 000 000, 51 populations): results pinned by properties that need no CPU pass
    over the panel -- additivity over locus blocks (bit-exact for counts, 1e-12 for Fst sums), and the SVD identity
    Z v_j = d_j u_j checked with the independent FP64 sweep kernel (fbm256_prod_and_rowSumsSq)."""
    import tidypopgen_amd as tpg

    n, m, G, k = 5_000, 1_000_000, 51, 20
    X = tpg.FBM.synth(3, n, m, npop=G, miss=0.02, imputed_bytes=True)
    gid = (np.arange(n) % G).astype(np.int32)
    v = tpg.View(X, code256=None)
    # pairwise: one pass == three unequal blocks, and the diagonal ties to the per-locus counts
    whole, parts = tpg.Pairwise(X.ctx, n), tpg.Pairwise(X.ctx, n)
    whole.accumulate(v)
    edges = [0, 128 * 1111, 128 * 5000, m]
    for a, b in zip(edges, edges[1:]):
        parts.accumulate(v, a, b)
    cw, cp = whole.counts(("ibs", "king_num", "n_Aa_i")), parts.counts(("ibs", "king_num", "n_Aa_i"))
    for key in cw:
        assert np.array_equal(cw[key], cp[key]), key
    cnt = tpg.loci_counts(v).astype(np.int64)
    assert np.diag(cw["n_Aa_i"]).sum() == cnt[:, 1].sum()
    assert np.diag(cw["ibs"]).sum() == 2 * cnt[:, :3].sum()
    del cw, cp, whole, parts
    # Fst: the sums over two halves of the loci add up to the sums over all of them
    half = 128 * 3906
    for method in ("Hudson", "WC84"):
        r_all = tpg.pairwise_pop_fst(X, None, None, gid, G, method=method, sums=True)
        r_a = tpg.pairwise_pop_fst(X, None, np.arange(1, half + 1, dtype=np.int32), gid, G, method=method, sums=True)
        r_b = tpg.pairwise_pop_fst(X, None, np.arange(half + 1, m + 1, dtype=np.int32), gid, G, method=method, sums=True)
        for key in ("sum_num", "sum_den"):
            assert np.allclose(r_a[key] + r_b[key], r_all[key], rtol=1e-11, atol=0), (method, key)
        assert np.all(np.isfinite(r_all["fst_tot"])) and np.all(np.abs(r_all["fst_tot"]) < 1)
    # PCA: orthonormal factors and Z v = u d through a different kernel
    alt = cnt[:, 1] + 2 * cnt[:, 2]  # (raw view: imputed bytes read as missing, fine for a polymorphism filter)
    cols = (np.where((alt > 0) & (alt < 2 * cnt[:, :3].sum(axis=1)))[0] + 1).astype(np.int32)
    r = tpg.gt_pca_partialSVD(X, None, cols, k=k)
    assert np.all(np.diff(r["d"]) < 0) and np.all(r["d"] > 0)
    assert np.allclose(r["u"].T @ r["u"], np.eye(k), atol=1e-9)
    assert np.allclose(r["v"].T @ r["v"], np.eye(k), atol=1e-7)
    XV, rss = tpg.fbm256_prod_and_rowSumsSq(X, None, cols, r["center"], r["scale"], r["v"], code256=tpg.CODE_IMPUTE_PRED)
    assert np.allclose(XV, r["u"] * r["d"], rtol=0, atol=1e-6 * r["d"][0])
    assert rss.sum() == pytest.approx(r["square_frobenius"], rel=1e-9)


def test_pair_counts_stay_exact_past_2_pow_24_loci():
    """The cross-products run on the FP4 matrix cores with FP32 accumulators, which hold integers exactly up to 2^24:
    a panel with more loci than that, every locus the same genotype column, so that every pair count is a multiple of
    the number of loci and any lost unit shows.  (A wave-unit's K range is capped at 2^24 loci by the launcher.)"""
    import tidypopgen_amd as tpg

    n, m = 37, (1 << 24) + 128 * 1003 + 5
    col = np.array([0, 1, 2, 3, 1, 1, 2, 0, 3, 2] * 4, dtype=np.uint8)[:n]  # 3 = missing
    Xb = np.empty((n, m), dtype=np.uint8, order="F")
    Xb[:] = col[:, None]
    X = tpg.FBM.from_numpy(Xb)
    del Xb
    v = tpg.View(X, code256=None)
    pw = tpg.Pairwise(X.ctx, n)
    pw.accumulate(v)
    c = pw.counts()
    typed = (col < 3).astype(np.int64)
    d = np.where(col < 3, col.astype(np.int64) - 1, 0)
    h = (col == 1).astype(np.int64)
    V, D, H, A = np.outer(typed, typed), np.outer(d, d), np.outer(h, h), np.outer(h, typed)
    assert np.array_equal(c["as_den"], m * V)
    assert np.array_equal(c["as_num"], m * D)
    assert np.array_equal(c["n_Aa_i"], m * A)
    assert np.array_equal(c["ibs"], m * (V + D + H))
    assert np.array_equal(c["ibs_valid"], 2 * m * V)
    assert np.array_equal(c["king_num"], m * (D - V + A + A.T))
    del c
    # the three product-set kernels (tpg_pairwise_set_kernel: what a stand-alone snp_ibs / snp_king / pairwise_grm runs on)
    # cap and flush their wave-units' K ranges themselves
    for products, names in ((tpg.PW_FOR_AS, ("as_num", "as_den", "ibs_valid")), (tpg.PW_FOR_IBS, ("ibs", "ibs_valid", "as_num")),
                            (tpg.PW_FOR_IBS_ALONE, ("ibs", "ibs_valid", "as_den")),
                            (tpg.PW_FOR_KING, ("king_num", "n_Aa_i", "as_den"))):
        pw.zero()
        pw.accumulate(v, products=products)
        c = pw.counts(names)
        want = dict(as_num=m * D, as_den=m * V, ibs_valid=2 * m * V, ibs=m * (V + D + H), king_num=m * (D - V + A + A.T),
                    n_Aa_i=m * A)
        for k in names:
            assert np.array_equal(c[k], want[k]), (products, k)
