"""The chunk driver inside the library (tpg_stream_run, include/tpg.h "streamed whole analyses"): a genotype store that
stays on the host is swept in blocks of loci -- the reference's own block loop (R/snp_ibs.R:59-82,
R/loci_alt_freq.R:351-359, big_SVD's two sweeps behind R/gt_pca_partialSVD.R:82-89) -- under a budget on the HBM its bytes
and views may take, and must give what the resident entry points give on the whole panel: integer counts bit for bit
(so IBS / KING / allele sharing identical), per-locus outputs identical, Fst sums to 1e-12 (block order), the PCA to
1e-10 (d) / 1e-8 (u, v).  The resident entry points themselves are checked against the oracle elsewhere
(tests/test_gpu_parity.py, tests/test_gpu_oracle_at_scale.py); the small cases here are checked against the oracle too."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

G = 7


@pytest.fixture(scope="module")
def tpg():
    import tidypopgen_amd as t

    t.default_context()
    return t


def _aligned(a, b):
    """columns of b with the signs of a"""
    return b * np.sign((a * b).sum(axis=0))


def _resident(tpg, X, rows, cols, gid, k, fst=("Hudson", "WC84"), by_locus=False):
    out = {}
    v = tpg.View(X, rows, cols, code256=None)
    pw = tpg.Pairwise(X.ctx, v.n)
    pw.accumulate(v)
    out.update(pw.epilogues(("ibs", "king", "allele_sharing", "grm"), m=v.m))
    v012 = tpg.View(X, rows, cols, code256=tpg.CODE_012)
    out["alt_freq"] = tpg.alt_freq_dip_pseudo_cpp(v012, None, False)
    out["grouped_alt_freq"] = tpg.grouped_alt_freq_dip_pseudo_cpp(v012, gid, G, None, False)
    out["grouped_missingness"] = tpg.grouped_missingness_cpp(v012, gid, G)
    out["loci_counts"] = tpg.loci_counts(v012)
    out["fst_tot"], out["fst_locus"] = {}, {}
    for method in fst:
        r = tpg.pairwise_pop_fst(X, rows, cols, gid, G, method=method, by_locus=by_locus)
        out["fst_tot"][method] = r["fst_tot"]
        if by_locus:
            out["fst_locus"][method] = r["fst_locus"]
    if k:
        out["pca"] = tpg.gt_pca_partialSVD(X, rows, cols, k=k)
    return out


def _compare(s, r, k, by_locus=False, pca_tol_u=1e-8, pca_tol_d=1e-10):
    for name in ("ibs", "king", "allele_sharing"):
        assert np.array_equal(s[name], r[name], equal_nan=True), name
    assert np.allclose(s["grm"], r["grm"], rtol=1e-13, atol=1e-14, equal_nan=True)
    for name in ("alt_freq", "grouped_alt_freq", "grouped_missingness", "loci_counts"):
        assert np.array_equal(s[name], r[name], equal_nan=True), name
    for method, tot in r["fst_tot"].items():
        assert np.allclose(s["fst_tot"][method], tot, rtol=1e-12, atol=0, equal_nan=True), method
        if by_locus:
            assert np.array_equal(s["fst_locus"][method], r["fst_locus"][method], equal_nan=True), method
    if k:
        p = r["pca"]
        assert np.array_equal(s["center"], p["center"]) and np.array_equal(s["scale"], p["scale"])
        assert s["square_frobenius"] == pytest.approx(p["square_frobenius"], rel=1e-12)
        # Panels this short take the digit-split Gram kernel on both routes (the class path does not pay): the resident call
        # with per-locus weights within 2^-23 (its documented 1e-7), and so do the blocks of a run without a budget; under a
        # budget the streamed blocks keep eight bits more.  So the budgeted result is held to the FP64 oracle (numpy eigh of
        # the FP64 Gram matrix) at 1e-10 / 1e-8, and to the resident one at the resident one's own bound.
        assert np.allclose(s["d"], p["d"], rtol=1e-7, atol=0)
        assert np.abs(_aligned(p["u"], s["u"]) - p["u"]).max() <= 1e-6
        o = r.get("pca_oracle")
        if o is not None:
            assert np.allclose(s["d"], o["d"], rtol=pca_tol_d, atol=0), s["d"] / o["d"] - 1
            assert np.abs(_aligned(o["u"], s["u"]) - o["u"]).max() <= pca_tol_u
            assert np.abs(_aligned(o["v"], s["v"]) - o["v"]).max() <= pca_tol_u


@pytest.mark.parametrize("budget", [0, 2 << 20, 600 << 10])
@pytest.mark.parametrize("n,m", [(200, 3000), (333, 5001)])
def test_stream_equals_resident_and_oracle(tpg, n, m, budget):
    """host byte store, three budgets (no bound: 8 blocks, views kept; 2 MiB; 600 KiB: blocks of 128 loci and, for the larger
    panel, a second sweep for the loadings), everything the job can ask for at once"""
    fbm = orc.synth_fbm(41, n, m, npop=G, miss=0.03, imputed_bytes=True)
    gid = (np.arange(n) % G).astype(np.int32)
    k = 5
    X = tpg.FBM.from_numpy(fbm)
    r = _resident(tpg, X, None, None, gid, k, by_locus=True)
    r["pca_oracle"] = orc.gt_pca_partialSVD(fbm, None, None, k=k)
    st = tpg.Stream.from_numpy(fbm, budget_bytes=budget)
    s = st.run(pairwise=("ibs", "king", "allele_sharing", "grm"), groupIds=gid, ngroups=G, alt_freq=True, grouped_alt_freq=True,
               grouped_missingness=True, loci_counts=True, fst=("Hudson", "WC84"), fst_by_locus=True, k=k)
    _compare(s, r, k, by_locus=True, pca_tol_d=1e-10 if budget else 1e-8, pca_tol_u=1e-8 if budget else 1e-6)
    rep = s["report"]
    # what crossed PCIe upwards: the store's bytes once (two code tables are wanted: raw + imputed); a second sweep for the
    # loadings wants ONE table and takes the store as 2 bits per genotype, packed on the host (csrc/host/host_bedpack.h)
    assert rep["blocks"] == -(-m // rep["block_loci"])
    assert rep["bytes_up"] == n * m + (rep["sweeps"] - 1) * m * ((n + 3) // 4)
    if budget:
        assert rep["planned_bytes"] <= budget and rep["blocks"] > 1
    assert rep["sweeps"] == (1 if rep["views_kept"] else 2)
    if budget == 600 << 10:
        assert rep["block_loci"] == 128 and (rep["sweeps"] == 2 or n == 200)
    else:
        assert rep["views_kept"]
    # ... and against the oracle (the reference's statements): counts-based outputs identical
    assert np.array_equal(s["ibs"], orc.snp_ibs(fbm), equal_nan=True)
    assert np.array_equal(s["king"], orc.snp_king(fbm), equal_nan=True)
    assert np.array_equal(s["alt_freq"], orc.alt_freq_dip_pseudo_cpp(fbm, None, None, np.full(n, 2.0)), equal_nan=True)
    assert np.array_equal(s["grouped_alt_freq"], orc.grouped_alt_freq_dip_pseudo_cpp(fbm, None, None, gid, G, np.full(n, 2.0)),
                          equal_nan=True)
    for method in ("Hudson", "WC84"):
        o = orc.pairwise_pop_fst(fbm, None, None, gid, G, method=method)["fst_tot"]
        assert np.allclose(s["fst_tot"][method], o, rtol=1e-10, atol=0)
    st.close()


def test_stream_subsets_scattered_columns_and_single_requests(tpg):
    """rowInd / colInd: a row subset in reverse order, columns scattered (gathered on the host block by block), ascending with
    gaps, and a contiguous run; jobs that ask for one thing only"""
    n, m = 260, 4000
    fbm = orc.synth_fbm(43, n, m, npop=G, miss=0.05, imputed_bytes=True)
    X = tpg.FBM.from_numpy(fbm)
    rng = np.random.default_rng(1)
    rows = np.arange(n, 0, -2).astype(np.int32)
    gid = (np.arange(len(rows)) % G).astype(np.int32)
    for cols in ((rng.permutation(m)[:1500] + 1).astype(np.int32), np.arange(1, m + 1, 3).astype(np.int32),
                 np.arange(701, 2750).astype(np.int32)):
        r = _resident(tpg, X, rows, cols, gid, 4)
        r["pca_oracle"] = orc.gt_pca_partialSVD(fbm, rows, cols, k=4)
        st = tpg.Stream.from_numpy(fbm, budget_bytes=1 << 20)
        s = st.run(rows, cols, pairwise=("ibs", "king", "allele_sharing", "grm"), groupIds=gid, ngroups=G, alt_freq=True,
                   grouped_alt_freq=True, grouped_missingness=True, loci_counts=True, fst=("Hudson", "WC84"), k=4)
        _compare(s, r, 4)
        assert s["report"]["blocks"] > 2
        # one analysis at a time: only its products / views / outputs
        one = st.run(rows, cols, pairwise=("king",))
        assert set(one) == {"king", "report"} and np.array_equal(one["king"], r["king"], equal_nan=True)
        # one code table: a contiguous colInd goes up as 2 bits per genotype (all n rows of the store: rowInd is applied on the
        # device), a scattered one is gathered on the host and goes up as bytes
        contiguous = bool(np.all(np.diff(cols) == 1))
        assert one["report"]["bytes_up"] == (len(cols) * ((n + 3) // 4) if contiguous else len(cols) * n)
        one = st.run(rows, cols, pairwise=("grm",))
        assert np.allclose(one["grm"], r["grm"], rtol=1e-13, atol=1e-14)
        one = st.run(rows, cols, alt_freq=True)
        assert np.array_equal(one["alt_freq"], r["alt_freq"], equal_nan=True)
        one = st.run(rows, cols, groupIds=gid, ngroups=G, fst=("Nei87",))
        assert np.allclose(one["fst_tot"]["Nei87"], tpg.pairwise_pop_fst(X, rows, cols, gid, G, method="Nei87")["fst_tot"], rtol=1e-12)
        one = st.run(rows, cols, k=4, total_var=False)
        assert np.allclose(one["d"], r["pca_oracle"]["d"], rtol=1e-10) and "square_frobenius" not in one
        st.close()


def test_stream_bed_and_synth_sources(tpg, tmp_path):
    """a PLINK .bed as the store (file and payload; contiguous blocks of SNPs and gathered ones) and the synthetic store
    generated block by block on the device"""
    from tests.test_gpu_parity import _bed_file

    n, m = 131, 3001  # n not a multiple of 4: padding bits in every SNP's last byte
    fbm = orc.synth_fbm(47, n, m, npop=G, miss=0.04)
    path = str(tmp_path / "s.bed")
    _bed_file(fbm, path)
    gid = (np.arange(n) % G).astype(np.int32)
    X = tpg.FBM.from_numpy(fbm)
    code_imp = np.array([0, 1, 2, 0] + [np.nan] * 252)  # "missing imputed as 0" (the PCA may not see a missing value)
    for cols in (None, (np.random.default_rng(2).permutation(m)[:1000] + 1).astype(np.int32)):
        r = _resident(tpg, X, None, cols, gid, 0)
        rp = tpg.gt_pca_partialSVD(X, None, cols, k=3, code256=code_imp)
        for st in (tpg.Stream.open_bed(path, n, m, budget_bytes=512 << 10),
                   tpg.Stream.from_bed_payload(np.fromfile(path, dtype=np.uint8)[3:], n, m, budget_bytes=0)):
            s = st.run(None, cols, pairwise=("ibs", "king", "allele_sharing", "grm"), groupIds=gid, ngroups=G, alt_freq=True,
                       grouped_alt_freq=True, grouped_missingness=True, loci_counts=True, fst=("Hudson", "WC84"), k=3,
                       code256_pca=code_imp)
            r["pca"] = rp
            r["pca_oracle"] = orc.gt_pca_partialSVD(fbm, None, cols, k=3, code256=code_imp)
            _compare(s, r, 3, pca_tol_d=1e-10 if st.report["budget_bytes"] else 1e-8, pca_tol_u=1e-8 if st.report["budget_bytes"] else 1e-6)
            st.close()
    # a row subset in reverse order on the .bed store (the generic pack front end), blocks of SNPs under a budget
    rows = np.arange(n, 0, -2).astype(np.int32)
    gid_r = (np.arange(len(rows)) % G).astype(np.int32)
    r = _resident(tpg, X, rows, None, gid_r, 0)
    st = tpg.Stream.open_bed(path, n, m, budget_bytes=512 << 10)
    s = st.run(rows, None, pairwise=("ibs", "king", "allele_sharing", "grm"), groupIds=gid_r, ngroups=G, alt_freq=True,
               grouped_alt_freq=True, grouped_missingness=True, loci_counts=True, fst=("Hudson", "WC84"))
    _compare(s, r, 0)
    assert s["report"]["blocks"] > 1
    st.close()
    with pytest.raises(tpg._lib.TpgError):
        tpg.Stream.open_bed(path, n, m + 1)
    # the synthetic store: equal to the resident synthetic FBM
    n, m = 300, 6000
    Xs = tpg.FBM.synth(5, n, m, npop=G, miss=0.02, imputed_bytes=True)
    gid = (np.arange(n) % G).astype(np.int32)
    r = _resident(tpg, Xs, None, None, gid, 4)
    st = tpg.Stream.synth(5, n, m, npop=G, miss=0.02, imputed_bytes=True, budget_bytes=512 << 10)
    s = st.run(pairwise=("ibs", "king", "allele_sharing", "grm"), groupIds=gid, ngroups=G, alt_freq=True, grouped_alt_freq=True,
               grouped_missingness=True, loci_counts=True, fst=("Hudson", "WC84"), k=4)
    _compare(s, r, 4)
    assert s["report"]["bytes_up"] == 0  # nothing crossed PCIe upwards
    cols = np.arange(1001, 4001, dtype=np.int32)
    s = st.run(None, cols, pairwise=("ibs",))
    assert np.array_equal(s["ibs"], tpg.snp_ibs(Xs, None, cols), equal_nan=True)
    with pytest.raises(tpg._lib.TpgError) as e:
        st.run(None, cols[::2], pairwise=("ibs",))
    assert e.value.code == 3
    st.close()


def test_stream_errors(tpg):
    n, m = 50, 600
    fbm = orc.synth_fbm(3, n, m, npop=3, miss=0.1)
    st = tpg.Stream.from_numpy(fbm, budget_bytes=256 << 10)
    with pytest.raises(tpg._lib.TpgError) as e:
        st.run()  # nothing asked for
    assert e.value.code == 1
    with pytest.raises(tpg._lib.TpgError) as e:
        st.run(k=3)  # missing values in the PCA's view: big_SVD's error, from whichever block meets it first
    assert e.value.code == 4 and "missing values" in str(e.value)
    with pytest.raises(tpg._lib.TpgError) as e:
        st.run(None, np.array([1, m + 1], dtype=np.int32), alt_freq=True)
    assert e.value.code == 1
    with pytest.raises(tpg._lib.TpgError) as e:
        st.run(grouped_alt_freq=True)  # no groups
    assert e.value.code == 1
    bad = np.array([0, 1, 2, 7] + [np.nan] * 252)
    with pytest.raises(tpg._lib.TpgError) as e:
        st.run(code256=bad, alt_freq=True)  # an occurring byte maps outside {0, 1, 2, NA}
    assert e.value.code == 3
    st.close()
    tiny = tpg.Stream.from_numpy(fbm, budget_bytes=1000)
    with pytest.raises(tpg._lib.TpgError) as e:
        tiny.run(alt_freq=True)
    assert e.value.code == 1 and "budget" in str(e.value)
    # the stream is usable after a failed run
    ok = tpg.Stream.from_numpy(fbm, budget_bytes=256 << 10)
    with pytest.raises(tpg._lib.TpgError):
        ok.run(k=3)
    a = ok.run(alt_freq=True)["alt_freq"]
    assert np.array_equal(a, orc.alt_freq_dip_pseudo_cpp(fbm, None, None, np.full(n, 2.0)), equal_nan=True)


def test_stream_config2_under_an_eighth_of_the_panel(tpg):
    """BASELINE config 2's shape, 1 000 x 650 000, with the HBM budget forced to 1 / 8 of the panel's bytes: counts bit-exact,
    Fst 1e-12, PCA d 1e-10 / u 1e-8 against the resident path; the loadings need a second sweep; the device's memory grows
    by no more than budget + additive state"""
    n, m, k = 1000, 650_000, 10
    G51 = 51
    X = tpg.FBM.synth(2, n, m, npop=G51, miss=0.02, imputed_bytes=True)
    fbm = X.to_numpy()
    gid = (np.arange(n) % G51).astype(np.int32)
    budget = n * m // 8
    st = tpg.Stream.from_numpy(fbm, budget_bytes=budget)
    s = st.run(pairwise=("ibs", "king", "allele_sharing", "grm"), groupIds=gid, ngroups=G51, alt_freq=True, grouped_alt_freq=True,
               fst=("Hudson", "WC84"), k=k)
    rep = s["report"]
    assert rep["planned_bytes"] <= budget and rep["blocks"] >= 8 and rep["sweeps"] == 2 and not rep["views_kept"]
    assert rep["peak_device_bytes"] <= budget + rep["state_bytes"] + (64 << 20), rep
    assert rep["bytes_up"] == n * m + m * (n // 4)  # the second sweep: 2 bits per genotype
    v = tpg.View(X, code256=None)
    pw = tpg.Pairwise(X.ctx, n)
    pw.accumulate(v)
    ep = pw.epilogues(("ibs", "king", "allele_sharing", "grm"), m=m)
    for name in ("ibs", "king", "allele_sharing"):
        assert np.array_equal(s[name], ep[name], equal_nan=True), name
    assert np.allclose(s["grm"], ep["grm"], rtol=1e-13, atol=1e-14)
    v012 = tpg.View(X, code256=tpg.CODE_012)
    assert np.array_equal(s["alt_freq"], tpg.alt_freq_dip_pseudo_cpp(v012, None, False), equal_nan=True)
    assert np.array_equal(s["grouped_alt_freq"], tpg.grouped_alt_freq_dip_pseudo_cpp(v012, gid, G51, None, False), equal_nan=True)
    for method in ("Hudson", "WC84"):
        tot = tpg.pairwise_pop_fst(X, None, None, gid, G51, method=method)["fst_tot"]
        assert np.allclose(s["fst_tot"][method], tot, rtol=1e-12, atol=0), method
    p = tpg.gt_pca_partialSVD(X, None, None, k=k)
    assert np.array_equal(s["center"], p["center"]) and np.array_equal(s["scale"], p["scale"])
    assert np.allclose(s["d"], p["d"], rtol=1e-10, atol=0)
    assert np.abs(_aligned(p["u"], s["u"]) - p["u"]).max() <= 1e-8
    assert np.abs(_aligned(p["v"], s["v"]) - p["v"]).max() <= 1e-8
    assert s["square_frobenius"] == pytest.approx(p["square_frobenius"], rel=1e-12)


def test_stream_twenty_gigabytes_in_under_six(tpg):
    """5 000 x 4 000 000 (20 GB of genotype bytes, generated block by block on the device) under a 2-GB budget: the device's
    memory grows by less than 6 GB over the whole run -- pairwise slabs, two Gram matrices and N x N outputs included.
    Checked against the resident results of the same synthetic panel (20 GB in HBM, afterwards) where those exist at this
    size: IBS / KING identical, alt_freq identical, Fst and singular values to block-order rounding."""
    n, m, k, G51 = 5000, 4_000_000, 10, 51
    gid = (np.arange(n) % G51).astype(np.int32)
    ctx = tpg.Context(0)  # a context of its own: an empty pool, so that growth is what the run really takes
    st = tpg.Stream.synth(3, n, m, npop=G51, miss=0.02, imputed_bytes=True, budget_bytes=2 << 30, ctx=ctx)
    s = st.run(pairwise=("ibs", "king"), groupIds=gid, ngroups=G51, alt_freq=True, fst=("Hudson",), k=k)
    rep = s["report"]
    assert rep["peak_device_bytes"] < 6 << 30, rep
    assert rep["planned_bytes"] <= 2 << 30 and rep["blocks"] > 10 and rep["bytes_up"] == 0
    st.close()
    ctx.close()
    X = tpg.FBM.synth(3, n, m, npop=G51, miss=0.02, imputed_bytes=True)
    assert np.array_equal(s["ibs"], tpg.snp_ibs(X), equal_nan=True)
    assert np.array_equal(s["king"], tpg.snp_king(X), equal_nan=True)
    assert np.array_equal(s["alt_freq"][:, 0], tpg.loci_alt_freq(X), equal_nan=True)
    tot = tpg.pairwise_pop_fst(X, None, None, gid, G51, method="Hudson")["fst_tot"]
    assert np.allclose(s["fst_tot"]["Hudson"], tot, rtol=1e-12, atol=0)
    p = tpg.gt_pca_partialSVD(X, None, None, k=k)
    assert np.allclose(s["d"], p["d"], rtol=1e-10, atol=0)
    assert np.abs(_aligned(p["u"], s["u"]) - p["u"]).max() <= 1e-8
    assert np.array_equal(s["center"], p["center"])


def test_stream_many_individuals(tpg):
    """20 000 individuals x 4 096 loci under a 48-MiB budget: every N x N output is 3.2 GB (past 2^31 bytes: the sizes
    and offsets of the slab reads, the epilogues and the downloads), the PCA takes the Gram route at N = 20 000.
    Against the resident entry points on the same panel, one output at a time (each is 3.2 GB on the host)."""
    n, m, k, G51 = 20_000, 4096, 4, 51
    X = tpg.FBM.synth(5, n, m, npop=G51, miss=0.03, imputed_bytes=True)
    fbm = X.to_numpy()
    gid = (np.arange(n) % G51).astype(np.int32)
    st = tpg.Stream.from_numpy(fbm, budget_bytes=48 << 20)
    s = st.run(pairwise=("ibs", "king", "grm"), groupIds=gid, ngroups=G51, grouped_alt_freq=True, fst=("Hudson",), k=k)
    rep = s["report"]
    assert rep["blocks"] >= 3 and rep["planned_bytes"] <= 48 << 20, rep
    st.close()
    for name, fn in (("ibs", tpg.snp_ibs), ("king", tpg.snp_king)):
        r = fn(X)
        assert np.array_equal(s[name], r, equal_nan=True), name
        del r
        s.pop(name)
    pw = tpg.Pairwise(X.ctx, n)
    pw.accumulate(tpg.View(X, code256=None), products=tpg.PW_FOR_AS)  # GRM wants V and D
    r = pw.epilogues(("grm",), m=m)["grm"]
    assert np.allclose(s["grm"], r, rtol=1e-13, atol=1e-14)
    del r, pw
    s.pop("grm")
    v012 = tpg.View(X, code256=tpg.CODE_012)
    assert np.array_equal(s["grouped_alt_freq"], tpg.grouped_alt_freq_dip_pseudo_cpp(v012, gid, G51, None, False), equal_nan=True)
    tot = tpg.pairwise_pop_fst(X, None, None, gid, G51, method="Hudson")["fst_tot"]
    assert np.allclose(s["fst_tot"]["Hudson"], tot, rtol=1e-12, atol=0)
    # a panel this short takes the digit-split Gram kernel on both routes: the resident call at its documented 1e-7, the
    # budgeted blocks with eight bits more -- held to FP64 through the SMALL side (eigenvalues of Z'Z, 4 096 x 4 096)
    p = tpg.gt_pca_partialSVD(X, None, None, k=k)
    assert np.array_equal(s["center"], p["center"]) and np.array_equal(s["scale"], p["scale"])
    assert np.allclose(s["d"], p["d"], rtol=1e-7, atol=0)
    assert np.abs(_aligned(p["u"], s["u"]) - p["u"]).max() <= 1e-6
    Z = (np.asarray(tpg.CODE_IMPUTE_PRED)[fbm] - s["center"]) / s["scale"]
    w, W = np.linalg.eigh(Z.T @ Z)
    d64, v64 = np.sqrt(w[::-1][:k]), W[:, ::-1][:, :k]
    assert np.allclose(s["d"], d64, rtol=1e-10, atol=0), s["d"] / d64 - 1
    assert np.abs(_aligned(v64, s["v"]) - v64).max() <= 1e-8
    u64 = Z @ v64 / d64
    assert np.abs(_aligned(u64, s["u"]) - u64).max() <= 1e-8


@pytest.mark.parametrize("seed", range(12))
def test_stream_fuzz(tpg, seed):
    """random shapes around the tile / block edges (m below one block, one locus past a block, n = 1 ... 300), random budgets,
    random subsets, random choice of what is asked for: streamed == resident"""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 31, 33, 64, 127, 129, 200, 300]))
    m = int(rng.choice([1, 5, 127, 128, 129, 255, 257, 1000, 1025, 2048, 3333]))
    Gf = int(min(n, rng.integers(1, 6)))
    fbm = orc.synth_fbm(500 + seed, n, m, npop=Gf, miss=float(rng.choice([0.0, 0.05, 0.4])), imputed_bytes=True)
    X = tpg.FBM.from_numpy(fbm)
    rows = None if rng.random() < 0.5 or n < 3 else (rng.permutation(n)[: max(2, n // 2)] + 1).astype(np.int32)
    kind = rng.integers(0, 3)
    cols = None if kind == 0 or m < 4 else ((rng.permutation(m)[: max(2, m // 2)] + 1).astype(np.int32) if kind == 1 else
                                            np.arange(m // 4 + 1, m - m // 5 + 1, dtype=np.int32))
    nn = n if rows is None else len(rows)
    mm = m if cols is None else len(cols)
    gid = (np.arange(nn) % Gf).astype(np.int32)
    # a budget between "one 128-locus block" and "everything": per-locus bytes are a few nn + a few hundred
    per = 8 * max(nn, 128) + 4000
    budget = int(rng.choice([0, 140 * per, 400 * per, 4000 * per]))
    st = tpg.Stream.from_numpy(fbm, budget_bytes=budget)
    want_pw = [w for w in ("ibs", "king", "allele_sharing", "grm") if rng.random() < 0.6]
    want_fst = Gf >= 2 and rng.random() < 0.7
    s = st.run(rows, cols, pairwise=tuple(want_pw), groupIds=gid, ngroups=Gf, alt_freq=True, grouped_alt_freq=bool(rng.random() < 0.7),
               grouped_missingness=bool(rng.random() < 0.5), loci_counts=True, fst=("Hudson", "WC84") if want_fst else (),
               fst_by_locus=bool(rng.random() < 0.5))
    v = tpg.View(X, rows, cols, code256=None)
    if want_pw:
        pw = tpg.Pairwise(X.ctx, v.n)
        pw.accumulate(v)
        ep = pw.epilogues(tuple(want_pw), m=mm)
        for name in want_pw:
            if name == "grm":
                assert np.allclose(s[name], ep[name], rtol=1e-13, atol=1e-14, equal_nan=True)
            else:
                assert np.array_equal(s[name], ep[name], equal_nan=True), name
    v012 = tpg.View(X, rows, cols, code256=tpg.CODE_012)
    assert np.array_equal(s["alt_freq"], tpg.alt_freq_dip_pseudo_cpp(v012, None, False), equal_nan=True)
    assert np.array_equal(s["loci_counts"], tpg.loci_counts(v012))
    if "grouped_alt_freq" in s:
        assert np.array_equal(s["grouped_alt_freq"], tpg.grouped_alt_freq_dip_pseudo_cpp(v012, gid, Gf, None, False), equal_nan=True)
    if "grouped_missingness" in s:
        assert np.array_equal(s["grouped_missingness"], tpg.grouped_missingness_cpp(v012, gid, Gf))
    if want_fst:
        for method in ("Hudson", "WC84"):
            r = tpg.pairwise_pop_fst(X, rows, cols, gid, Gf, method=method, by_locus="fst_locus" in s)
            assert np.allclose(s["fst_tot"][method], r["fst_tot"], rtol=1e-12, atol=0, equal_nan=True), method
            if "fst_locus" in s:
                assert np.array_equal(s["fst_locus"][method], r["fst_locus"], equal_nan=True), method
    rep = s["report"]
    assert rep["blocks"] == -(-mm // rep["block_loci"]) and (not budget or rep["planned_bytes"] <= budget)
    st.close()


@pytest.mark.parametrize("bedpack", ["1", "0"])
def test_stream_one_table_goes_up_as_two_bits(tpg, monkeypatch, bedpack):
    """A streamed run that needs ONE code table packs the store's bytes to 2 bits per genotype on the host (the layout of a
    .bed payload) and the device packs its views with the .bed front end: n not a multiple of 4 (per-column packing, padding bits),
    n a multiple of 4 (a block is one contiguous piece), bytes 4 .. 6 of an imputed store through both tables, a byte >= 16 in
    ONE block (that block goes as bytes), and TPG_STREAM_BEDPACK=0 as the A/B -- same results, different bytes_up."""
    monkeypatch.setenv("TPG_STREAM_BEDPACK", bedpack)
    for n, m in ((131, 3001), (256, 4096)):
        fbm = orc.synth_fbm(61, n, m, npop=G, miss=0.05, imputed_bytes=True)
        gid = (np.arange(n) % G).astype(np.int32)
        X = tpg.FBM.from_numpy(fbm)
        st = tpg.Stream.from_numpy(fbm, budget_bytes=1 << 20)
        packed = m * ((n + 3) // 4)
        s = st.run(pairwise=("ibs", "king", "allele_sharing", "grm"))  # raw table
        assert s["report"]["bytes_up"] == (packed if bedpack == "1" else n * m) and s["report"]["blocks"] > 1
        assert np.array_equal(s["ibs"], orc.snp_ibs(fbm), equal_nan=True) and np.array_equal(s["king"], orc.snp_king(fbm), equal_nan=True)
        assert np.array_equal(s["allele_sharing"], orc.snp_allele_sharing(fbm), equal_nan=True)
        s = st.run(code256=tpg.CODE_IMPUTE_PRED, groupIds=gid, ngroups=G, alt_freq=True, grouped_alt_freq=True, loci_counts=True,
                   fst=("Hudson",))  # the imputed table: bytes 4 .. 6 are dosages
        assert s["report"]["bytes_up"] == (packed if bedpack == "1" else n * m)
        vi = tpg.View(X, code256=tpg.CODE_IMPUTE_PRED)
        assert np.array_equal(s["alt_freq"], tpg.alt_freq_dip_pseudo_cpp(vi, None, False), equal_nan=True)
        assert np.array_equal(s["grouped_alt_freq"], tpg.grouped_alt_freq_dip_pseudo_cpp(vi, gid, G, None, False), equal_nan=True)
        assert np.array_equal(s["loci_counts"], tpg.loci_counts(vi)) and s["loci_counts"][:, 3].sum() == 0
        s = st.run(k=4)  # the PCA's table
        o = orc.gt_pca_partialSVD(fbm, None, None, k=4)
        assert np.allclose(s["d"], o["d"], rtol=1e-10) and np.array_equal(s["center"], o["center"])
        assert np.abs(_aligned(o["v"], s["v"]) - o["v"]).max() <= 1e-8
        st.close()
        # a byte no 16-entry table holds, in one block only: that block goes as bytes, the others packed; raw semantics: missing
        odd = fbm.copy(order="F")
        odd[5, 1000] = 200
        st = tpg.Stream.from_numpy(odd, budget_bytes=1 << 20)
        s = st.run(pairwise=("ibs",))
        assert np.array_equal(s["ibs"], orc.snp_ibs(odd), equal_nan=True)
        if bedpack == "1":
            B = s["report"]["block_loci"]
            mb = min(m, (1000 // B + 1) * B) - 1000 // B * B
            assert s["report"]["bytes_up"] == (m - mb) * ((n + 3) // 4) + mb * n
        st.close()


def test_stream_api_from_c(tpg, tmp_path):
    """include/tpg.h's streamed entry points from plain C (tests/host/stream_example.c: gcc, the header, libtpg_hip.so -- no Python
    in the call path): the struct layouts and the calling sequence of INTEGRATION.md 3a, results equal to the Python mirror's"""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "stream_example")
    libdir = os.path.join(root, "tidypopgen_amd")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "host", "stream_example.c"), "-o", exe, "-L", libdir, "-ltpg_hip",
                           f"-Wl,-rpath,{libdir}"])
    n, m, Gc, k = 150, 2600, 4, 3
    fbm = orc.synth_fbm(71, n, m, npop=Gc, miss=0.03, imputed_bytes=True)
    bk = tmp_path / "c.bk"
    fbm.T.tofile(bk)
    budget = 400 << 10
    r = subprocess.run([exe, str(bk), str(n), str(m), str(budget), str(Gc), str(k)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C_OK" in r.stdout, (r.stdout, r.stderr)
    got = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln and ln != "C_OK"}
    gid = (np.arange(n) % Gc).astype(np.int32)
    s = tpg.Stream.open_bk(str(bk), n, m, budget_bytes=budget).run(pairwise=("ibs",), code256=None, groupIds=gid, ngroups=Gc,
                                                                    grouped_alt_freq=True, fst=("Hudson",), k=k)
    assert int(got["blocks"][0]) == s["report"]["blocks"] > 1 and int(got["blocks"][2]) == s["report"]["sweeps"]
    assert float(got["ibs_sum"][0]) == pytest.approx(np.nansum(s["ibs"]), rel=1e-14)
    assert float(got["gaf_sum"][0]) == pytest.approx(np.nansum(s["grouped_alt_freq"]), rel=1e-14)
    assert float(got["fst_sum"][0]) == pytest.approx(np.nansum(s["fst_tot"]["Hudson"]), rel=1e-12)
    assert np.allclose([float(x) for x in got["d"]], s["d"], rtol=1e-12)
    assert float(got["fro"][0]) == pytest.approx(s["square_frobenius"], rel=1e-14)
    assert float(got["center_sum"][0]) == pytest.approx(s["center"].sum(), rel=1e-14)
