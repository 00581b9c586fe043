"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs,
against the committed golden fixtures, and edge cases the reference tests cover."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import fixtures as fx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tpg():
    import tidypopgen_amd as t

    t.default_context()
    return t


def _X(tpg, g):
    return tpg.FBM.from_numpy(orc.fbm_from_genotypes(g))


# ---------------------------------------------------------------- pack / synth
def test_synth_matches_host_generator(tpg):
    for (seed, n, m, G, imp) in ((3, 50, 200, 7, False), (9, 333, 1000, 51, True)):
        X = tpg.FBM.synth(seed, n, m, npop=G, imputed_bytes=imp)
        assert np.array_equal(X.to_numpy(), orc.synth_fbm(seed, n, m, npop=G, imputed_bytes=imp))
    X = tpg.FBM.synth(3, 50, 80, j0=120, npop=7)
    assert np.array_equal(X.to_numpy(), orc.synth_fbm(3, 50, 200, npop=7)[:, 120:])


@pytest.mark.parametrize("n,m", [(1, 1), (12, 961), (129, 257), (300, 1000)])
def test_pack_roundtrip_and_views(tpg, n, m):
    fbm = orc.synth_fbm(5, n, m, npop=3, miss=0.1)
    X = tpg.FBM.from_numpy(fbm)
    assert np.array_equal(tpg.View(X).unpack(), fbm)  # also cross-checks T against L on the device
    rng = np.random.default_rng(0)
    rows = rng.permutation(n)[: max(1, n // 2)] + 1
    cols = rng.permutation(m)[: max(1, m // 3)] + 1
    assert np.array_equal(tpg.View(X, rows, cols).unpack(), fbm[np.ix_(rows - 1, cols - 1)])
    # imputed code table: bytes 4..6 decode to 0..2, raw view keeps them missing
    fbi = orc.synth_fbm(5, n, m, npop=3, miss=0.1, imputed_bytes=True)
    Xi = tpg.FBM.from_numpy(fbi, code256=tpg.CODE_IMPUTE_PRED)
    assert np.array_equal(tpg.View(Xi).unpack(), np.where(fbi > 3, fbi - 4, fbi))
    assert np.array_equal(tpg.View(Xi, code256=None).unpack(), np.where(fbi > 2, 3, fbi))


@pytest.mark.parametrize("n,m", [(12, 961), (256, 1024), (300, 1000), (1000, 2500)])
def test_view_pair_equals_two_views(tpg, n, m):
    # the raw and the imputed view from one read of the FBM bytes == the two views packed one after the other
    fbi = orc.synth_fbm(7, n, m, npop=3, miss=0.1, imputed_bytes=True)
    X = tpg.FBM.from_numpy(fbi)
    va, vb = tpg.View.pair(X)
    assert np.array_equal(va.unpack(), tpg.View(X, code256=None).unpack())
    assert np.array_equal(vb.unpack(), tpg.View(X, code256=tpg.CODE_IMPUTE_PRED).unpack())
    assert np.array_equal(va.unpack(), np.where(fbi > 2, 3, fbi)) and np.array_equal(vb.unpack(), np.where(fbi > 3, fbi - 4, fbi))
    rng = np.random.default_rng(n)
    rows = (rng.permutation(n)[: max(1, n // 2)] + 1).astype(np.int32)
    cols = (rng.permutation(m)[: max(1, m // 3)] + 1).astype(np.int32)
    for r, c in ((rows, cols), (None, cols)):
        pa, pb = tpg.View.pair(X, r, c, tpg.CODE_012, tpg.CODE_IMPUTE_PRED)
        assert np.array_equal(pa.unpack(), tpg.View(X, r, c, code256=tpg.CODE_012).unpack())
        assert np.array_equal(pb.unpack(), tpg.View(X, r, c, code256=tpg.CODE_IMPUTE_PRED).unpack())
    bad = tpg.CODE_IMPUTE_PRED.copy()
    bad[5] = 0.5  # a value the second table cannot represent, and byte 5 occurs
    with pytest.raises(tpg._lib.TpgError) as e:
        tpg.View.pair(X, None, None, None, bad)
    assert e.value.code == 3


def test_direct_bed_ingest(tpg):
    # the .bed payload is the store; views, counts and IBS equal the FBM route
    import os

    for name, n, m in (("related/families", 12, 961), ("lobster/lobster", 176, 79)):
        path = os.path.join(fx.GOLDEN, name + ".bed")
        fbm = orc.read_bed(path, n, m)
        X = tpg.FBM.open_bed(path, n, m)
        assert np.array_equal(tpg.View(X).unpack(), fbm)
        rows = np.arange(n, 0, -2).astype(np.int32)
        cols = np.arange(1, m + 1, 3).astype(np.int32)
        assert np.array_equal(tpg.View(X, rows, cols).unpack(), fbm[np.ix_(rows - 1, cols - 1)])
        assert np.array_equal(tpg.loci_alt_freq(X, as_counts=True), orc.loci_alt_freq(fbm, as_counts=True))
    fam = tpg.FBM.open_bed(os.path.join(fx.GOLDEN, "related/families.bed"), 12, 961)
    assert np.array_equal(np.round(tpg.snp_ibs(fam), 6), fx.plink_mibs())  # PLINK golden straight from the .bed
    # a larger synthetic .bed (exercises the 128-individual fast path): encode an FBM as .bed and compare
    fbm = orc.synth_fbm(8, 300, 700, npop=3, miss=0.1)
    enc = np.array([3, 2, 0, 1], dtype=np.uint8)[fbm]  # FBM byte 0,1,2,3 -> bed code 11,10,00,01
    pad = np.vstack([enc, np.zeros(((-300) % 4, 700), dtype=np.uint8)])
    bed = (pad[0::4] | (pad[1::4] << 2) | (pad[2::4] << 4) | (pad[3::4] << 6)).T.copy()  # (m, bpl)
    tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), "tpg_test.bed")
    with open(tmp, "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]) + bed.tobytes())
    assert np.array_equal(tpg.View(tpg.FBM.open_bed(tmp, 300, 700)).unpack(), fbm)
    with pytest.raises(tpg._lib.TpgError):
        tpg.FBM.open_bed(tmp, 300, 701)  # file too small
    os.remove(tmp)


def _bed_file(fbm, path):
    """an FBM (bytes 0 / 1 / 2 / 3 = missing) as a PLINK .bed file; odd padding bits of the last byte set to garbage"""
    n, m = fbm.shape
    enc = np.array([3, 2, 0, 1], dtype=np.uint8)[fbm]  # FBM byte 0,1,2,3 -> bed code 11,10,00,01
    pad = np.vstack([enc, np.full(((-n) % 4, m), 2, dtype=np.uint8)])  # (garbage in the unused bit pairs)
    bed = (pad[0::4] | (pad[1::4] << 2) | (pad[2::4] << 4) | (pad[3::4] << 6)).T.copy()  # (m, bytes per SNP)
    with open(path, "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]) + bed.tobytes())


@pytest.mark.parametrize("n,m", [(1, 1), (3, 5), (16, 128), (17, 129), (127, 300), (128, 256), (131, 257), (300, 700), (1030, 2051),
                                 (5000, 1500)])
def test_bed_store_fast_pack_equals_the_byte_store(tpg, tmp_path, monkeypatch, n, m):
    """The .bed payload through the fast pack kernel's own front end (tpg_pack_fast_kernel<NV, true>: one unaligned dword = 16
    individuals of a SNP, 2-bit fields spread and looked up in registers): every layout it writes against the byte FBM of the
    same genotypes -- single views (L + T), pairs (L + T4 | L) through the pairwise kernel and the PCA, per-locus counts left
    behind by the pack, column subsets (the fast path) and row subsets (the generic kernel), n not a multiple of 4 / 16 / 128
    (padding bits of a SNP's last byte are garbage), and the generic kernel as the A/B (TPG_PACK_BED_GENERIC=1)."""
    fbm = orc.synth_fbm(83, n, m, npop=min(n, 4), miss=0.07)
    path = str(tmp_path / "x.bed")
    _bed_file(fbm, path)
    Xb, Xf = tpg.FBM.open_bed(path, n, m), tpg.FBM.from_numpy(fbm)
    cols = (np.random.default_rng(n + m).permutation(m)[: max(1, m // 2)] + 1).astype(np.int32)
    rows = np.arange(n, 0, -2).astype(np.int32)
    for gen in ("0", "1"):
        monkeypatch.setenv("TPG_PACK_BED_GENERIC", gen)
        assert np.array_equal(tpg.View(Xb).unpack(), fbm)
        assert np.array_equal(tpg.View(Xb, None, cols).unpack(), fbm[:, cols - 1])
        assert np.array_equal(tpg.View(Xb, rows, cols).unpack(), fbm[np.ix_(rows - 1, cols - 1)])
        vb, vf = tpg.View(Xb), tpg.View(Xf)
        assert np.array_equal(tpg.loci_counts(vb), tpg.loci_counts(vf))  # (from what the pack leaves per chunk)
        assert np.array_equal(tpg.alt_freq_dip_pseudo_cpp(vb, np.full(n, 2.0), True), orc.alt_freq_dip_pseudo_cpp(fbm, None, None, np.full(n, 2.0), True))
        # a pair: raw view as L + T4 (pairwise kernel), second view as L
        code_imp = np.array([0, 1, 2, 0] + [np.nan] * 252)  # "missing imputed as 0": a table that differs from the raw one
        pa, pb = tpg.View.pair(Xb, None, cols, tpg.CODE_012, code_imp)
        qa, qb = tpg.View.pair(Xf, None, cols, tpg.CODE_012, code_imp)
        assert np.array_equal(pa.unpack(), qa.unpack()) and np.array_equal(pb.unpack(), qb.unpack())
        assert np.array_equal(tpg.loci_counts(pb), tpg.loci_counts(qb))
        pw_b, pw_f = tpg.Pairwise(Xb.ctx, n), tpg.Pairwise(Xf.ctx, n)
        pw_b.accumulate(pa); pw_f.accumulate(qa)
        cb, cf = pw_b.counts(), pw_f.counts()
        for key in cf:
            assert np.array_equal(cb[key], cf[key]), key
        if n >= 16:
            gid = (np.arange(n) % 3).astype(np.int32)
            assert np.array_equal(tpg.grouped_alt_freq_dip_pseudo_cpp(pb, gid, 3, np.full(n, 2.0), True),
                                  tpg.grouped_alt_freq_dip_pseudo_cpp(qb, gid, 3, np.full(n, 2.0), True))
    # the raw-byte view (code256 = NULL) of a .bed store: L + T4 from a single view
    vb, vf = tpg.View(Xb, code256=None), tpg.View(Xf, code256=None)
    pw_b, pw_f = tpg.Pairwise(Xb.ctx, n), tpg.Pairwise(Xf.ctx, n)
    pw_b.accumulate(vb, products=tpg.PW_FOR_AS); pw_f.accumulate(vf, products=tpg.PW_FOR_AS)
    assert np.array_equal(pw_b.counts(("as_num", "as_den"))["as_num"], pw_f.counts(("as_num", "as_den"))["as_num"])


def test_packed_upload_roundtrip(tpg, monkeypatch):
    """Large FBMs cross PCIe as nibbles (host_nibpack.h + tpg_nib_expand_kernel) and arrive as the bytes they were: genotype
    bytes 0 .. 6, odd sizes (an unpacked tail), and a chunk that holds a byte >= 16 (sent as it is)."""
    rng = np.random.default_rng(5)
    for nrow, ncol, spoil in ((8192, 9001, False), (5000, 14001, False), (8200, 9000, True)):
        a = np.asfortranarray(rng.integers(0, 7, size=(nrow, ncol), dtype=np.uint8))
        if spoil:
            a[17, 3] = 200        # in the first chunk
            a[nrow - 1, ncol - 1] = 16  # in the tail
        X = tpg.FBM.from_numpy(a)
        assert np.array_equal(X.to_numpy(), a)
        X.free()
    # the plain path gives the same device bytes
    monkeypatch.setenv("TPG_UPLOAD_PACKED", "1")
    a = np.asfortranarray(rng.integers(0, 4, size=(4096, 20000), dtype=np.uint8))
    X = tpg.FBM.from_numpy(a)
    f = tpg.loci_alt_freq(X, as_counts=True)
    assert np.array_equal(f, orc.loci_alt_freq(a, as_counts=True))
    X.free()


def test_prof_only_times_the_listed_launches(tpg):
    """tpg_prof_only: HIP events only around the launches named (what bench.py's timed region uses), all of them again
    after prof_only(None)."""
    ctx = tpg.default_context()
    X = tpg.FBM.from_numpy(orc.synth_fbm(5, 200, 3000, npop=3))
    try:
        ctx.prof_enable(True)
        ctx.prof_only(["loci_counts"])
        ctx.prof_reset()
        tpg.loci_alt_freq(X)
        only = ctx.prof_dump()
        assert set(only) == {"loci_counts"} and only["loci_counts"][0] >= 1
        ctx.prof_only(None)
        ctx.prof_reset()
        tpg.loci_alt_freq(X)
        every = ctx.prof_dump()
        assert "loci_counts" in every and len(every) > 1
    finally:
        ctx.prof_only(None)
        ctx.prof_enable(False)
        X.free()


def test_small_transfers_through_the_mailbox(tpg, monkeypatch):
    """Small host <-> device transfers go through coherent pinned memory by one-workgroup kernels (runtime.hip:
    tpg_push_small / tpg_fetch_small): every size class, unaligned sizes (which take the copy engine), and more pushes in
    flight than the ring holds (a lap of the ring waits for the lap before)."""
    import ctypes as C

    lib, chk = tpg._lib.lib, tpg._lib.check
    ctx = tpg.default_context()
    rng = np.random.default_rng(11)
    sizes = [4, 8, 60, 8000, 65532, 65536, 65540, 70001, 3, 131072]
    d = ctx.dev_alloc(1 << 20)
    for nb in sizes:
        a = rng.integers(0, 256, size=nb, dtype=np.uint8)
        b = np.zeros(nb, dtype=np.uint8)
        chk(lib.tpg_dev_from_host(ctx.h, d, tpg.api._ptr(a), C.c_size_t(nb)))
        chk(lib.tpg_dev_to_host(ctx.h, tpg.api._ptr(b), d, C.c_size_t(nb)))
        assert np.array_equal(a, b), nb
    # 60 pushes of 40 KiB (the ring holds 12) into distinct places, read back in two ways
    piece = 40 << 10
    big = ctx.dev_alloc(60 * piece)
    src = rng.integers(0, 256, size=(60, piece), dtype=np.uint8)
    for i in range(60):
        chk(lib.tpg_dev_from_host(ctx.h, C.c_void_p(big.value + i * piece), tpg.api._ptr(src[i]), C.c_size_t(piece)))
    back = np.zeros((60, piece), dtype=np.uint8)
    chk(lib.tpg_dev_to_host(ctx.h, tpg.api._ptr(back), big, C.c_size_t(60 * piece)))  # one large copy
    assert np.array_equal(src, back)
    for i in (0, 13, 59):
        one = np.zeros(piece, dtype=np.uint8)
        chk(lib.tpg_dev_to_host(ctx.h, tpg.api._ptr(one), C.c_void_p(big.value + i * piece), C.c_size_t(piece)))
        assert np.array_equal(one, src[i])
    ctx.dev_free(d)
    ctx.dev_free(big)


def test_view_errors(tpg):
    X = tpg.FBM.from_numpy(orc.synth_fbm(1, 10, 20, npop=2))
    with pytest.raises(tpg._lib.TpgError):
        tpg.View(X, [0, 1], None)  # 1-based: 0 is out of range
    with pytest.raises(tpg._lib.TpgError):
        tpg.View(X, None, [21])
    bad = tpg.CODE_012.copy()
    bad[1] = 0.5  # dosage the 2-bit path cannot represent
    with pytest.raises(tpg._lib.TpgError) as e:
        tpg.View(X, None, None, code256=bad)
    assert e.value.code == 3


# ---------------------------------------------------------------- per-locus
def test_alt_freq_reference_cases(tpg):
    # tests/testthat/test_loci_freq.R:1-96
    g = fx.FREQ_3x6
    X = _X(tpg, g)
    freq = np.nansum(g, axis=0) / (np.array([3, 3, 3, 2, 3, 1]) * 2)
    assert np.array_equal(tpg.loci_alt_freq(X), freq)
    counts = tpg.loci_alt_freq(X, as_counts=True)
    assert np.array_equal(counts[:, 0] / counts[:, 1], freq)
    f1 = tpg.loci_alt_freq(X, ind_row=[1, 3], ind_col=[1, 2, 4, 6])
    assert np.array_equal(f1, np.nansum(g[[0, 2]][:, [0, 1, 3, 5]], axis=0) / (np.array([2, 2, 2, 1]) * 2))
    f2 = tpg.loci_alt_freq(X, ind_row=[2, 3], ind_col=[1, 2, 5, 6])
    assert np.isnan(f2[3]) and np.array_equal(f2[:3], np.nansum(g[1:][:, [0, 1, 4]], axis=0) / 4)
    f3 = tpg.loci_alt_freq(_X(tpg, fx.FREQ2_3x6), ind_row=[2, 3])
    assert np.isnan(f3[3]) and np.isnan(f3[5])


def test_missingness_reference_cases(tpg):
    # tests/testthat/test_loci_missingness.R:27-75
    g = fx.FREQ_3x6
    X = _X(tpg, g)
    n_na = np.isnan(g).sum(axis=0)
    assert np.array_equal(tpg.loci_missingness(X, as_counts=True), n_na)
    assert np.array_equal(tpg.loci_missingness(X), n_na / 3)
    assert np.array_equal(tpg.loci_missingness(X, ind_row=[2, 3], as_counts=True), np.isnan(g[1:]).sum(axis=0))


@pytest.mark.parametrize("n,m,G,hap", [(7, 6, 3, False), (200, 1500, 5, False), (333, 2100, 51, False),
                                       (150, 700, 40, True), (1000, 4000, 70, False)])
def test_per_locus_and_grouped_bit_exact(tpg, n, m, G, hap):
    fbm = orc.synth_fbm(21, n, m, npop=G, miss=0.05)
    X = tpg.FBM.from_numpy(fbm)
    rng = np.random.default_rng(1)
    gid = rng.integers(0, G, n).astype(np.int32)
    ploidy = np.full(n, 2.0)
    if hap:
        hapmask = rng.random(n) < 0.3
        ploidy[hapmask] = 1.0
        fbm = fbm.copy()
        sub = fbm[hapmask]
        sub[sub == 1] = 2  # pseudohaploid genotypes are 0 / 2
        fbm[hapmask] = sub
        X = tpg.FBM.from_numpy(fbm)
    v = tpg.View(X)
    for as_counts in (True, False):
        a = tpg.alt_freq_dip_pseudo_cpp(v, ploidy, as_counts)
        b = orc.alt_freq_dip_pseudo_cpp(fbm, None, None, ploidy, as_counts)
        assert np.array_equal(a, b, equal_nan=True)
        with np.errstate(invalid="ignore", divide="ignore"):
            ga = tpg.grouped_alt_freq_dip_pseudo_cpp(v, gid, G, ploidy, as_counts)
            gb = orc.grouped_alt_freq_dip_pseudo_cpp(fbm, None, None, gid, G, ploidy, as_counts)
        assert np.array_equal(ga, gb, equal_nan=True)
    assert np.array_equal(tpg.grouped_missingness_cpp(v, gid, G), orc.grouped_missingness_cpp(fbm, None, None, gid, G))
    with np.errstate(invalid="ignore", divide="ignore"):
        sa = tpg.grouped_summaries_dip_pseudo_cpp(v, gid, G, ploidy)
        sb = orc.grouped_summaries_dip_pseudo_cpp(fbm, None, None, gid, G, ploidy)
    for k in sb:
        assert np.array_equal(sa[k], sb[k], equal_nan=True), k
    cnt = tpg.loci_counts(v)
    for c in range(3):
        assert np.array_equal(cnt[:, c], (fbm == c).sum(axis=0))
    assert np.array_equal(cnt[:, 3], (fbm > 2).sum(axis=0))


def test_next_rows_ind_hetero_and_pi(tpg):
    # gt_ind_hetero, gt_pi_diploid, gt_grouped_pi_diploid -- bit exact
    n, m, G = 333, 2100, 7
    fbm = orc.synth_fbm(23, n, m, npop=G, miss=0.07)
    fbm[:, 5] = 3  # an all-missing locus -> NA
    X = tpg.FBM.from_numpy(fbm)
    rows = np.random.default_rng(5).permutation(n)[:200].astype(np.int32) + 1
    v = tpg.View(X, rows, None)
    gid = (np.arange(200) % G).astype(np.int32)
    assert np.array_equal(tpg.gt_ind_hetero(v), orc.gt_ind_hetero(fbm, rows, None))
    ic = tpg.indiv_counts(v)
    sub = fbm[rows - 1]
    for c in range(3):
        assert np.array_equal(ic[:, c], (sub == c).sum(axis=1))
    assert np.array_equal(tpg.gt_pi_diploid(v), orc.gt_pi_diploid(fbm, rows, None), equal_nan=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        a, b = tpg.gt_grouped_pi_diploid(v, gid, G), orc.gt_grouped_pi_diploid(fbm, rows, None, gid, G)
    assert np.array_equal(a["pi"], b["pi"], equal_nan=True) and np.array_equal(a["n"], b["n"])


# ---------------------------------------------------------------- pairwise individual matrices
def test_ibs_reference_cases_and_plink_golden(tpg):
    # tests/testthat/test_snp_ibs.R:28-105
    X = _X(tpg, fx.IBS_3x6)
    raw = tpg.snp_ibs(X, type="raw_counts")
    assert raw["ibs"][0, 1] == sum([1, 2, 2, 1, 1, 2])
    sub = tpg.snp_ibs(X, ind_row=[1, 3], ind_col=[2, 3, 5, 6], type="raw_counts")
    assert sub["ibs"][0, 1] == sum(np.array([1, 1, 2, 1, 2, 1])[[1, 2, 4, 5]])
    fam = tpg.FBM.from_numpy(fx.families_fbm())
    assert np.array_equal(np.round(tpg.snp_ibs(fam), 6), fx.plink_mibs())
    assert np.array_equal(tpg.snp_ibs(fam, type="adjusted_counts"), orc.snp_ibs(fx.families_fbm(), type="adjusted_counts"))


def test_king_reference_cases_and_golden(tpg):
    # tests/testthat/test_snp_king.R:155-236
    X = _X(tpg, fx.IBS_3x6)
    assert np.array_equal(tpg.snp_king(X), fx.king_r(fx.IBS_3x6), equal_nan=True)
    fam_h = fx.families_fbm()
    Xf = fam_h.astype(float)
    Xf[Xf == 3] = np.nan
    k = tpg.snp_king(tpg.FBM.from_numpy(fam_h))
    assert np.array_equal(k, fx.king_r(Xf), equal_nan=True)
    assert np.nanmax(np.abs(k - fx.king_kin0_matrix())) < 1e-4


def test_allele_sharing_and_grm_reference_cases(tpg):
    # tests/testthat/test_pairwise_allele_sharing.R:29-47, test_pairwise_grm.R:33-41
    X = _X(tpg, fx.AS_3x6)
    assert np.allclose(tpg.snp_allele_sharing(X), fx.matching(fx.AS_3x6), rtol=0, atol=1e-15, equal_nan=True)
    X7 = _X(tpg, fx.FST_7x6)
    M = fx.matching(fx.FST_7x6)
    off = M[~np.eye(7, dtype=bool)]
    assert np.allclose(tpg.pairwise_grm(X7), 2 * (M - off.mean()) / (1 - off.mean()), rtol=0, atol=1e-14)
    # a pair with no locus typed in common -> NA
    g = np.array([[0, np.nan, 1, np.nan], [np.nan, 2, np.nan, 1], [1, 1, 1, 1]], dtype=float)
    a = tpg.snp_allele_sharing(_X(tpg, g))
    assert np.isnan(a[0, 1]) and np.isnan(a[1, 0]) and np.isnan(tpg.snp_ibs(_X(tpg, g))[0, 1])
    assert np.array_equal(a, orc.snp_allele_sharing(orc.fbm_from_genotypes(g)), equal_nan=True)


@pytest.mark.parametrize("variant", [0, 1])  # 1: the five products through the product-set template (A/B form)
@pytest.mark.parametrize("n,m,miss", [(64, 128, 0.0), (65, 129, 0.05), (200, 3000, 0.02), (333, 5001, 0.3),
                                      (500, 8000, 0.02)])
def test_pairwise_counts_bit_exact_and_epilogues(tpg, monkeypatch, n, m, miss, variant):
    monkeypatch.setenv("TPG_PW_VARIANT", str(variant))
    fbm = orc.synth_fbm(31, n, m, npop=9, miss=miss, imputed_bytes=(n == 333))
    X = tpg.FBM.from_numpy(fbm)
    v = tpg.View(X, code256=None)
    pw = tpg.Pairwise(X.ctx, n)
    pw.accumulate(v)
    c = pw.counts()
    K = np.zeros((n, n), order="F"); K2 = np.zeros((n, n), order="F")
    orc.increment_ibs_counts(K, K2, fbm, None, None)
    assert np.array_equal(c["ibs"], K) and np.array_equal(c["ibs_valid"], K2)
    K[:] = 0; K2[:] = 0
    orc.increment_king_numerator(K, K2, fbm, None, None)
    assert np.array_equal(c["king_num"], K) and np.array_equal(c["n_Aa_i"], K2)
    K[:] = 0; K2[:] = 0
    orc.increment_as_counts(K, K2, fbm, None, None)
    assert np.array_equal(c["as_num"], K) and np.array_equal(c["as_den"], K2)
    # epilogues: same formulas, same operation order -> identical doubles
    assert np.array_equal(pw.ibs("proportion"), orc.snp_ibs(fbm), equal_nan=True)
    assert np.array_equal(pw.king(), orc.snp_king(fbm), equal_nan=True)
    assert np.array_equal(pw.allele_sharing(), orc.snp_allele_sharing(fbm), equal_nan=True)
    assert np.allclose(pw.grm(), orc.pairwise_grm(orc.snp_allele_sharing(fbm)), rtol=1e-12, atol=1e-13, equal_nan=True)
    # the fused epilogue gives the same matrices as the individual entry points
    ep = pw.epilogues(ibs_type="adjusted_counts", m=m)
    assert np.array_equal(ep["ibs"], pw.ibs("adjusted_counts", m), equal_nan=True)
    assert np.array_equal(ep["king"], pw.king(), equal_nan=True)
    assert np.array_equal(ep["allele_sharing"], pw.allele_sharing(), equal_nan=True)
    assert np.array_equal(ep["grm"], pw.grm(), equal_nan=True)
    assert np.array_equal(pw.epilogues(which=("grm",))["grm"], ep["grm"], equal_nan=True)


@pytest.mark.parametrize("variant", [0, 1, 2, 13, 14, 16])  # 13 ... 16: operands shared through LDS (3 ... 6 stages)
@pytest.mark.parametrize("which", ["as", "ibs", "ibs1", "king"])
@pytest.mark.parametrize("n,m,miss", [(1, 1, 0.0), (65, 129, 0.05), (130, 700, 0.3), (333, 5001, 0.1), (500, 8000, 0.02)])
def test_pairwise_product_sets_bit_exact(tpg, monkeypatch, n, m, miss, which, variant):
    """tpg_pairwise_accumulate_products: the kernels specialised for {V, D} / {V, D, H} / {V, D, A} (every wave-tile
    variant: TPG_PW_VARIANT) fill the same accumulators as the five-product kernel -- counts bit-exact against the oracle
    (src/snp_as.cpp:64-65, src/snp_ibs.cpp:67-72, src/snp_king.cpp:70-72), epilogues identical -- and the entry points
    refuse outputs whose products were left out."""
    monkeypatch.setenv("TPG_PW_VARIANT", str(variant))
    fbm = orc.synth_fbm(37, n, m, npop=min(n, 7), miss=miss)
    X = tpg.FBM.from_numpy(fbm)
    v = tpg.View(X, code256=None)
    pw = tpg.Pairwise(X.ctx, n)
    assert pw.products() == tpg.PW_ALL
    if which == "ibs1" and variant > 4:
        pytest.skip("the D + H kernel has no workgroup form")
    sets = {"as": tpg.PW_FOR_AS, "ibs": tpg.PW_FOR_IBS, "ibs1": tpg.PW_FOR_IBS_ALONE, "king": tpg.PW_FOR_KING}
    # in two aligned pieces when the panel is long enough: a second launch adds to the same slabs
    cut = (m // 2) // 128 * 128
    if cut:
        pw.accumulate(v, 0, cut, products=sets[which])
        pw.accumulate(v, cut, m, products=sets[which])
    else:
        pw.accumulate(v, products=sets[which])
    assert pw.products() == sets[which]
    K = np.zeros((n, n), order="F"); K2 = np.zeros((n, n), order="F")
    if which == "as":
        c = pw.counts(("as_num", "as_den"))
        orc.increment_as_counts(K, K2, fbm, None, None)
        assert np.array_equal(c["as_num"], K) and np.array_equal(c["as_den"], K2)
        assert np.array_equal(pw.allele_sharing(), orc.snp_allele_sharing(fbm), equal_nan=True)
        assert np.allclose(pw.grm(), orc.pairwise_grm(orc.snp_allele_sharing(fbm)), rtol=1e-12, atol=1e-13, equal_nan=True)
        refused = (lambda: pw.ibs(), lambda: pw.king(), lambda: pw.counts(("ibs",)), lambda: pw.counts(("n_Aa_i",)),
                   lambda: pw.epilogues(which=("king",)))
    elif which == "ibs":
        c = pw.counts(("ibs", "ibs_valid", "as_num", "as_den"))
        orc.increment_ibs_counts(K, K2, fbm, None, None)
        assert np.array_equal(c["ibs"], K) and np.array_equal(c["ibs_valid"], K2)
        assert np.array_equal(pw.ibs("proportion"), orc.snp_ibs(fbm), equal_nan=True)
        assert np.array_equal(pw.allele_sharing(), orc.snp_allele_sharing(fbm), equal_nan=True)  # V, D are in the set
        refused = (lambda: pw.king(), lambda: pw.counts(("king_num",)), lambda: pw.epilogues(which=("ibs", "king")))
    elif which == "ibs1":
        # snp_ibs on its own: D and H in ONE sum (TPG_PW_DH) -- IBS and IBS_valid are all it gives (src/snp_ibs.cpp:67-72)
        c = pw.counts(("ibs", "ibs_valid", "as_den"))
        orc.increment_ibs_counts(K, K2, fbm, None, None)
        assert np.array_equal(c["ibs"], K) and np.array_equal(c["ibs_valid"], K2) and np.array_equal(2 * c["as_den"], K2)
        assert np.array_equal(pw.ibs("proportion"), orc.snp_ibs(fbm), equal_nan=True)
        assert np.array_equal(pw.epilogues(which=("ibs",), ibs_type="adjusted_counts", m=m)["ibs"],
                              orc.snp_ibs(fbm, type="adjusted_counts"), equal_nan=True)
        refused = (lambda: pw.king(), lambda: pw.allele_sharing(), lambda: pw.grm(), lambda: pw.counts(("as_num",)),
                   lambda: pw.counts(("king_num",)), lambda: pw.epilogues(which=("ibs", "grm")))
    else:
        c = pw.counts(("king_num", "n_Aa_i", "as_num", "as_den"))
        orc.increment_king_numerator(K, K2, fbm, None, None)
        assert np.array_equal(c["king_num"], K) and np.array_equal(c["n_Aa_i"], K2)
        assert np.array_equal(pw.king(), orc.snp_king(fbm), equal_nan=True)
        ep = pw.epilogues(which=("king", "grm"))  # BASELINE config 2's pair of analyses from one pass of four products
        assert np.array_equal(ep["king"], orc.snp_king(fbm), equal_nan=True)
        assert np.allclose(ep["grm"], orc.pairwise_grm(orc.snp_allele_sharing(fbm)), rtol=1e-12, atol=1e-13, equal_nan=True)
        refused = (lambda: pw.ibs(), lambda: pw.counts(("ibs",)), lambda: pw.epilogues(which=("ibs",)))
    for f in refused:
        with pytest.raises(tpg._lib.TpgError):
            f()
    # a later pass with all five products does not make the earlier loci's missing products appear
    pw.accumulate(v)
    assert pw.products() == sets[which]
    if which == "ibs1":  # ... but D + H stays a sum whatever kernel adds to it: IBS of both passes = twice the counts
        c2 = pw.counts(("ibs", "ibs_valid"))
        assert np.array_equal(c2["ibs"], 2 * K) and np.array_equal(c2["ibs_valid"], 2 * K2)
        with pytest.raises(tpg._lib.TpgError):
            pw.accumulate(v, products=tpg.PW_DH | tpg.PW_D)  # D + H goes with V only
    pw.zero()
    assert pw.products() == tpg.PW_ALL
    # the R-level entry points take the minimal set themselves
    if variant == 0 and n > 1:
        assert np.array_equal(tpg.snp_king(X), orc.snp_king(fbm), equal_nan=True)
        assert np.array_equal(tpg.snp_ibs(X), orc.snp_ibs(fbm), equal_nan=True)
        assert np.array_equal(tpg.snp_allele_sharing(X), orc.snp_allele_sharing(fbm), equal_nan=True)


def test_pairwise_block_invariance_and_subsets(tpg):
    # block invariance (test_snp_ibs.R:35-36, test_snp_king.R:160-161): accumulate in aligned pieces
    n, m = 150, 1000
    fbm = orc.synth_fbm(41, n, m, npop=4)
    X = tpg.FBM.from_numpy(fbm)
    v = tpg.View(X, code256=None)
    pw = tpg.Pairwise(X.ctx, n)
    pw.accumulate(v)
    whole = pw.counts()
    pw.zero()
    for a, b in ((0, 256), (256, 640), (640, m)):
        pw.accumulate(v, a, b)
    parts = pw.counts()
    for k in whole:
        assert np.array_equal(whole[k], parts[k]), k
    with pytest.raises(tpg._lib.TpgError):
        pw.accumulate(v, 100, 300)  # unaligned start
    # arbitrary row / locus subsets and orders (SURVEY.md §8a "other semantics")
    rows = np.array([5, 3, 149, 77, 10, 11, 12], dtype=np.int32)
    cols = np.random.default_rng(3).permutation(m)[:333].astype(np.int32) + 1
    assert np.array_equal(tpg.snp_king(X, rows, cols), orc.snp_king(fbm, rows, cols), equal_nan=True)
    assert np.array_equal(tpg.snp_ibs(X, rows, cols, type="raw_counts")["ibs"], orc.snp_ibs(fbm, rows, cols, type="raw_counts")["ibs"])


def test_increment_mirrors(tpg):
    fbm = fx.families_fbm()
    n = 12
    lo, up = orc.cut_by_size(961, 300)
    cols = np.arange(1, 962, dtype=np.int32)
    rows = np.arange(1, 13, dtype=np.int32)
    for inc_t, inc_o in ((tpg.increment_ibs_counts, orc.increment_ibs_counts),
                         (tpg.increment_king_numerator, orc.increment_king_numerator),
                         (tpg.increment_as_counts, orc.increment_as_counts)):
        A = np.zeros((n, n), order="F"); B = np.zeros((n, n), order="F")
        Ao = np.zeros((n, n), order="F"); Bo = np.zeros((n, n), order="F")
        for a, b in zip(lo, up):  # unequal blocks 240,240,241,240 as R/snp_ibs.R:59-82 would cut them
            inc_t(A, B, fbm, rows, cols[a - 1:b])  # flush = True: incremented when the call returns, as the reference
            inc_o(Ao, Bo, fbm, rows, cols[a - 1:b])
            assert np.array_equal(A, Ao) and np.array_equal(B, Bo)
    tpg.resident_drop()


@pytest.mark.parametrize("m", [30000, 33000, 70000])
def test_increment_wire_widths(tpg, m):
    """The count matrices of a block come down as 16 bits where the block's bounds allow it (IBS, its valid count and N_Aa up
    to 32 767 / 65 535 loci, the allele-sharing numerator biased by 32 768), as int32 otherwise: blocks on both sides of every
    bound, monomorphic individuals that drive the sums to the bounds, against the oracle."""
    n = 40
    fbm = orc.synth_fbm(77, n, m, npop=2, miss=0.01)
    fbm[0, :] = 2; fbm[1, :] = 2; fbm[2, :] = 0; fbm[3, :] = 1  # IBS(0,1) = 2 m, D(0,2) = -m, A(3,*) large
    rows = np.arange(1, n + 1, dtype=np.int32)
    cols = np.arange(1, m + 1, dtype=np.int32)
    for inc_t, inc_o in ((tpg.increment_ibs_counts, orc.increment_ibs_counts),
                         (tpg.increment_king_numerator, orc.increment_king_numerator),
                         (tpg.increment_as_counts, orc.increment_as_counts)):
        A = np.full((n, n), 5.0, order="F"); B = np.full((n, n), -2.0, order="F")
        Ao = A.copy(order="F"); Bo = B.copy(order="F")
        inc_t(A, B, fbm, rows, cols)
        inc_o(Ao, Bo, fbm, rows, cols)
        assert np.array_equal(A, Ao) and np.array_equal(B, Bo)
    tpg.resident_drop()


def test_increment_mirrors_resident_block_loop(tpg):
    """The R block loop unchanged (R/snp_ibs.R:69-82): FBM uploaded once, accumulators resident, one flush at the end.
    Accumulators that already hold values (a second pass over more loci) are incremented, not overwritten."""
    n, m = 70, 3000
    fbm = orc.synth_fbm(21, n, m, npop=4, miss=0.05)
    rows = (np.random.default_rng(2).permutation(n)[:50] + 1).astype(np.int32)
    cols = np.arange(1, m + 1, dtype=np.int32)
    lo, up = orc.cut_by_size(m, 700)
    mats = {}
    for name, inc_t, inc_o in (("ibs", tpg.increment_ibs_counts, orc.increment_ibs_counts),
                               ("king", tpg.increment_king_numerator, orc.increment_king_numerator),
                               ("as", tpg.increment_as_counts, orc.increment_as_counts)):
        A = np.full((50, 50), 7.0, order="F"); B = np.full((50, 50), -3.0, order="F")  # not zero: += semantics
        Ao = A.copy(order="F"); Bo = B.copy(order="F")
        mats[name] = (A, B, Ao, Bo)
        for a, b in zip(lo, up):
            inc_t(A, B, fbm, rows, cols[a - 1:b], flush=False)
            inc_o(Ao, Bo, fbm, rows, cols[a - 1:b])
        assert np.all(A == 7.0) and np.all(B == -3.0)  # deferred: nothing written before the flush
    tpg.increment_flush()  # all three pending pairs at once
    for name, (A, B, Ao, Bo) in mats.items():
        assert np.array_equal(A, Ao) and np.array_equal(B, Bo), name
    tpg.increment_flush()  # nothing pending: a no-op
    for name, (A, B, Ao, Bo) in mats.items():
        assert np.array_equal(A, Ao), name
    # changing rowInd between blocks of the same accumulators is refused
    A, B = np.zeros((50, 50), order="F"), np.zeros((50, 50), order="F")
    tpg.increment_ibs_counts(A, B, fbm, rows, cols[:100], flush=False)
    with pytest.raises(tpg._lib.TpgError):
        tpg.increment_ibs_counts(A, B, fbm, rows[::-1].copy(), cols[100:200], flush=False)
    with pytest.raises(tpg._lib.TpgError):
        tpg.resident_drop()  # pending increments
    tpg.increment_flush()
    tpg.resident_drop()


def test_allele_sharing_pad_quirk_opt_in(tpg):
    """Reference quirk Q1 (src/snp_as.cpp:57-63): off by default, reproducible on request -- through the whole-range
    driver (block count from CutBySize), through the accumulator object, and through the literal per-block mirror."""
    fbm = fx.families_fbm()
    X = tpg.FBM.from_numpy(fbm)
    lo, up = orc.cut_by_size(961, 300)  # 240, 240, 241, 240: three narrower blocks
    assert tpg.as_pad_quirk_blocks(961, 300) == 3 and tpg.as_pad_quirk_blocks(960, 300) == 0
    assert tpg.as_pad_quirk_blocks(6, 3) == 0 and tpg.as_pad_quirk_blocks(7, 3) == 2  # 2, 3, 2
    for bs in (300, 100, 961, 37):
        sizes = np.diff(np.concatenate([[0], orc.cut_by_size(961, bs)[1]]))
        assert tpg.as_pad_quirk_blocks(961, bs) == int((sizes < sizes.max()).sum())
    intended = orc.snp_allele_sharing(fbm, block_size=300)
    quirk = orc.snp_allele_sharing(fbm, block_size=300, emulate_as_pad_quirk=True)
    assert not np.array_equal(intended, quirk)
    assert np.array_equal(tpg.snp_allele_sharing(X, block_size=300), intended, equal_nan=True)
    assert np.array_equal(tpg.snp_allele_sharing(X, block_size=300, emulate_as_pad_quirk=True), quirk, equal_nan=True)
    assert np.allclose(tpg.pairwise_grm(X, block_size=300, emulate_as_pad_quirk=True), orc.pairwise_grm(quirk),
                       rtol=1e-12, atol=1e-14)
    # raw numerators through the accumulator object
    v = tpg.View(X, code256=None)
    pw = tpg.Pairwise(X.ctx, 12)
    pw.accumulate(v)
    base = pw.counts(("as_num", "as_den"))
    pw.set_as_pad_quirk(3)
    q = pw.counts(("as_num", "as_den"))
    assert np.array_equal(q["as_num"], base["as_num"] + 3) and np.array_equal(q["as_den"], base["as_den"])
    # literal mirror: the shim passes the width of the R driver's scratch matrices (the widest block)
    cols = np.arange(1, 962, dtype=np.int32)
    rows = np.arange(1, 13, dtype=np.int32)
    A = np.zeros((12, 12), order="F"); B = np.zeros((12, 12), order="F")
    Ao = np.zeros((12, 12), order="F"); Bo = np.zeros((12, 12), order="F")
    for a, b in zip(lo, up):
        tpg.increment_as_counts(A, B, fbm, rows, cols[a - 1:b], flush=False, scratch_cols=241, emulate_as_pad_quirk=True)
        orc.increment_as_counts(Ao, Bo, fbm, rows, cols[a - 1:b], pad_quirk=(b - a + 1) < 241)
    tpg.increment_flush()
    assert np.array_equal(A, Ao) and np.array_equal(B, Bo)
    tpg.resident_drop()


def test_two_contexts_keep_their_own_memory(tpg):
    """Two contexts (two streams) on one device, used alternately: each has its own device-memory pool, so scratch
    blocks freed by one are never handed to the other while its kernels may still be running."""
    n, m = 300, 20000
    fa, fb = orc.synth_fbm(31, n, m, npop=5, miss=0.03), orc.synth_fbm(32, n, m, npop=5, miss=0.03)
    ca, cb = tpg.Context(0), tpg.Context(0)
    Xa, Xb = tpg.FBM.from_numpy(fa, ctx=ca), tpg.FBM.from_numpy(fb, ctx=cb)
    gid = (np.arange(n) % 5).astype(np.int32)
    ea = orc.snp_ibs(fa, type="raw_counts")["ibs"]
    eb = orc.snp_ibs(fb, type="raw_counts")["ibs"]
    ga = orc.grouped_alt_freq_dip_pseudo_cpp(fa, None, None, gid, 5, np.full(n, 2.0))
    gb = orc.grouped_alt_freq_dip_pseudo_cpp(fb, None, None, gid, 5, np.full(n, 2.0))
    for _ in range(3):
        ra = tpg.snp_ibs(Xa, type="raw_counts")["ibs"]
        rb = tpg.snp_ibs(Xb, type="raw_counts")["ibs"]
        assert np.array_equal(ra, ea) and np.array_equal(rb, eb)
        va, vb = tpg.View(Xa), tpg.View(Xb)
        assert np.array_equal(tpg.grouped_alt_freq_dip_pseudo_cpp(va, gid, 5), ga, equal_nan=True)
        assert np.array_equal(tpg.grouped_alt_freq_dip_pseudo_cpp(vb, gid, 5), gb, equal_nan=True)
        va.free(); vb.free()
    Xa.free(); ca.close()          # destroying one context leaves the other's blocks alone
    assert np.array_equal(tpg.snp_ibs(Xb, type="raw_counts")["ibs"], eb)
    Xb.free(); cb.close()


# ---------------------------------------------------------------- Fst
def test_fst_scikit_allel_golden(tpg):
    # tests/testthat/test_pairwise_pop_fst.R:55-343
    gid = fx.FST_GROUPS_2
    for method, tag in (("Hudson", "fst_hudson"), ("WC84", "fst_wc")):
        r = tpg.pairwise_pop_fst(_X(tpg, fx.FST_7x6), None, None, gid, 2, method=method, by_locus=True)
        assert r["fst_tot"][0] == pytest.approx(float(fx.scikit(tag)), rel=0, abs=3e-16)
        assert np.allclose(r["fst_locus"][:, 0], fx.scikit(tag + "_per_loc"), rtol=0, atol=3e-16)
        mono = tpg.pairwise_pop_fst(_X(tpg, fx.FST_MONO_7x6), None, None, gid, 2, method=method)["fst_tot"]
        assert mono[0] == pytest.approx(float(fx.scikit(tag + "_monomorphic")), rel=0, abs=3e-16)
        a = tpg.pairwise_pop_fst(_X(tpg, fx.FST_MISSPOP_7x6), None, None, gid, 2, method=method)["fst_tot"]
        b = tpg.pairwise_pop_fst(_X(tpg, fx.FST_MISSPOP_7x5), None, None, gid, 2, method=method)["fst_tot"]
        assert a[0] == b[0]


@pytest.mark.parametrize("n,m,G", [(7, 6, 3), (300, 2500, 6), (500, 3000, 51)])
@pytest.mark.parametrize("method", ["Hudson", "WC84", "Nei87"])
def test_fst_vs_oracle(tpg, n, m, G, method):
    if n == 7:
        fbm, gid = orc.fbm_from_genotypes(fx.FST_7x6), fx.FST_GROUPS_3
    else:
        fbm = orc.synth_fbm(51, n, m, npop=G, miss=0.05)
        gid = (np.arange(n) % G).astype(np.int32)
    X = tpg.FBM.from_numpy(fbm)
    with np.errstate(invalid="ignore", divide="ignore"):
        o_tot = orc.pairwise_pop_fst(fbm, None, None, gid, G, method=method)["fst_tot"]
        o_loc = orc.pairwise_pop_fst(fbm, None, None, gid, G, method=method, by_locus=True)["fst_locus"]
        o_nd = orc.pairwise_pop_fst(fbm, None, None, gid, G, method=method, return_num_dem=True)
    t = tpg.pairwise_pop_fst(X, None, None, gid, G, method=method, by_locus=True)
    assert np.allclose(t["fst_tot"], o_tot, rtol=1e-12, atol=0, equal_nan=True)
    assert np.array_equal(t["fst_locus"], o_loc, equal_nan=True)  # same statements, contraction off -> same bits
    # sums-only call: WC84 takes the hand-reduced two-population form (fast reciprocals), rounding differs
    t_fast = tpg.pairwise_pop_fst(X, None, None, gid, G, method=method)["fst_tot"]
    assert np.allclose(t_fast, o_tot, rtol=1e-11, atol=1e-15, equal_nan=True)
    nd = tpg.pairwise_pop_fst(X, None, None, gid, G, method=method, return_num_dem=True)
    assert np.array_equal(nd["Fst_by_locus_num"], o_nd["Fst_by_locus_num"], equal_nan=True)
    assert np.array_equal(nd["Fst_by_locus_den"], o_nd["Fst_by_locus_den"], equal_nan=True)
    # literal loop mirrors fed with the oracle's summary matrices
    with np.errstate(invalid="ignore", divide="ignore"):
        pf = orc.grouped_summaries_dip_pseudo_cpp(fbm, None, None, gid, G, np.full(n, 2.0))
    pairs = tpg.combn2(G)
    if method == "Hudson":
        lm = tpg.pairwise_fst_hudson_loop(pairs, pf["n"], pf["freq_alt"], pf["freq_ref"])
    elif method == "WC84":
        lm = tpg.pairwise_fst_wc84_loop(pairs, pf["n"], pf["freq_alt"], pf["het_obs"])
    else:
        lm = tpg.pairwise_fst_nei87_loop(pairs, pf["n"], pf["het_obs"], pf["freq_alt"], pf["freq_ref"])
    assert np.allclose(lm["fst_tot"], o_tot, rtol=1e-12, atol=0, equal_nan=True)
    if method != "WC84":
        # the reference's Hudson / Nei87 loops read the caller's freq_ref; the device recomputes 1 - freq_alt, so a
        # matrix that is anything else is refused rather than silently ignored
        bad = pf["freq_ref"].copy()
        bad[3, 1] += 1e-9
        with pytest.raises(tpg._lib.TpgError):
            if method == "Hudson":
                tpg.pairwise_fst_hudson_loop(pairs, pf["n"], pf["freq_alt"], bad)
            else:
                tpg.pairwise_fst_nei87_loop(pairs, pf["n"], pf["het_obs"], pf["freq_alt"], bad)



def _fst_case(tpg, fbm, gid, G, method, pairs=None, tot_rtol=1e-11):
    """all outputs of pairwise_pop_fst for one panel against the oracle: by-locus values and numerators / denominators bit for
    bit, totals (both the by-locus call's and the totals-only kernels') and the sums=True path within tot_rtol"""
    import math

    n = fbm.shape[0]
    X = tpg.FBM.from_numpy(fbm)
    with np.errstate(invalid="ignore", divide="ignore"):
        pf = orc.grouped_summaries_dip_pseudo_cpp(fbm, None, None, gid, G, np.full(n, 2.0))
        allp = orc.combn2(G) if pairs is None else pairs
        if method == "Hudson":
            o = orc.pairwise_fst_hudson_loop(allp, pf["n"], pf["freq_alt"], pf["freq_ref"], True, True)
            o_loc = orc.pairwise_fst_hudson_loop(allp, pf["n"], pf["freq_alt"], pf["freq_ref"], True, False)
        elif method == "WC84":
            o = orc.pairwise_fst_wc84_loop(allp, pf["n"], pf["freq_alt"], pf["het_obs"], True, True)
            o_loc = orc.pairwise_fst_wc84_loop(allp, pf["n"], pf["freq_alt"], pf["het_obs"], True, False)
        else:
            o = orc.pairwise_fst_nei87_loop(allp, pf["n"], pf["het_obs"], pf["freq_alt"], pf["freq_ref"], True, True)
            o_loc = orc.pairwise_fst_nei87_loop(allp, pf["n"], pf["het_obs"], pf["freq_alt"], pf["freq_ref"], True, False)
    num, den = np.asarray(o["Fst_by_locus_num"]), np.asarray(o["Fst_by_locus_den"])
    P = allp.shape[1]
    # the correctly rounded sums over the loci the reference keeps (src/pairwise_fst_hudson_loop.cpp:43-52: a locus is
    # dropped when its numerator or denominator is NaN)
    keep = ~(np.isnan(num) | np.isnan(den))
    sn = np.array([math.fsum(num[keep[:, k], k]) for k in range(P)])
    sd = np.array([math.fsum(den[keep[:, k], k]) for k in range(P)])
    scale_n = np.array([math.fsum(np.abs(num[keep[:, k], k])) for k in range(P)])  # the numerators of a pair can cancel
    with np.errstate(invalid="ignore", divide="ignore"):
        o_tot = sn / sd
        tot_tol = tot_rtol * scale_n / np.abs(sd) + 1e-300

    def same_tot(got):
        assert np.array_equal(np.isnan(got), np.isnan(o_tot))
        fin = ~np.isnan(o_tot)
        assert np.all(np.abs(got[fin] - o_tot[fin]) <= tot_tol[fin])

    kw = dict(method=method, pairwise_combn=pairs)
    t = tpg.pairwise_pop_fst(X, None, None, gid, G, by_locus=True, **kw)
    assert np.array_equal(t["fst_locus"], np.asarray(o_loc["fst_locus"]), equal_nan=True)
    same_tot(t["fst_tot"])
    nd = tpg.pairwise_pop_fst(X, None, None, gid, G, return_num_dem=True, **kw)
    assert np.array_equal(nd["Fst_by_locus_num"], num, equal_nan=True)
    assert np.array_equal(nd["Fst_by_locus_den"], den, equal_nan=True)
    t_fast = tpg.pairwise_pop_fst(X, None, None, gid, G, **kw)["fst_tot"]  # totals-only kernels
    same_tot(t_fast)
    s = tpg.pairwise_pop_fst(X, None, None, gid, G, sums=True, **kw)
    assert np.all(np.abs(s["sum_num"] - sn) <= tot_rtol * scale_n + 1e-300)
    assert np.allclose(s["sum_den"], sd, rtol=tot_rtol, atol=0)
    X.free()


@pytest.mark.parametrize("G", [64, 65, 130, 300])
@pytest.mark.parametrize("method", ["Hudson", "WC84", "Nei87"])
def test_fst_many_small_populations(tpg, G, method):
    """More than 64 populations leave the 64-population kernels (csrc/fst.hip run_fst: the WC84 tile / table kernels at a
    stride of 64, the Hudson products in 4 x 4 tiles of 16): G = 64 is the last case on them, 65 the first off, 130 and 300
    (8 385 / 44 850 pairs) take the many-pairs passes; population G - 1 has ONE individual (n = 2 alleles: n - 1 = 1,
    WC84's n_c and Nei87's harmonic mean at their smallest), population G - 2 two."""
    m = 301 if G < 300 else 130
    n = 3 * (G - 2) + 3
    fbm = orc.synth_fbm(77 + G, n, m, npop=min(G, 51), miss=0.06)
    gid = np.concatenate([np.repeat(np.arange(G - 2), 3), [G - 2, G - 2, G - 1]]).astype(np.int32)
    fbm[gid == 5, 11] = 3      # a population without a valid genotype at a locus
    fbm[:, 17] = 2             # a monomorphic locus
    _fst_case(tpg, fbm, gid, G, method)


@pytest.mark.parametrize("G,n", [(2, 5000), (5, 5000), (3, 700)])
@pytest.mark.parametrize("method", ["Hudson", "WC84", "Nei87"])
def test_fst_large_populations(tpg, G, n, method):
    """Populations of 2 500 / 1 000 individuals: a pair's valid alleles reach 10 000 / 4 000, past the Hudson reciprocal
    table (4 096) and the WC84 tile kernel's FSTT_KMAX (511), and at G = 2 past what the WC84 table holds in LDS; (3, 700):
    populations of 233 -- inside the tables' range but past the tile kernel's."""
    m = 1500
    fbm = orc.synth_fbm(91 + G, n, m, npop=G, miss=0.03)
    gid = (np.arange(n) % G).astype(np.int32)
    _fst_case(tpg, fbm, gid, G, method)


def test_global_stats_many_populations(tpg):
    n, m, G = 520, 700, 130
    fbm = orc.synth_fbm(131, n, m, npop=51, miss=0.1)
    gid = (np.arange(n) % G).astype(np.int32)
    X = tpg.FBM.from_numpy(fbm)
    o_loc = orc.pop_global_stats(fbm, None, None, gid, G, by_locus=True)
    t_loc = tpg.pop_global_stats(X, None, None, gid, G, by_locus=True)
    fin = np.isfinite(o_loc)
    assert np.array_equal(np.isnan(t_loc), np.isnan(o_loc)) and np.array_equal(np.isinf(t_loc), np.isinf(o_loc))
    assert np.allclose(t_loc[fin], o_loc[fin], rtol=1e-12, atol=1e-13)
    assert np.allclose(tpg.pop_global_stats(X, None, None, gid, G), orc.pop_global_stats(fbm, None, None, gid, G),
                       rtol=1e-11, atol=1e-13, equal_nan=True)
    v = tpg.View(X)
    assert np.array_equal(tpg.grouped_genotype_counts(v, gid, G), orc.grouped_genotype_counts(fbm, None, None, gid, G))
    with np.errstate(invalid="ignore", divide="ignore"):
        sa = tpg.grouped_summaries_dip_pseudo_cpp(v, gid, G, np.full(n, 2.0))
        sb = orc.grouped_summaries_dip_pseudo_cpp(fbm, None, None, gid, G, np.full(n, 2.0))
    for k in sb:
        assert np.array_equal(sa[k], sb[k], equal_nan=True), k
    X.free()


def test_wc84_totals_tiles_any_pair_list(tpg):
    """More than 256 pairs of at most 64 populations take the tiled totals kernel (tpg_fst_wc84_tile_kernel: a thread owns a
    3 x 2 tile of populations): a pair list in any order, with (g2, g1) orientations and repeated pairs, an empty population,
    a population without a valid genotype at a locus, a last chunk of odd length -- against sums of the oracle's by-locus
    numerators and denominators."""
    import math

    n, m, G = 330, 1237, 31  # 1 237 = 77 chunks of 16 loci + 5
    Xb = orc.synth_fbm(23, n, m, npop=G - 1, miss=0.04)
    gid = (np.arange(n) % (G - 1)).astype(np.int32)  # population G - 1 is empty
    Xb[gid == 3, 7] = 3  # missing
    rng = np.random.default_rng(23)
    allp = tpg.combn2(G)
    cols = rng.permutation(allp.shape[1])[:420]
    pairs = allp[:, cols].copy()
    flip = rng.random(pairs.shape[1]) < 0.3
    pairs[:, flip] = pairs[::-1, flip]
    pairs = np.ascontiguousarray(np.concatenate([pairs, pairs[:, :9]], axis=1))  # nine pairs listed twice
    X = tpg.FBM.from_numpy(Xb)
    got = tpg.pairwise_pop_fst(X, None, None, gid, G, method="WC84", pairwise_combn=pairs, sums=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        pf = orc.grouped_summaries_dip_pseudo_cpp(Xb, None, None, gid, G, np.full(n, 2.0))
        ref = orc.pairwise_fst_wc84_loop(pairs, pf["n"], pf["freq_alt"], pf["het_obs"], True, True)
    num, den = np.asarray(ref["Fst_by_locus_num"]), np.asarray(ref["Fst_by_locus_den"])
    P = pairs.shape[1]
    sn = np.array([math.fsum(num[~np.isnan(den[:, k]), k]) for k in range(P)])
    sd = np.array([math.fsum(den[~np.isnan(den[:, k]), k]) for k in range(P)])
    ok = sd != 0  # (pairs with the empty population have no locus at all)
    assert ok.sum() > 380 and (~ok).sum() > 0
    assert np.allclose(got["sum_num"][ok], sn[ok], rtol=1e-10, atol=1e-12)
    assert np.allclose(got["sum_den"][ok], sd[ok], rtol=1e-10, atol=1e-12)
    assert np.all(got["sum_den"][~ok] == 0)
    assert np.array_equal(got["sum_num"][-9:], got["sum_num"][:9])  # the repeated pairs
    X.free()


def test_fst_pseudohaploid_hudson_only(tpg):
    n, m, G = 120, 900, 4
    fbm = orc.synth_fbm(61, n, m, npop=G, miss=0.05)
    ploidy = np.full(n, 2.0)
    ploidy[::3] = 1.0
    sub = fbm[::3]
    sub[sub == 1] = 0
    fbm[::3] = sub
    gid = (np.arange(n) % G).astype(np.int32)
    X = tpg.FBM.from_numpy(fbm)
    with np.errstate(invalid="ignore", divide="ignore"):
        o = orc.pairwise_pop_fst(fbm, None, None, gid, G, ploidy=ploidy, method="Hudson")["fst_tot"]
    assert np.allclose(tpg.pairwise_pop_fst(X, None, None, gid, G, ploidy=ploidy, method="Hudson")["fst_tot"], o, rtol=1e-12)
    with pytest.raises(tpg._lib.TpgError):  # R/pairwise_pop_fst.R:113-115
        tpg.pairwise_pop_fst(X, None, None, gid, G, ploidy=ploidy, method="WC84")


# ---------------------------------------------------------------- PCA
def _align_sign(a, b):
    """flip the columns of a so that they correlate positively with b"""
    s = np.sign((a * b).sum(axis=0))
    s[s == 0] = 1
    return a * s


def test_pca_families_against_prcomp_definition(tpg):
    # tests/testthat/test_gt_pca.R:320-374: std.dev = d / sqrt(n - 1) and percent = d^2 / ||Z||_F^2 at 1e-4
    fam = fx.families_fbm()
    Xf = fam.astype(float)
    keep = np.where(((Xf == 3).sum(axis=0) == 0))[0]
    maf = Xf[:, keep].sum(axis=0) / 24
    maf = np.minimum(maf, 1 - maf)
    keep = keep[maf > 0.01]
    cols = (keep + 1).astype(np.int32)
    X = tpg.FBM.from_numpy(fam)
    res = tpg.gt_pca_partialSVD(X, None, cols, k=10, code256=tpg.CODE_012)
    Z = (Xf[:, keep] - res["center"]) / res["scale"]
    s = np.linalg.svd(Z, compute_uv=False)
    assert np.allclose(res["d"] / np.sqrt(11), s[:10] / np.sqrt(11), rtol=1e-6)
    assert res["square_frobenius"] == pytest.approx((Z ** 2).sum(), rel=1e-12)
    o = orc.gt_pca_partialSVD(fam, None, cols, k=10, code256=orc.CODE_012)
    assert np.array_equal(res["center"], o["center"]) and np.array_equal(res["scale"], o["scale"])
    assert np.allclose(res["d"], o["d"], rtol=1e-6)


@pytest.mark.parametrize("n,m,G,k", [(60, 500, 3, 5), (210, 1500, 4, 6), (300, 4000, 6, 10), (500, 6000, 12, 20)])
def test_pca_vs_oracle(tpg, n, m, G, k):
    fbm = orc.synth_fbm(71, n, m, npop=G, miss=0.03, imputed_bytes=True)
    # drop monomorphic loci (big_SVD stops on a zero scale)
    dec = np.where(fbm > 3, fbm - 4, fbm)
    cols = (np.where((dec.sum(axis=0) > 0) & (dec.sum(axis=0) < 2 * n))[0] + 1).astype(np.int32)
    X = tpg.FBM.from_numpy(fbm)
    o = orc.gt_pca_partialSVD(fbm, None, cols, k=k)
    r = tpg.gt_pca_partialSVD(X, None, cols, k=k)
    assert np.array_equal(r["center"], o["center"]) and np.array_equal(r["scale"], o["scale"])
    assert r["square_frobenius"] == pytest.approx(o["square_frobenius"], rel=1e-12)
    assert np.allclose(r["d"], o["d"], rtol=1e-6, atol=0)  # tolerance of BASELINE.json: 1e-6 relative
    # scores u*d and loadings v, sign aligned (eigenvectors are defined up to sign)
    so, sr = o["u"] * o["d"], _align_sign(r["u"] * r["d"], o["u"] * o["d"])
    assert np.max(np.abs(sr - so)) <= 1e-6 * np.max(np.abs(so))
    vr = _align_sign(r["v"], o["v"])
    assert np.max(np.abs(vr - o["v"])) <= 1e-6 * np.max(np.abs(o["v"]))
    # Gram matrix itself
    v = tpg.View(X, None, cols, code256=tpg.CODE_IMPUTE_PRED)
    c, s = tpg.pca_center_scale(v)
    K = tpg.pca_gram(v, c, s)
    _, _, Ko = orc.pca_gram(fbm, None, cols)
    assert np.max(np.abs(K - Ko)) <= 1e-6 * np.max(np.abs(Ko))
    assert np.array_equal(K, K.T)
    # projection kernel (src/fbm_prod_and_rowSumSq.cpp), incl. missing values -> 0 under CODE_012
    XV, rss = tpg.fbm256_prod_and_rowSumsSq(X, None, cols, o["center"], o["scale"], o["v"], code256=tpg.CODE_012)
    XVo, rsso = orc.fbm256_prod_and_rowSumsSq(fbm, None, cols, o["center"], o["scale"], o["v"], code256=orc.CODE_012)
    assert np.allclose(XV, XVo, rtol=1e-9, atol=1e-9 * np.max(np.abs(XVo)))
    assert np.allclose(rss, rsso, rtol=1e-10)
    assert tpg.square_frobenius(X, None, cols, o["center"], o["scale"]) == pytest.approx(o["square_frobenius"], rel=1e-12)


@pytest.mark.parametrize("n,m", [(8, 1), (24, 130), (104, 257), (136, 64), (264, 1000), (520, 333), (200, 129)])
def test_loci_counts_from_the_pack_equal_counts_over_L(tpg, n, m):
    """Per-locus genotype counts: the fast pack kernel (FBM rows a multiple of 8, all rows in file order) leaves per-chunk counts
    beside the layouts and tpg_loci_counts adds them up; a view of a row subset takes the kernel over the L layout.  Both against
    numpy: ragged last chunks of 8 / 16 individuals, a last locus group of one locus, a column subset, the pair of views."""
    rng = np.random.default_rng(n * 1000 + m)
    g = rng.integers(0, 4, size=(n, m)).astype(np.uint8)  # 3 = missing
    X = tpg.FBM.from_numpy(g)
    want = np.stack([(g == c).sum(0) for c in range(4)], axis=1).astype(np.int32)
    v = tpg.View(X)
    assert np.array_equal(tpg.loci_counts(v), want)
    a, b = tpg.View.pair(X, None, None, tpg.CODE_012, tpg.CODE_IMPUTE_PRED)
    assert np.array_equal(tpg.loci_counts(a), want) and np.array_equal(tpg.loci_counts(b), want)
    if m > 2:
        cols = np.sort(rng.choice(m, size=max(1, m // 2), replace=False)).astype(np.int32) + 1
        assert np.array_equal(tpg.loci_counts(tpg.View(X, None, cols)), want[cols - 1])
    rows = (np.arange(0, n, 2) + 1).astype(np.int32)  # a row subset: the generic pack kernel, counts over L
    wr = np.stack([(g[::2] == c).sum(0) for c in range(4)], axis=1).astype(np.int32)
    assert np.array_equal(tpg.loci_counts(tpg.View(X, rows, None)), wr)


def test_pca_loadings_entry_point_and_small_context_calls(tpg):
    """tpg_pca_loadings on its own (v = Z' u / d, the second sweep of big_SVD: the driver reaches it through the PCA, a caller
    with its own u may not), on a caller's stream (tpg_ctx_set_stream), and the two one-line queries of the ABI."""
    n, m, k = 150, 2100, 7
    g = orc.synth_fbm(5, n, m, npop=3, miss=0.0)
    g = g[:, (g.sum(0) > 0) & (g.sum(0) < 2 * n)]
    X = _X(tpg, g)
    v = tpg.View(X)
    center, scale = tpg.pca_center_scale(v)
    rng = np.random.default_rng(3)
    U, _ = np.linalg.qr(rng.standard_normal((n, k)))
    d = np.linspace(9.0, 2.0, k)
    want = ((g.astype(float) - center) / scale).T @ U / d
    got = tpg.pca_loadings(v, center, scale, U, d)
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-9 * np.abs(want).max()
    # a caller's stream, made with the HIP runtime the library itself runs on (dlopen by soname gives the copy already loaded.
    # Not torch.cuda.Stream(): when torch is imported AFTER the library the process holds two HIP runtimes -- torch's bundled one
    # and /opt/rocm's -- and the second to initialise sees no device; README.md "with PyTorch in the same process")
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")
    st = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0 and st.value
    ctx = tpg.default_context()
    ctx.set_stream(st.value)
    try:
        again = tpg.pca_loadings(v, center, scale, U, d)
        ctx.sync()
    finally:
        ctx.set_stream(None)
        assert hip.hipStreamDestroy(st) == 0
    assert np.array_equal(again, got)
    assert tpg.Pairwise.buffer_bytes(n) >= 5 * 4 * n * (n + 1) // 2  # five int32 planes over the tiles of one triangle
    try:
        uid = tpg.Comm.unique_id()
    except RuntimeError:
        uid = None  # no RCCL on this box: the library says so instead of handing out a made-up id
    assert uid is None or (len(uid) == 128 and any(uid))


def test_pca_errors_like_big_svd(tpg):
    fbm = orc.synth_fbm(72, 50, 300, npop=2, miss=0.05)
    X = tpg.FBM.from_numpy(fbm)
    with pytest.raises(tpg._lib.TpgError) as e:  # missing values (CODE_012 keeps byte 3 missing)
        tpg.gt_pca_partialSVD(X, k=3, code256=tpg.CODE_012)
    assert e.value.code == 4
    mono = orc.synth_fbm(72, 50, 300, npop=2, miss=0.0)
    mono[:, 7] = 0
    with pytest.raises(tpg._lib.TpgError) as e:  # zero scale
        tpg.gt_pca_partialSVD(tpg.FBM.from_numpy(mono), k=3)
    assert e.value.code == 4


# ---------------------------------------------------------------- more §8f(2) rows
@pytest.mark.parametrize("n,m,G,miss", [(7, 6, 3, None), (60, 200, 4, 0.1), (300, 2500, 6, 0.05), (500, 1500, 51, 0.3)])
def test_grouped_genotype_counts_and_global_stats(tpg, n, m, G, miss):
    if miss is None:
        fbm, gid = orc.fbm_from_genotypes(fx.FST_7x6), fx.FST_GROUPS_3
    else:
        fbm = orc.synth_fbm(71, n, m, npop=G, miss=miss)
        gid = (np.arange(n) % G).astype(np.int32)
    X = tpg.FBM.from_numpy(fbm)
    v = tpg.View(X)
    # the genotype table gt_grouped_hwe builds per locus (src/hwe.cpp:238-250): exact
    assert np.array_equal(tpg.grouped_genotype_counts(v, gid, G), orc.grouped_genotype_counts(fbm, None, None, gid, G))
    # pop_global_stats (R/pop_global_stats.R:113-212): same statements in FP64, contraction off
    o_loc = orc.pop_global_stats(fbm, None, None, gid, G, by_locus=True)
    t_loc = tpg.pop_global_stats(X, None, None, gid, G, by_locus=True)
    assert np.array_equal(np.isnan(t_loc), np.isnan(o_loc))
    assert np.array_equal(np.isinf(t_loc), np.isinf(o_loc))
    fin = np.isfinite(o_loc)
    assert np.allclose(t_loc[fin], o_loc[fin], rtol=1e-12, atol=1e-13)
    o_all = orc.pop_global_stats(fbm, None, None, gid, G)
    t_all = tpg.pop_global_stats(X, None, None, gid, G)
    assert np.allclose(t_all, o_all, rtol=1e-11, atol=1e-13, equal_nan=True)
    # subsets of rows / loci go through the same view machinery
    rows = np.arange(1, n + 1, 2, dtype=np.int32)
    cols = np.arange(m, 0, -3, dtype=np.int32)
    o_sub = orc.pop_global_stats(fbm, rows, cols, gid[rows - 1], G, by_locus=True)
    t_sub = tpg.pop_global_stats(X, rows, cols, gid[rows - 1], G, by_locus=True)
    fin = np.isfinite(o_sub)
    assert np.array_equal(np.isfinite(t_sub), fin) and np.allclose(t_sub[fin], o_sub[fin], rtol=1e-12, atol=1e-13)


def test_global_stats_single_population_and_ploidy(tpg):
    # tests/testthat/test_pop_basic_stats.R:160-176: Fstp is not defined for a single population
    fbm = orc.fbm_from_genotypes(fx.FST_7x6)
    X = tpg.FBM.from_numpy(fbm)
    gid = np.zeros(7, dtype=np.int32)
    loc = tpg.pop_global_stats(X, None, None, gid, 1, by_locus=True)
    assert np.all(np.isnan(loc[:, 7]))
    with pytest.raises(tpg._lib.TpgError):  # stopifnot_diploid
        tpg.pop_global_stats(X, None, None, gid, 1, ploidy=np.array([2, 2, 1, 2, 2, 2, 2.0]))


@pytest.mark.parametrize("n,m,G", [(7, 6, 3), (300, 2500, 6), (500, 1500, 51)])
def test_pop_fst_and_fis_wg17_from_allele_sharing(tpg, n, m, G):
    if n == 7:
        fbm, gid = orc.fbm_from_genotypes(fx.FST_7x6), fx.FST_GROUPS_3
    else:
        fbm = orc.synth_fbm(81, n, m, npop=G, miss=0.05)
        gid = (np.arange(n) % G).astype(np.int32)
    X = tpg.FBM.from_numpy(fbm)
    A_o = orc.snp_allele_sharing(fbm)
    A_t = tpg.snp_allele_sharing(X)
    # block means of an arbitrary matrix (NaN entries and the diagonal are skipped)
    B = A_o.copy()
    B[1, 0] = np.nan
    mean, cnt = tpg.block_means(B, gid, G, skip_diag=True)
    Bn = B.copy()
    np.fill_diagonal(Bn, np.nan)
    for g1 in range(G):
        for g2 in range(G):
            blk = Bn[np.ix_(np.where(gid == g1)[0], np.where(gid == g2)[0])]
            assert cnt[g1, g2] == np.sum(~np.isnan(blk))
            if cnt[g1, g2]:
                assert mean[g1, g2] == pytest.approx(np.nanmean(blk), rel=1e-12, abs=1e-15)
            else:
                assert np.isnan(mean[g1, g2])
    for glob in (False, True):
        assert np.allclose(tpg.pop_fst(X, None, None, gid, G, include_global=glob), orc.pop_fst(A_o, gid, G, glob),
                           rtol=1e-9, atol=1e-12, equal_nan=True)
        assert np.allclose(tpg.pop_fis_wg17(X, None, None, gid, G, include_global=glob),
                           orc.pop_fis_wg17(A_o, gid, G, glob), rtol=1e-9, atol=1e-12, equal_nan=True)
    # a pre-computed matrix gives the same answer (R/pop_fst.R:36-38)
    assert np.allclose(tpg.pop_fst(X, None, None, gid, G, allele_sharing_mat=A_t), tpg.pop_fst(X, None, None, gid, G),
                       rtol=0, atol=0, equal_nan=True)


# ---------------------------------------------------------------- §8f(3): windows
def test_windows_stats_generic_reference_expectations(tpg):
    # tests/testthat/test_window_stats_generic.R:1-86
    x = np.array([1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 14, 15, 16], dtype=float)
    chrom = np.array(["chr1"] * 6 + ["chr2"] * 7)
    pos = np.array([50, 120, 150, 180, 230, 390, 110, 120, 150, 180, 230, 280, 350])
    for kw in (dict(operator="sum", window_size=4, step_size=3, size_unit="snp", min_loci=1),
               dict(operator="sum", window_size=4, step_size=3, size_unit="snp", min_loci=1, complete=True),
               dict(operator="sum", window_size=100, step_size=50, size_unit="bp", min_loci=1),
               dict(operator="mean", window_size=100, step_size=50, size_unit="bp", min_loci=2),
               dict(operator="mean", window_size=5, step_size=1, size_unit="snp", min_loci=3, complete=True)):
        t = tpg.windows_stats_generic(x, chrom, pos, **kw)
        o = orc.windows_stats_generic(x, chrom, pos, **kw)
        for key in ("start", "end"):
            assert np.array_equal(t[key], o[key])
        assert list(t["chromosome"]) == list(o["chromosome"])
        assert np.array_equal(t["n_loci"], o["n_loci"], equal_nan=True)
        assert np.allclose(t["stat"], o["stat"], rtol=1e-15, atol=0, equal_nan=True)
    w = tpg.windows_stats_generic(x, chrom, pos, operator="sum", window_size=4, step_size=3, size_unit="snp", min_loci=1)
    assert list(w["n_loci"]) == [4, 3, 4, 4] and w["stat"][0] == 10 and w["stat"][3] == 58
    for bad in (dict(window_size=-1, step_size=1), dict(window_size=4, step_size=-1),
                dict(window_size=4, step_size=1, min_loci=-1), dict(window_size=4, step_size=1, min_loci=9),
                dict(window_size=4, step_size=1, complete="blah"), dict(window_size=4, step_size=1, operator="nope")):
        with pytest.raises(ValueError):
            tpg.windows_stats_generic(x, chrom, pos, **{"operator": "sum", "min_loci": 1, **bad})


@pytest.mark.parametrize("n,m,G", [(10, 8, 3), (300, 2500, 6)])
def test_windows_pairwise_pop_fst(tpg, n, m, G):
    fbm = orc.synth_fbm(91, n, m, npop=G, miss=0.1)
    gid = (np.arange(n) % G).astype(np.int32)
    X = tpg.FBM.from_numpy(fbm)
    chrom = np.array(["chr1"] * (m // 3) + ["chr2"] * (m - m // 3))
    rng = np.random.default_rng(5)
    pos = np.concatenate([np.sort(rng.integers(1, 50 * m, m // 3)), np.sort(rng.integers(1, 50 * m, m - m // 3))])
    for kw in (dict(window_size=3, step_size=2, size_unit="snp", min_loci=2),
               dict(window_size=40, step_size=40, size_unit="snp", min_loci=1, complete=True),
               dict(window_size=5000, step_size=2500, size_unit="bp", min_loci=2)):
        t = tpg.windows_pairwise_pop_fst(X, None, None, gid, G, chrom, pos, **kw)
        o = orc.windows_pairwise_pop_fst(fbm, None, None, gid, G, chrom, pos, **kw)
        assert np.array_equal(t["start"], o["start"]) and np.array_equal(t["end"], o["end"])
        assert np.array_equal(np.isnan(t["fst"]), np.isnan(o["fst"]))
        assert np.allclose(t["fst"], o["fst"], rtol=1e-11, atol=1e-14, equal_nan=True)
    # tests/testthat/test_window_pairwise_pop_fst.R:113-160: the first SNP window of a chromosome equals the
    # pairwise Fst of those loci on their own
    t = tpg.windows_pairwise_pop_fst(X, None, None, gid, G, chrom, pos, window_size=3, step_size=2, size_unit="snp", min_loci=2)
    wr = tpg.window_index_ranges(chrom, pos, 3, 2, "snp")
    for w in (0, len(wr["lo"]) - 1):
        loci = np.arange(wr["lo"][w] + 1, wr["hi"][w] + 1, dtype=np.int32)
        if len(loci) < 2:
            continue
        alone = tpg.pairwise_pop_fst(X, None, loci, gid, G, method="Hudson")["fst_tot"]
        assert np.allclose(t["fst"][w], alone, rtol=1e-11, atol=1e-14, equal_nan=True)


def test_open_bk_chunked_upload(tpg, tmp_path):
    # backing files go up in 32-MiB pieces through pinned slots (runtime.hip: tpg_upload): 2.x chunks + a ragged tail
    n, m = 1001, 80_123
    rng = np.random.default_rng(3)
    a = np.asfortranarray(rng.integers(0, 4, size=(n, m), dtype=np.uint8))
    path = tmp_path / "panel.bk"
    a.T.tofile(path)  # column-major bytes, as bigstatsr writes them
    X = tpg.FBM.open_bk(str(path), n, m)
    assert np.array_equal(X.to_numpy(), a)
    cnt = tpg.loci_counts(tpg.View(X, code256=None))
    assert np.array_equal(cnt, np.stack([(a == c).sum(axis=0) for c in range(4)], axis=1))


@pytest.mark.parametrize("n,m,G,method", [(10, 6, 4, "Hudson"), (300, 2500, 5, "Hudson"), (300, 2500, 5, "WC84")])
def test_nwise_pop_pbs(tpg, n, m, G, method):
    # R/nwise_pop_pbs.R; tests/testthat/test_nwise_pop_pbs.R:55-62 expects 6 columns per triplet (24 for 4 populations)
    fbm = orc.synth_fbm(101, n, m, npop=G, miss=0.1)
    gid = (np.arange(n) % G).astype(np.int32)
    X = tpg.FBM.from_numpy(fbm)
    t = tpg.nwise_pop_pbs(X, None, None, gid, G, fst_method=method, return_fst=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        o = orc.nwise_pop_pbs(fbm, None, None, gid, G, fst_method=method)
        o_fst = orc.pairwise_pop_fst(fbm, None, None, gid, G, method=method, by_locus=True)["fst_locus"]
    ntrip = G * (G - 1) * (G - 2) // 6
    assert t["pbs"].shape == (m, 6 * ntrip) == o.shape and len(t["names"]) == 6 * ntrip
    assert np.array_equal(t["fst"], o_fst, equal_nan=True)
    assert np.array_equal(np.isnan(t["pbs"]), np.isnan(o))
    assert np.allclose(t["pbs"], o, rtol=1e-12, atol=1e-14, equal_nan=True)
    with pytest.raises(ValueError):
        tpg.nwise_pop_pbs(X, None, None, gid % 2, 2)


@pytest.mark.parametrize("n,m,G", [(7, 6, 3), (300, 2500, 6), (500, 1500, 51)])
def test_pop_basic_stats(tpg, n, m, G):
    # pop_het_obs / pop_het_exp / pop_fis (Nei87): R/pop_het_obs.R, R/pop_het_exp.R, R/pop_fis.R:81-133
    if n == 7:
        fbm, gid = orc.fbm_from_genotypes(fx.FST_7x6), fx.FST_GROUPS_3
    else:
        fbm = orc.synth_fbm(111, n, m, npop=G, miss=0.1)
        gid = (np.arange(n) % G).astype(np.int32)
    X = tpg.FBM.from_numpy(fbm)
    fns = {"Ho": tpg.pop_het_obs, "Hs": tpg.pop_het_exp, "Fis": lambda *a, **k: tpg.pop_fis(*a, method="Nei87", **k)}
    for which, fn in fns.items():
        for by_locus in (True, False):
            for glob in (False, True):
                t = fn(X, None, None, gid, G, by_locus=by_locus, include_global=glob)
                o = orc.pop_basic_stat(fbm, None, None, gid, G, which, by_locus, glob)
                assert t.shape == o.shape
                assert np.array_equal(np.isnan(t), np.isnan(o)), (which, by_locus, glob)
                fin = np.isfinite(o)
                assert np.array_equal(np.isfinite(t), fin)
                assert np.allclose(t[fin], o[fin], rtol=1e-11, atol=1e-13), (which, by_locus, glob)
    assert tpg.pop_gene_div is tpg.pop_het_exp
    with pytest.raises(ValueError):
        tpg.pop_fis(X, None, None, gid, G, method="Nei87", allele_sharing_mat=np.ones((n, n)))
    with pytest.raises(ValueError):
        tpg.pop_fis(X, None, None, gid, G, method="WG17", by_locus=True)


# ---------------------------------------------------------------- SURVEY.md 8f(3), 8f(4)
def test_filter_high_relatedness(tpg):
    # reference case (tests/testthat/test_filter_high_relatedness.R): families, KING, threshold 0.2; re-filtering keeps all
    fam = fx.families_fbm()
    X = tpg.FBM.from_numpy(fam)
    king = tpg.snp_king(X)
    res = tpg.filter_high_relatedness(king, 0.2)
    o_passed, o_removed, o_keep = orc.filter_high_relatedness(king, 0.2)
    assert list(res[0]) == [str(k) for k in o_passed] and list(res[1]) == [str(k) for k in o_removed]
    assert np.array_equal(res[2], o_keep) and res[2].sum() == 10
    sub = king[np.ix_(res[2], res[2])]
    assert tpg.filter_high_relatedness(sub, 0.2)[2].all()
    assert tpg.filter_high_relatedness(np.array([[0.5]]), 0.2)[2].all()
    # the matrix straight out of HBM (no host copy of the KING matrix in between)
    v = tpg.View(X, code256=None)
    pw = tpg.Pairwise(X.ctx, 12)
    pw.accumulate(v)
    d_king = X.ctx.dev_alloc(8 * 144)
    tpg._lib.check(tpg._lib.lib.tpg_pairwise_king(X.ctx.h, pw.h, d_king))
    res_d = tpg.filter_high_relatedness(d_king, 0.2, ids=[f"ind{k}" for k in range(12)])
    X.ctx.dev_free(d_king)
    assert np.array_equal(res_d[2], o_keep) and list(res_d[1]) == [f"ind{k - 1}" for k in o_removed]
    # random symmetric matrices with many pairs over the threshold, duplicates (exact ties of the two means) and signs
    rng = np.random.default_rng(5)
    for n, thr in ((25, 0.3), (40, 0.15), (33, 0.05)):
        A = rng.random((n, n)) * 0.5 - 0.1
        A = (A + A.T) / 2
        A[3] = A[7]; A[:, 3] = A[:, 7]          # two identical individuals
        np.fill_diagonal(A, 0.5)
        t = tpg.filter_high_relatedness(A, thr)
        o = orc.filter_high_relatedness(A, thr)
        assert np.array_equal(t[2], o[2]) and list(t[0]) == [str(k) for k in o[0]], (n, thr)
    # an NA among the coefficients: if the greedy walk gets to compare it, R stops (`if (NA)`), and so do we; if the
    # pair is never compared (one of the two was dropped before), the result is the oracle's
    saw_error = saw_result = False
    for seed in range(12):
        r2 = np.random.default_rng(100 + seed)
        B = r2.random((15, 15)) * 0.3
        B = (B + B.T) / 2
        i, j = r2.choice(15, 2, replace=False)
        B[i, j] = B[j, i] = np.nan
        try:
            o = orc.filter_high_relatedness(B, 0.25)
        except ValueError:
            with pytest.raises(tpg._lib.TpgError):
                tpg.filter_high_relatedness(B, 0.25)
            saw_error = True
            continue
        t = tpg.filter_high_relatedness(B, 0.25)
        assert np.array_equal(t[2], o[2]) and list(t[0]) == [str(k) for k in o[0]]
        saw_result = True
    assert saw_error and saw_result


def test_predict_gt_pca_projections(tpg):
    # R/predict_gt_pca.R:73-236: no new data, "none", "simple", "least_squares" against the oracle's restatement
    n, m, k = 150, 1200, 6
    fbm = orc.synth_fbm(41, n, m, npop=4, miss=0.0)
    keep = np.where((fbm.sum(axis=0) > 0) & (fbm.sum(axis=0) < 2 * n))[0]
    cols = (keep + 1).astype(np.int32)
    X = tpg.FBM.from_numpy(fbm)
    pca = tpg.gt_pca_partialSVD(X, None, cols, k=k)
    opca = dict(d=pca["d"], u=pca["u"], v=pca["v"], center=pca["center"], scale=pca["scale"])
    assert np.array_equal(tpg.predict_gt_pca(pca), pca["u"] * pca["d"])
    # new data: other individuals, some genotypes missing (stored as imputed bytes and as plain NA)
    new = orc.synth_fbm(42, 60, m, npop=4, miss=0.06, imputed_bytes=True)
    Xn = tpg.FBM.from_numpy(new)
    for method in ("none", "simple"):
        t = tpg.predict_gt_pca(pca, Xn, None, cols, project_method=method)
        o = orc.predict_gt_pca(opca, new, None, cols, project_method=method)
        assert np.allclose(t, o, rtol=1e-9, atol=1e-9 * np.abs(o).max()), method
    # projecting the data the PCA was built on reproduces U D
    self_proj = tpg.predict_gt_pca(pca, X, None, cols, project_method="simple")
    assert np.allclose(self_proj, pca["u"] * pca["d"], atol=1e-7 * pca["d"][0])
    new_na = orc.synth_fbm(42, 60, m, npop=4, miss=0.06)           # byte 3: missing also under the imputed code table
    Xna = tpg.FBM.from_numpy(new_na)
    with pytest.raises(ValueError):
        tpg.predict_gt_pca(pca, Xna, None, cols, project_method="none")
    t = tpg.predict_gt_pca(pca, Xna, None, cols, project_method="simple")
    o = orc.predict_gt_pca(opca, new_na, None, cols, project_method="simple")
    assert np.allclose(t, o, rtol=1e-9, atol=1e-9 * np.abs(o).max())
    for lsq in ((1, 2), (1, 2, 3), (2, 5)):
        t = tpg.predict_gt_pca(pca, Xna, None, cols, project_method="least_squares", lsq_pcs=lsq)
        o = orc.predict_gt_pca(opca, new_na, None, cols, project_method="least_squares", lsq_pcs=lsq)
        assert np.allclose(t, o, rtol=1e-8, atol=1e-8 * np.abs(o).max()), lsq
    for bad in ((), (0, 1), (1, 1), (1, 7), (1.5, 2)):
        with pytest.raises(ValueError):
            tpg.predict_gt_pca(pca, Xna, None, cols, project_method="least_squares", lsq_pcs=bad)
    with pytest.raises(NotImplementedError):
        tpg.predict_gt_pca(pca, Xna, None, cols, project_method="OADP")
    XV, xnorm = tpg.oadp_inputs(pca, Xna, None, cols)
    oXV, orss = orc.fbm256_prod_and_rowSumsSq(new_na, None, cols, pca["center"], pca["scale"], pca["v"],
                                              code256=orc.CODE_IMPUTE_PRED)
    assert np.allclose(XV, oXV, rtol=1e-9, atol=1e-9 * np.abs(oXV).max()) and np.allclose(xnorm, orss, rtol=1e-9)


def test_gt_pca_randomSVD_tolerance(tpg):
    # R/gt_pca_randomSVD.R:77-135: same truncated SVD, accepted at the relative residual tol
    n, m, k = 300, 4000, 8
    fbm = orc.synth_fbm(77, n, m, npop=6, miss=0.0)
    keep = np.where((fbm.sum(axis=0) > 0) & (fbm.sum(axis=0) < 2 * n))[0]
    cols = (keep + 1).astype(np.int32)
    X = tpg.FBM.from_numpy(fbm)
    exact = tpg.gt_pca_partialSVD(X, None, cols, k=k)
    K = tpg.pca_gram(tpg.View(X, None, cols, code256=tpg.CODE_IMPUTE_PRED), exact["center"], exact["scale"])
    for tol in (1e-4, 1e-8):
        r = tpg.gt_pca_randomSVD(X, None, cols, k=k, tol=tol)
        assert r["method"] == "randomSVD"
        res = np.linalg.norm(K @ r["u"] - r["u"] * r["d"] ** 2, axis=0)
        assert np.all(res <= tol * r["d"][0] ** 2 * 1.001)
        assert np.allclose(r["d"], exact["d"], rtol=max(10 * tol ** 2, 1e-10))
        assert np.allclose(r["u"].T @ r["u"], np.eye(k), atol=1e-9)
        assert r["square_frobenius"] == exact["square_frobenius"]
    with pytest.raises(ValueError):
        tpg.gt_pca_randomSVD(X, None, cols, k=k, tol=0.0)


# ---------------------------------------------------------------- PCA Gram: weight-class path against the digit path
def _np_gram(g, center, scale):
    Z = (g.astype(np.float64) - center[None, :]) / scale[None, :]
    return Z @ Z.T


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        import os
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *a):
        import os
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("n,m", [(2, 40), (31, 64), (33, 65), (37, 700), (96, 129), (100, 5000), (129, 1000), (333, 20011),
                                 (700, 3001)])
def test_pca_gram_weight_classes_equal_digits_and_numpy(tpg, n, m):
    """S' = sum over weight classes of w_c G_c on the FP4 matrix cores (gramcls.hip) against the digit-split int8
    kernel on the same view and against a float64 numpy Gram; both paths forced, whatever the cost model would pick."""
    g = orc.synth_fbm(11, n, m, npop=5, miss=0.0)
    g = g[:, (g.sum(0) > 0) & (g.sum(0) < 2 * n)]
    X = _X(tpg, g)
    v = tpg.View(X)
    center, scale = tpg.pca_center_scale(v)
    ref = _np_gram(g, center, scale)
    sc = np.abs(ref).max()
    # exact class Gram matrices and double weights; the default fold takes neighbouring classes in groups and adds their
    # small weight differences in FP32 (TPG_GRAM_FOLD64=0; 1e-9 is a few times what it shows), =1 folds every class in FP64;
    # left alone the library picks by the number of classes per block
    for fold64, tol in (("0", 1e-9), ("1", 1e-11)):
        with _env(TPG_GRAM_CLASSES="1", TPG_GRAM_DIGITS=None, TPG_GRAM_FOLD64=fold64):
            Kc = tpg.pca_gram(v, center, scale)
            assert np.array_equal(Kc, Kc.T)
            assert np.abs(Kc - ref).max() <= tol * sc
            assert np.array_equal(tpg.pca_gram(v, center, scale), Kc)  # run-to-run identical (ordered slab sums)
            if fold64 == "0":  # the one-wave-per-SIMD form of the mixed fold (an A/B kept in the library), plain and interleaved
                for kern in ("14", "34"):
                    with _env(TPG_GRAM_KERNEL=kern):
                        K1 = tpg.pca_gram(v, center, scale)
                        assert np.array_equal(K1, K1.T) and np.abs(K1 - ref).max() <= tol * sc
            for qw in ("1", "2", "4"):  # the gather's chunks of 128 individuals per task: the same operand layout
                with _env(TPG_GATHER_QW=qw):
                    assert np.array_equal(tpg.pca_gram(v, center, scale), Kc)
    with _env(TPG_GRAM_DIGITS="1", TPG_GRAM_CLASSES=None):
        Kd = tpg.pca_gram(v, center, scale)
    assert np.abs(Kd - ref).max() <= 1e-6 * sc    # 2^-24 weight rounding


def test_pca_gram_weight_classes_general_center_and_scale(tpg):
    """center is not the column mean, the scale takes a handful of values (few classes) or one value per locus (every
    locus its own class: the path still has to be right when it is forced)."""
    n, m = 150, 2000
    g = orc.synth_fbm(12, n, m, npop=3, miss=0.0)
    X = _X(tpg, g)
    v = tpg.View(X)
    rng = np.random.default_rng(4)
    center = rng.uniform(0.1, 1.9, m)
    for scale in (rng.choice([0.5, 0.75, 1.0, 1.25, 3.0], m), rng.uniform(0.3, 2.0, m)):
        ref = _np_gram(g, center, scale)
        for fold64, tol in (("0", 1e-9), ("1", 1e-11)):
            with _env(TPG_GRAM_CLASSES="1", TPG_GRAM_DIGITS=None, TPG_GRAM_FOLD64=fold64):
                Kc = tpg.pca_gram(v, center, scale)
            assert np.abs(Kc - ref).max() <= tol * np.abs(ref).max()
        Ka = tpg.pca_gram(v, center, scale)  # whatever the cost model picks
        assert np.abs(Ka - ref).max() <= 1e-6 * np.abs(ref).max()


def test_pca_gram_one_class_longer_than_an_exact_fp32_sum(tpg):
    """Every locus the same genotype column: ONE weight class of 4.2 million loci, i.e. more 64-locus blocks than FP32
    accumulators can take (4 x 2^22 = 2^24): the class is folded every 2^14 blocks (every 2^16 by the FP64-fold kernel)."""
    n, m = 40, 64 * 65536 + 64 * 100 + 3
    col = np.array([2, 2, 0, 1, 2, 0, 1, 2, 2, 1] * 4, dtype=np.uint8)[:n]
    Xb = np.empty((n, m), dtype=np.uint8, order="F")
    Xb[:] = col[:, None]
    X = tpg.FBM.from_numpy(Xb)
    del Xb
    v = tpg.View(X)
    center, scale = tpg.pca_center_scale(v)
    z = (col.astype(np.float64) - center[0]) / scale[0]
    ref = m * np.outer(z, z)
    for fold64 in ("0", "1"):
        with _env(TPG_GRAM_CLASSES="1", TPG_GRAM_DIGITS=None, TPG_GRAM_FOLD64=fold64):
            K = tpg.pca_gram(v, center, scale)
        assert np.abs(K - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("n,m,subset", [(300, 1000, False), (1000, 2500, False), (256, 1024, True)])
def test_view_pair_layouts_serve_every_consumer(tpg, n, m, subset):
    """A pair is packed as (L + the FP4 layout of the pairwise kernel, L): whatever contracts over loci on the 2-bit T
    layout (individual counts, the digit Gram, the sweeps) gets T made from L, and a pair that goes through the generic
    pack kernel (row subset) gets T and has its FP4 layout made on first use.  Everything must equal what the two views
    created one by one give."""
    fbi = orc.synth_fbm(21, n, m, npop=4, miss=0.05, imputed_bytes=True)
    X = tpg.FBM.from_numpy(fbi)
    rng = np.random.default_rng(3)
    rows = (np.sort(rng.permutation(n)[: n - 7]) + 1).astype(np.int32) if subset else None
    va, vb = tpg.View.pair(X, rows)
    sa, sb = tpg.View(X, rows, code256=None), tpg.View(X, rows, code256=tpg.CODE_IMPUTE_PRED)
    nn = va.n
    pa, ps = tpg.Pairwise(X.ctx, nn), tpg.Pairwise(X.ctx, nn)
    pa.accumulate(va)
    ps.accumulate(sa)
    ca, cs = pa.counts(), ps.counts()
    for k in ca:
        assert np.array_equal(ca[k], cs[k]), k
    assert np.array_equal(tpg.indiv_counts(va), tpg.indiv_counts(sa))
    assert np.array_equal(tpg.indiv_counts(vb), tpg.indiv_counts(sb))
    cnt = tpg.loci_counts(vb)
    alt = cnt[:, 1] + 2 * cnt[:, 2]
    keep = (np.where((alt > 0) & (alt < 2 * nn))[0] + 1).astype(np.int32)
    vb2, sb2 = tpg.View.pair(X, rows, keep)[1], tpg.View(X, rows, keep, code256=tpg.CODE_IMPUTE_PRED)
    center, scale = tpg.pca_center_scale(sb2)
    with _env(TPG_GRAM_DIGITS="1", TPG_GRAM_CLASSES=None):
        assert np.array_equal(tpg.pca_gram(vb2, center, scale), tpg.pca_gram(sb2, center, scale))
    assert np.array_equal(va.unpack(), sa.unpack()) and np.array_equal(vb.unpack(), sb.unpack())


def test_pca_more_than_52_components(tpg):
    """k is free in the reference (R/gt_pca_partialSVD.R:70-89).  Beyond 52 components the eigen step runs in batches of
    26 with explicit deflation: d against LAPACK, u orthonormal and an eigenbasis of the device Gram, v = Z'u / d."""
    n, m, k = 300, 4000, 100
    fbm = orc.synth_fbm(73, n, m, npop=6, miss=0.02, imputed_bytes=True)
    dec = np.where(fbm > 3, fbm - 4, fbm)
    cols = (np.where((dec.sum(axis=0) > 0) & (dec.sum(axis=0) < 2 * n))[0] + 1).astype(np.int32)
    X = tpg.FBM.from_numpy(fbm)
    o = orc.gt_pca_partialSVD(fbm, None, cols, k=k)
    r = tpg.gt_pca_partialSVD(X, None, cols, k=k)
    assert np.allclose(r["d"], o["d"], rtol=1e-6, atol=0)
    assert np.all(np.diff(r["d"]) <= 0)
    assert np.abs(r["u"].T @ r["u"] - np.eye(k)).max() <= 1e-9
    v = tpg.View(X, None, cols, code256=tpg.CODE_IMPUTE_PRED)
    K = tpg.pca_gram(v, r["center"], r["scale"])
    assert np.abs(K @ r["u"] - r["u"] * r["d"] ** 2).max() <= 1e-9 * r["d"][0] ** 2
    Z = (dec[:, cols - 1] - r["center"]) / r["scale"]
    assert np.abs(r["v"] - (Z.T @ r["u"]) / r["d"]).max() <= 1e-9 * np.abs(r["v"]).max()
    # the leading scores are the oracle's (the trailing ones sit in the bulk of the spectrum, where vectors of close
    # eigenvalues mix: compared as a subspace)
    so, sr = o["u"][:, :20] * o["d"][:20], _align_sign(r["u"][:, :20] * r["d"][:20], o["u"][:, :20] * o["d"][:20])
    assert np.max(np.abs(sr - so)) <= 1e-6 * np.max(np.abs(so))
    P_o, P_r = o["u"] @ o["u"].T, r["u"] @ r["u"].T
    gap = (o["d"][k - 1] ** 2 - np.linalg.eigvalsh(K)[::-1][k]) / o["d"][0] ** 2
    assert np.abs(P_o - P_r).max() <= 1e-9 / gap
    lam, U = tpg.sym_eig_topk(K, 60)  # the stand-alone entry point takes the batched route too
    assert np.allclose(lam, o["d"][:60] ** 2, rtol=1e-9)


def test_code_dosage_table_is_accepted_when_no_dosage_byte_occurs(tpg):
    """bigsnpr's CODE_DOSAGE (bytes 7 .. 207 = 0.00 .. 2.00) is one of the two tables gt_uses_imputed accepts
    (R/gt_has_imputed.R:54-55).  tidypopgen's own imputations only ever write bytes 4 .. 6, for which it decodes like
    CODE_IMPUTE_PRED: same results; an FBM that does hold a fractional dosage is refused, not rounded."""
    code_dosage = np.full(256, np.nan)
    code_dosage[:3] = [0, 1, 2]
    code_dosage[4:7] = [0, 1, 2]
    code_dosage[7:208] = np.round(np.arange(201) * 0.01, 2)
    n, m = 120, 1500
    fbm = orc.synth_fbm(74, n, m, npop=3, miss=0.05, imputed_bytes=True)
    X = tpg.FBM.from_numpy(fbm)
    a = tpg.loci_alt_freq(tpg.FBM.from_numpy(fbm, code256=code_dosage))
    b = tpg.loci_alt_freq(tpg.FBM.from_numpy(fbm, code256=tpg.CODE_IMPUTE_PRED))
    assert np.array_equal(a, b)
    va, vb = tpg.View(X, code256=code_dosage), tpg.View(X, code256=tpg.CODE_IMPUTE_PRED)
    assert np.array_equal(va.unpack(), vb.unpack())
    bad = fbm.copy(order="F")
    bad[5, 7] = 57  # dosage 0.50
    with pytest.raises(tpg._lib.TpgError) as e:
        tpg.View(tpg.FBM.from_numpy(bad), code256=code_dosage)
    assert e.value.code == 3


def test_host_bind_near_device_in_a_child_process():
    """tpg_host_bind_near_device (include/tpg.h): in a child process (the binding is the calling thread's for good), the node it
    reports is the one sysfs gives for the GPU, the thread's CPUs afterwards are CPUs of that node, a thread started later
    inherits them; -1 = nothing changed; a device that does not exist is an error"""
    import subprocess
    import sys

    code = r"""
import os, sys, threading
sys.path.insert(0, os.getcwd())
import tidypopgen_amd as tpg
before = os.sched_getaffinity(0)
node = tpg.bind_host_near_device(0)
after = os.sched_getaffinity(0)
seen = []
t = threading.Thread(target=lambda: seen.append(os.sched_getaffinity(0)))
t.start(); t.join()
if node < 0:
    assert after == before, "affinity changed although nothing was reported"
else:
    cpus = set()
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    assert after == before & cpus and len(after) >= 16 and seen[0] == after, (node, len(after))
try:
    tpg.bind_host_near_device(99)
    raise SystemExit("device 99 accepted")
except tpg._lib.TpgError:
    pass
X = tpg.FBM.synth(3, 100, 500, npop=3)
assert tpg.snp_ibs(X).shape == (100, 100)  # the library works from the bound thread
print("node", node, "cpus", len(after))
"""
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.startswith("node ")
