"""Randomised shapes through the whole C ABI against the CPU oracle: ragged sizes around the tile edges of every
kernel (32 / 64 / 128 individuals, 128 loci), column-only subsets (the fast pack path), row + column subsets (the
generic path), bytes outside the code table."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1), (2, 3), (31, 127), (32, 128), (33, 129), (63, 300), (64, 256), (65, 511), (96, 1000), (127, 77),
          (128, 640), (129, 130), (160, 999), (200, 2049), (224, 384), (257, 1300)]


@pytest.fixture(scope="module")
def tpg():
    import tidypopgen_amd as t

    t.default_context()
    return t


@pytest.mark.parametrize("n,m", SHAPES)
def test_random_shape_end_to_end(tpg, n, m):
    rng = np.random.default_rng(n * 1000 + m)
    G = int(rng.integers(1, min(n, 7) + 1))
    fbm = orc.synth_fbm(n + m, n, m, npop=G, miss=float(rng.choice([0.0, 0.05, 0.4])))
    # a few bytes the code table maps to NA (the raw path treats everything > 2 as missing)
    k = max(1, (n * m) // 50)
    fbm[rng.integers(0, n, k), rng.integers(0, m, k)] = rng.integers(3, 256, k)
    gid = rng.integers(0, G, n).astype(np.int32)
    X = tpg.FBM.from_numpy(fbm)
    for rows, cols in ((None, None),
                       (None, np.sort(rng.permutation(m)[: max(1, m // 2)] + 1).astype(np.int32)),
                       ((rng.permutation(n)[: max(1, (2 * n) // 3)] + 1).astype(np.int32),
                        (rng.permutation(m)[: max(1, m // 3)] + 1).astype(np.int32))):
        g = gid if rows is None else gid[rows - 1]
        nn = n if rows is None else len(rows)
        # counts: exact
        v = tpg.View(X, rows, cols)
        ref = np.where(fbm > 2, 3, fbm)[np.ix_(np.arange(n) if rows is None else rows - 1,
                                                np.arange(m) if cols is None else cols - 1)]
        assert np.array_equal(v.unpack(), ref)
        cnt = tpg.loci_counts(v)
        assert np.array_equal(cnt, np.stack([(ref == c).sum(axis=0) for c in range(4)], axis=1))
        # pairwise statistics
        with np.errstate(invalid="ignore", divide="ignore"):
            o_ibs = orc.snp_ibs(fbm, rows, cols, type="raw_counts")
            o_king = orc.snp_king(fbm, rows, cols)
            o_as = orc.snp_allele_sharing(fbm, rows, cols)
        t_ibs = tpg.snp_ibs(X, rows, cols, type="raw_counts")
        assert np.array_equal(t_ibs["ibs"], o_ibs["ibs"]) and np.array_equal(t_ibs["valid_n"], o_ibs["valid_n"])
        assert np.allclose(tpg.snp_king(X, rows, cols), o_king, rtol=1e-12, atol=0, equal_nan=True)
        assert np.allclose(tpg.snp_allele_sharing(X, rows, cols), o_as, rtol=1e-12, atol=0, equal_nan=True)
        # per-locus and per-population statistics
        with np.errstate(invalid="ignore", divide="ignore"):
            assert np.array_equal(tpg.loci_alt_freq(X, rows, cols), orc.loci_alt_freq(fbm, rows, cols), equal_nan=True)
            if G >= 2:
                for method in ("Hudson", "WC84", "Nei87"):
                    o = orc.pairwise_pop_fst(fbm, rows, cols, g, G, method=method)["fst_tot"]
                    t = tpg.pairwise_pop_fst(X, rows, cols, g, G, method=method)["fst_tot"]
                    assert np.allclose(t, o, rtol=1e-10, atol=1e-13, equal_nan=True)
        assert nn == v.n
