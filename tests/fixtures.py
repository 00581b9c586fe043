"""Literal matrices and data files that the reference's own tests use for the
hot path (facts / data, cited per item), shared by the oracle and GPU tests."""
import os

import numpy as np

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
nan = np.nan

# tests/testthat/test_snp_ibs.R:6-10 and test_snp_king.R:6-10 (3 x 6, no missing)
IBS_3x6 = np.array([[1, 1, 0, 1, 1, 0],
                    [2, 1, 0, 0, 0, 0],
                    [2, 2, 0, 0, 1, 1]], dtype=float)

# tests/testthat/test_loci_freq.R:6-10, test_loci_missingness.R:5-9 (3 x 6 with NA)
FREQ_3x6 = np.array([[1, 1, 0, 1, 1, 2],
                     [2, 1, 0, nan, 0, nan],
                     [2, 2, 0, 0, 1, nan]], dtype=float)

# tests/testthat/test_loci_freq.R:66-70
FREQ2_3x6 = np.array([[1, 1, 0, 1, 1, 2],
                      [2, 1, 0, nan, 0, nan],
                      [2, 2, 0, nan, 1, nan]], dtype=float)

# tests/testthat/test_pairwise_allele_sharing.R:3-7 (3 x 6 with NA)
AS_3x6 = np.array([[1, 1, 0, 1, 1, 0],
                   [2, 1, 0, nan, 0, 0],
                   [2, nan, 0, 0, 1, 1]], dtype=float)

# tests/testthat/test_pairwise_grm.R:3-11, test_pairwise_pop_fst.R:1-9,57-65 (7 x 6 with NA)
FST_7x6 = np.array([[1, 1, 0, 1, 1, 0],
                    [2, 1, 0, nan, 0, 0],
                    [2, nan, 0, 0, 1, 1],
                    [1, 0, 0, 1, 0, 0],
                    [1, 2, 0, 1, 2, 1],
                    [0, 0, 0, 0, nan, 1],
                    [0, 1, 1, 0, 1, nan]], dtype=float)
# population = pop1,pop1,pop1,pop2,pop2,pop2,pop2 (test_pairwise_pop_fst.R:68, 210)
FST_GROUPS_2 = np.array([0, 0, 0, 1, 1, 1, 1], dtype=np.int32)
# population = pop1,pop1,pop2,pop2,pop1,pop3,pop3 (test_pairwise_pop_fst.R:12)
FST_GROUPS_3 = np.array([0, 0, 1, 1, 0, 2, 2], dtype=np.int32)

# tests/testthat/test_pairwise_pop_fst.R:119-127 (first locus monomorphic)
FST_MONO_7x6 = FST_7x6.copy()
FST_MONO_7x6[:, 0] = 2
FST_MONO_7x6[:, 1] = [1, 1, nan, 0, 2, 0, 1]

# tests/testthat/test_pairwise_pop_fst.R:147-155 (locus 1 missing in all of pop1)
FST_MISSPOP_7x6 = np.array([[nan, 1, 0, 1, 1, 0],
                            [nan, 1, 0, nan, 0, 0],
                            [nan, nan, 0, 0, 1, 1],
                            [2, 0, 0, 1, 0, 0],
                            [1, 2, 0, 1, 2, 1],
                            [2, 0, 0, 0, nan, 1],
                            [2, 1, 1, 0, 1, nan]], dtype=float)
# :166-174 the same data with that locus removed
FST_MISSPOP_7x5 = FST_MISSPOP_7x6[:, 1:].copy()


def families_fbm():
    return orc.read_bed(os.path.join(GOLDEN, "related", "families.bed"), 12, 961)


def lobster_fbm():
    return orc.read_bed(os.path.join(GOLDEN, "lobster", "lobster.bed"), 176, 79)


def plink_mibs():
    return np.loadtxt(os.path.join(GOLDEN, "related", "test_plinkIBS.mibs"))


def king_kin0_matrix():
    """12 x 12 matrix filled as tests/testthat/test_snp_king.R:214-226 does."""
    rows = np.loadtxt(os.path.join(GOLDEN, "related", "test_king.kin0"))
    K = np.full((12, 12), np.nan)
    for r in rows:
        x, y = int(r[1]) - 1, int(r[2]) - 1
        K[x, y] = r[7]
        K[y, x] = r[7]
    np.fill_diagonal(K, 0.5)
    return K


def scikit(name):
    return np.loadtxt(os.path.join(GOLDEN, "fst_scikit-allel", name + ".txt"))


def king_r(X):
    """tests/testthat/test_snp_king.R:134-151 (the reference's in-test R restatement)."""
    X0 = np.nan_to_num((X == 0).astype(float))
    X1 = np.nan_to_num((X == 1).astype(float))
    X2 = np.nan_to_num((X == 2).astype(float))
    num = X1 @ X1.T - 2 * (X0 @ X2.T + X2 @ X0.T)
    valid = (~np.isnan(X)).astype(float)
    Ni = X1 @ valid.T
    Nj = Ni.T
    mn = np.minimum(Ni, Nj)
    with np.errstate(invalid="ignore", divide="ignore"):
        return num / (2 * mn) + 0.5 - 0.25 * (Ni + Nj) / mn


def matching(dos):
    """Allele-sharing by its definition (R/snp_allele_sharing.R:3-7; what
    hierfstat::matching computes): per locus 1 if both homozygous for the same
    allele, 0 if homozygous for different alleles, 1/2 if at least one is
    heterozygous; averaged over loci where both are typed."""
    n = dos.shape[0]
    out = np.full((n, n), np.nan)
    for a in range(n):
        for b in range(n):
            ok = ~np.isnan(dos[a]) & ~np.isnan(dos[b])
            if ok.sum() == 0:
                continue
            x, y = dos[a, ok], dos[b, ok]
            s = np.where((x == 1) | (y == 1), 0.5, np.where(x == y, 1.0, 0.0))
            out[a, b] = s.mean()
    return out
