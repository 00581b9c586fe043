"""CPU ORACLE -- test infrastructure, NOT the product.

Python face of ``oracle/tpg_oracle.c`` plus restatements of the R-level drivers
that sit around the reference's native kernels.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / timed CPU baseline.

Parity pinning: ``tests/test_oracle_golden.py`` checks these functions against
the golden vectors of the reference's own tests (PLINK .mibs, KING .kin0,
scikit-allel Fst files, literal matrices).  The reference itself cannot be
built or run here (R + Rcpp + bigstatsr are absent), so there is no
``oracle/_ref``.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  Third-party arithmetic that is not under the reference
tree (bigstatsr / bigsnpr / bigparallelr) is marked "(recalled)".
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import warnings

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtpg_oracle.so")

NA = float("nan")

# bigsnpr code tables (recalled; R/gt_has_imputed.R:101-106 switches between them)
CODE_012 = np.full(256, np.nan)
CODE_012[:3] = [0.0, 1.0, 2.0]
CODE_IMPUTE_PRED = np.full(256, np.nan)
CODE_IMPUTE_PRED[:3] = [0.0, 1.0, 2.0]
CODE_IMPUTE_PRED[4:7] = [0.0, 1.0, 2.0]


def build(force: bool = False) -> str:
    """Compile the C oracle (gcc) if needed and return the .so path."""
    src = os.path.join(_HERE, "tpg_oracle.c")
    hdr = os.path.join(_HERE, "..", "tidypopgen_amd", "csrc", "synth_common.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        # TPG_ORACLE_SO: another build of the same source (tests/test_host_sanitizers.py: -fsanitize=address,undefined)
        _lib = C.CDLL(os.environ.get("TPG_ORACLE_SO") or build())
        _lib.orc_square_frobenius.restype = C.c_double
        _lib.orc_pca_center_scale_gram.restype = C.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _fbm(fbm):
    fbm = np.asarray(fbm)
    assert fbm.dtype == np.uint8 and fbm.ndim == 2
    if not fbm.flags.f_contiguous:
        fbm = np.asfortranarray(fbm)
    return fbm


def _ind(x, default_len):
    if x is None:
        return np.arange(1, default_len + 1, dtype=np.int32)
    return np.ascontiguousarray(x, dtype=np.int32)


def _view(fbm, rowInd, colInd):
    fbm = _fbm(fbm)
    r = _ind(rowInd, fbm.shape[0])
    c = _ind(colInd, fbm.shape[1])
    return fbm, r, c


def _args(fbm, r, c):
    return (_p(fbm, C.c_uint8), C.c_int64(fbm.shape[0]), _p(r, C.c_int32), C.c_int(len(r)),
            _p(c, C.c_int32), C.c_int(len(c)))


def _d(a):
    return _p(a, C.c_double)


# --------------------------------------------------------------------------
# helpers that exist on the R side

def fbm_from_genotypes(g) -> np.ndarray:
    """Dosage matrix (NaN / negative = missing) -> FBM bytes, NA = 3
    (R/gen_tibble_fbm.R:185-194: missing is written as max_ploidy + 1)."""
    g = np.asarray(g, dtype=float)
    out = np.where(np.isnan(g), 3, g).astype(np.uint8)
    return np.asfortranarray(out)


def read_bed(path: str, n: int, m: int) -> np.ndarray:
    """PLINK .bed (SNP-major) -> FBM bytes via bigsnpr's getCode() table
    (third-party, recalled): 2-bit 00,01,10,11 -> byte 2,3,1,0."""
    raw = np.fromfile(path, dtype=np.uint8)
    assert raw[0] == 0x6C and raw[1] == 0x1B and raw[2] == 0x01, "not a SNP-major .bed"
    bpl = (n + 3) // 4
    body = raw[3:3 + bpl * m].reshape(m, bpl)
    codes = np.stack([(body >> s) & 3 for s in (0, 2, 4, 6)], axis=2).reshape(m, bpl * 4)[:, :n]
    table = np.array([2, 3, 1, 0], dtype=np.uint8)
    return np.asfortranarray(table[codes].T)


def split_len(total_len: int, nb: int):
    """bigparallelr::split_len (recalled) -> (lower, upper), 1-based inclusive."""
    lo = np.zeros(nb, dtype=np.int32)
    up = np.zeros(nb, dtype=np.int32)
    lib().orc_split_len(C.c_int(total_len), C.c_int(nb), _p(lo, C.c_int32), _p(up, C.c_int32))
    return lo, up


def cut_by_size(m: int, block_size: int):
    """CutBySize, R/local_reimplementations.R:13-15."""
    nb = int(math.ceil(m / block_size))
    return split_len(m, nb)


def block_size_default(n: int, ncores: int = 1) -> int:
    """bigstatsr::block_size (recalled): floor(block.sizeGB * 1024^3 / (8 n ncores)),
    block.sizeGB = 1, at least 1."""
    return max(1, int(math.floor(1024.0 ** 3 / (8.0 * n * ncores))))


# --------------------------------------------------------------------------
# native per-block kernels

def increment_ibs_counts(K, K2, fbm, rowInd, colInd):
    """src/snp_ibs.cpp:22-74"""
    fbm, r, c = _view(fbm, rowInd, colInd)
    lib().orc_increment_ibs_counts(*_args(fbm, r, c), _d(K), _d(K2))


def increment_king_numerator(K, N_Aa_i, fbm, rowInd, colInd):
    """src/snp_king.cpp:21-74"""
    fbm, r, c = _view(fbm, rowInd, colInd)
    lib().orc_increment_king_numerator(*_args(fbm, r, c), _d(K), _d(N_Aa_i))


def increment_as_counts(K, K2, fbm, rowInd, colInd, pad_quirk=False):
    """src/snp_as.cpp:22-67 (pad_quirk: reference quirk Q1, see tpg_oracle.c)"""
    fbm, r, c = _view(fbm, rowInd, colInd)
    lib().orc_increment_as_counts(*_args(fbm, r, c), C.c_int(int(pad_quirk)), _d(K), _d(K2))


def alt_freq_dip_pseudo_cpp(fbm, rowInd, colInd, ploidy, as_counts=False, code256=CODE_012):
    """src/alt_freq_dip_pseudo_cpp.cpp:8-58 -> (m, 2) matrix"""
    fbm, r, c = _view(fbm, rowInd, colInd)
    ploidy = np.ascontiguousarray(ploidy, dtype=float)
    out = np.zeros((len(c), 2), order="F")
    lib().orc_alt_freq_dip_pseudo(*_args(fbm, r, c), _d(np.ascontiguousarray(code256)), _d(ploidy),
                                  C.c_int(int(as_counts)), _d(out))
    return out


def grouped_alt_freq_dip_pseudo_cpp(fbm, rowInd, colInd, groupIds, ngroups, ploidy, as_counts=False,
                                    code256=CODE_012):
    """src/grouped_alt_freq_dip_pseudo_cpp.cpp:8-58 -> (m, 2G) matrix"""
    fbm, r, c = _view(fbm, rowInd, colInd)
    ploidy = np.ascontiguousarray(ploidy, dtype=float)
    gid = np.ascontiguousarray(groupIds, dtype=np.int32)
    out = np.zeros((len(c), 2 * ngroups), order="F")
    lib().orc_grouped_alt_freq_dip_pseudo(*_args(fbm, r, c), _d(np.ascontiguousarray(code256)),
                                          _p(gid, C.c_int32), C.c_int(ngroups), _d(ploidy),
                                          C.c_int(int(as_counts)), _d(out))
    return out


def grouped_missingness_cpp(fbm, rowInd, colInd, groupIds, ngroups, code256=CODE_012):
    """src/grouped_missingness_cpp.cpp:8-33 -> (m, G) matrix"""
    fbm, r, c = _view(fbm, rowInd, colInd)
    gid = np.ascontiguousarray(groupIds, dtype=np.int32)
    out = np.zeros((len(c), ngroups), order="F")
    lib().orc_grouped_missingness(*_args(fbm, r, c), _d(np.ascontiguousarray(code256)),
                                  _p(gid, C.c_int32), C.c_int(ngroups), _d(out))
    return out


def grouped_summaries_dip_pseudo_cpp(fbm, rowInd, colInd, groupIds, ngroups, ploidy, code256=CODE_012):
    """src/grouped_summaries_dip_pseudo_cpp.cpp:11-63 -> dict of four (m, G) matrices"""
    fbm, r, c = _view(fbm, rowInd, colInd)
    ploidy = np.ascontiguousarray(ploidy, dtype=float)
    gid = np.ascontiguousarray(groupIds, dtype=np.int32)
    outs = [np.zeros((len(c), ngroups), order="F") for _ in range(4)]
    lib().orc_grouped_summaries_dip_pseudo(*_args(fbm, r, c), _d(np.ascontiguousarray(code256)),
                                           _p(gid, C.c_int32), C.c_int(ngroups), _d(ploidy),
                                           *[_d(o) for o in outs])
    return dict(freq_alt=outs[0], freq_ref=outs[1], n=outs[2], het_obs=outs[3])


def gt_ind_hetero(fbm, rowInd=None, colInd=None, code256=CODE_012):
    """src/gt_ind_hetero.cpp:11-42 -> (2, n) int matrix"""
    fbm, r, c = _view(fbm, rowInd, colInd)
    out = np.zeros((2, len(r)), dtype=np.int32, order="F")
    lib().orc_gt_ind_hetero(*_args(fbm, r, c), _d(np.ascontiguousarray(code256)), _p(out, C.c_int32))
    return out


def indiv_het_obs(fbm, rowInd=None, colInd=None, as_counts=False, code256=CODE_012):
    """R/indiv_het_obs.R:42-75 on top of gt_ind_hetero: het_n / (ncol(X) - na_n) -- ncol(X) is the column count of
    the whole FBM (:69), as the reference writes it -- or the (n, 2) count matrix {het_n, na_n}."""
    fbm_, r, c = _view(fbm, rowInd, colInd)
    cnt = gt_ind_hetero(fbm_, r, c, code256)
    if as_counts:
        return cnt.T.copy()
    with np.errstate(invalid="ignore", divide="ignore"):
        return cnt[0] / (fbm_.shape[1] - cnt[1])


def gt_pi_diploid(fbm, rowInd=None, colInd=None, code256=CODE_012):
    """src/gt_pi_diploid.cpp:7-38"""
    fbm, r, c = _view(fbm, rowInd, colInd)
    out = np.zeros(len(c))
    lib().orc_gt_pi_diploid(*_args(fbm, r, c), _d(np.ascontiguousarray(code256)), _d(out))
    return out


def gt_grouped_pi_diploid(fbm, rowInd, colInd, groupIds, ngroups, code256=CODE_012):
    """src/gt_grouped_pi_diploid.cpp:7-42"""
    fbm, r, c = _view(fbm, rowInd, colInd)
    gid = np.ascontiguousarray(groupIds, dtype=np.int32)
    pi, n = np.zeros((len(c), ngroups), order="F"), np.zeros((len(c), ngroups), order="F")
    lib().orc_gt_grouped_pi_diploid(*_args(fbm, r, c), _d(np.ascontiguousarray(code256)), _p(gid, C.c_int32),
                                    C.c_int(ngroups), _d(pi), _d(n))
    return dict(pi=pi, n=n)


def grouped_genotype_counts(fbm, rowInd, colInd, groupIds, ngroups, code256=CODE_012):
    """the 3 x ngroups genotype table gt_grouped_hwe fills per locus (src/hwe.cpp:238-250) -> (3, m, G) int32"""
    fbm, r, c = _view(fbm, rowInd, colInd)
    g = np.asarray(code256)[fbm[np.ix_(r - 1, c - 1)]]  # (n, m) decoded, NaN = missing (indices are 1-based)
    gid = np.asarray(groupIds)
    out = np.zeros((3, len(c), ngroups), dtype=np.int32)
    for k in range(3):
        hit = (g == k)
        for grp in range(ngroups):
            out[k, :, grp] = hit[gid == grp].sum(axis=0)
    return out


def compute_np_mn(n):
    """src/compute_np_mn.cpp:8-34"""
    n = np.asarray(n, dtype=float)
    ok = ~np.isnan(n)
    np_ = ok.sum(axis=1).astype(float)
    with np.errstate(divide="ignore"):
        denom = np.where(ok, 1.0 / n, 0.0).sum(axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        mn = np.where(denom > 0.0, np_ / denom, np.nan)
    return np_, mn


def pop_global_stats(fbm, rowInd, colInd, groupIds, ngroups, ploidy=None, by_locus=False, code256=CODE_012):
    """R/pop_global_stats.R:113-212, statement by statement (hierfstat::basic.stats arithmetic)"""
    fbm_, r, c = _view(fbm, rowInd, colInd)
    ploidy = np.full(len(r), 2.0) if ploidy is None else np.asarray(ploidy, dtype=float)
    if not np.all(ploidy == 2.0):
        raise ValueError("pop_global_stats only works on diploid data")  # stopifnot_diploid, :117
    with np.errstate(invalid="ignore", divide="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        pf = grouped_summaries_dip_pseudo_cpp(fbm, rowInd, colInd, groupIds, ngroups, ploidy, code256)
        n = pf["n"] / 2                                   # :146
        sHo = pf["het_obs"]                               # :148
        mHo = np.nanmean(sHo, axis=1)                     # :149
        sp2 = pf["freq_alt"] ** 2 + pf["freq_ref"] ** 2   # :151
        np_, mn = compute_np_mn(n)                        # :166-168
        msp2 = np.nanmean(sp2, axis=1)                    # :170
        mp2 = pf["freq_alt"].mean(axis=1) ** 2 + pf["freq_ref"].mean(axis=1) ** 2  # :171
        mHs = mn / (mn - 1) * (1 - msp2 - mHo / 2 / mn)   # :172
        Ht = 1 - mp2 + mHs / mn / np_ - mHo / 2 / mn / np_  # :173
        mFis = 1 - mHo / mHs                              # :174
        Dst = Ht - mHs
        Dstp = np_ / (np_ - 1) * Dst
        Htp = mHs + Dstp
        Fst = Dst / Ht
        Fstp = Dstp / Htp
        Dest = Dstp / (1 - mHs)
        res = np.column_stack([mHo, mHs, Ht, Dst, Htp, Dstp, Fst, Fstp, mFis, Dest])
        if by_locus:
            return res
        res = np.where(np.isinf(res), np.nan, res)        # :203
        overall = np.nanmean(res, axis=0)                 # :204
        overall[6] = overall[3] / overall[2]
        overall[7] = overall[5] / overall[4]
        overall[8] = 1 - overall[0] / overall[1]
        overall[9] = overall[5] / (1 - overall[1])
    return overall


def _as_blocks(Mij, groupIds, ngroups):
    """the block part shared by pop_fst and pop_fis_wg17: R/pop_fst.R:40-63"""
    Mij = np.array(Mij, dtype=float)
    np.fill_diagonal(Mij, np.nan)
    gid = np.asarray(groupIds)
    wil = [np.where(gid == z)[0] for z in range(ngroups)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        Fsts = np.array([np.nanmean(Mij[np.ix_(w, w)]) if len(w) else np.nan for w in wil])
        Mb = 0.0
        for i in range(1, ngroups):
            for j in range(i):
                Mb = Mb + np.nanmean(Mij[np.ix_(wil[i], wil[j])])
        with np.errstate(invalid="ignore", divide="ignore"):
            Mb = Mb * 2 / (ngroups * (ngroups - 1))
    return wil, Fsts, Mb


def pop_fst(allele_sharing_mat, groupIds, ngroups, include_global=False):
    """R/pop_fst.R:31-76"""
    _, Fsts, Mb = _as_blocks(allele_sharing_mat, groupIds, ngroups)
    with np.errstate(invalid="ignore", divide="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        fst = (Fsts - Mb) / (1 - Mb)
        if include_global:
            fst = np.append(fst, np.nanmean(fst))
    return fst


def pop_fis_wg17(allele_sharing_mat, groupIds, ngroups, include_global=False):
    """R/pop_fis.R:136-197"""
    Mii = np.diag(np.asarray(allele_sharing_mat, dtype=float)) * 2 - 1
    wil, Fsts, _ = _as_blocks(allele_sharing_mat, groupIds, ngroups)
    with np.errstate(invalid="ignore", divide="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        fis = np.array([np.nanmean((Mii[w] - Fsts[g]) / (1 - Fsts[g])) if len(w) else np.nan for g, w in enumerate(wil)])
        if include_global:
            fis = np.append(fis, np.nanmean(fis))
    return fis


def windows_stats_generic(x, chromosome, position=None, operator="mean", window_size=None, step_size=None,
                          size_unit="snp", min_loci=1, complete=False):
    """R/windows_stats_generic.R:47-184 with runner::sum_run / mean_run (third party, not in the checkout; its
    window rule restated from its documentation: the window ending at `at` holds the elements whose index lies in
    (at - k, at]; na_pad = TRUE returns NA for a window reaching outside the index range; na_rm = TRUE drops NA
    and an empty window is NA).  Pinned by the values tests/testthat/test_window_stats_generic.R:1-86 expects.
    Plain loops on purpose."""
    x = np.asarray(x, dtype=float)
    chromosome = np.asarray(chromosome)
    chroms = []
    for ch in chromosome:
        if ch not in chroms:
            chroms.append(ch)
    out = dict(chromosome=[], start=[], end=[], stat=[], n_loci=[])
    for ch in chroms:                                               # :113
        sel = chromosome == ch
        x_sub = x[sel]
        pos = np.asarray(position, dtype=float)[sel] if size_unit == "bp" else np.arange(1, len(x_sub) + 1, dtype=float)
        r = (math.ceil(pos.min() / window_size), math.ceil(pos.max() / window_size))   # :123
        at = r[0] * window_size
        while at <= r[1] * window_size + 1e-9:                       # seq(from, to, by = step_size), :124-128
            inside = (pos > at - window_size) & (pos <= at)
            vals = x_sub[inside]
            vals = vals[~np.isnan(vals)]
            incomplete = complete and (at - window_size + 1 < pos[0] or at > pos[-1])
            if incomplete:
                stat, n_loci = np.nan, np.nan
            else:
                n_loci = float(len(vals))
                stat = np.nan if len(vals) == 0 else (vals.sum() if operator == "sum" else vals.sum() / len(vals))
            out["chromosome"].append(ch); out["start"].append(at - window_size + 1); out["end"].append(at)
            out["stat"].append(stat); out["n_loci"].append(n_loci)
            at += step_size
    res = {k: np.array(v) for k, v in out.items()}
    with np.errstate(invalid="ignore"):
        res["stat"] = np.where(res["n_loci"] < min_loci, np.nan, res["stat"])       # :179
    return res


def windows_pairwise_pop_fst(fbm, rowInd, colInd, groupIds, ngroups, chromosome, position=None, ploidy=None,
                             window_size=None, step_size=None, size_unit="snp", min_loci=1, complete=False,
                             code256=CODE_012):
    """R/windows_pairwise_pop_fst.R:49-118, type = "matrix" -> dict(chromosome, start, end, fst (nw, P))"""
    nd = pairwise_pop_fst(fbm, rowInd, colInd, groupIds, ngroups, ploidy, method="Hudson", return_num_dem=True,
                          code256=code256)                          # :62-65, Hudson whatever `method` says
    num, den = nd["Fst_by_locus_num"], nd["Fst_by_locus_den"]
    cols = []
    first = None
    for c in range(num.shape[1]):
        wn = windows_stats_generic(num[:, c], chromosome, position, "mean", window_size, step_size, size_unit, min_loci, complete)
        wd = windows_stats_generic(den[:, c], chromosome, position, "mean", window_size, step_size, size_unit, min_loci, complete)
        first = first or wn
        with np.errstate(invalid="ignore", divide="ignore"):
            cols.append(wn["stat"] / wd["stat"])
    return dict(chromosome=first["chromosome"], start=first["start"], end=first["end"], fst=np.column_stack(cols))


def nwise_pop_pbs(fbm, rowInd, colInd, groupIds, ngroups, ploidy=None, fst_method="Hudson", code256=CODE_012):
    """R/nwise_pop_pbs.R:36-156 (type = "matrix"): columns per triplet in combn(levels, 3) order:
    pbs_1, pbs_2, pbs_3, pbsn1_1, pbsn1_2, pbsn1_3"""
    fst = pairwise_pop_fst(fbm, rowInd, colInd, groupIds, ngroups, ploidy, method=fst_method, by_locus=True,
                           code256=code256)["fst_locus"]
    pairs = combn2(ngroups)
    col = {(int(a), int(b)): k for k, (a, b) in enumerate(pairs.T)}
    cols = []
    with np.errstate(invalid="ignore", divide="ignore"):
        for a in range(1, ngroups + 1):
            for b in range(a + 1, ngroups + 1):
                for c in range(b + 1, ngroups + 1):
                    fst12, fst13, fst23 = fst[:, col[(a, b)]], fst[:, col[(a, c)]], fst[:, col[(b, c)]]
                    t12, t13, t23 = -np.log(1 - fst12), -np.log(1 - fst13), -np.log(1 - fst23)   # :138-140
                    pbs_1 = (t12 + t13 - t23) / 2
                    pbs_2 = (t12 + t23 - t13) / 2
                    pbs_3 = (t13 + t23 - t12) / 2
                    cols += [pbs_1, pbs_2, pbs_3, pbs_1 / (1 + pbs_1 + pbs_2 + pbs_3), pbs_2 / (1 + pbs_1 + pbs_2 + pbs_3),
                             pbs_3 / (1 + pbs_1 + pbs_2 + pbs_3)]
    return np.column_stack(cols)


def pop_basic_stat(fbm, rowInd, colInd, groupIds, ngroups, which, by_locus=False, include_global=False, code256=CODE_012):
    """pop_het_obs (which="Ho", R/pop_het_obs.R:78-92), pop_het_exp ("Hs", R/pop_het_exp.R:85-103) and
    pop_fis(method = "Nei87") ("Fis", R/pop_fis.R:108-133)"""
    _, r, _c = _view(fbm, rowInd, colInd)
    with np.errstate(invalid="ignore", divide="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        pf = grouped_summaries_dip_pseudo_cpp(fbm, rowInd, colInd, groupIds, ngroups, np.full(len(r), 2.0), code256)
        sHo = pf["het_obs"]
        n = pf["n"] / 2
        sp2 = pf["freq_alt"] ** 2 + pf["freq_ref"] ** 2
        Hs = (1 - sp2 - sHo / 2 / n)
        Hs = n / (n - 1) * Hs
        x = {"Ho": sHo, "Hs": Hs, "Fis": 1 - sHo / Hs}[which]
        col = {"Ho": 0, "Hs": 1, "Fis": 8}[which]
        if which == "Fis" and include_global and not by_locus:
            return np.append(np.nanmean(x, axis=0), pop_global_stats(fbm, rowInd, colInd, groupIds, ngroups)[col])
        if include_global:
            x = np.column_stack([x, pop_global_stats(fbm, rowInd, colInd, groupIds, ngroups, by_locus=True)[:, col]])
        return x if by_locus else np.nanmean(x, axis=0)


def _fst_loop(fn, pairs1, m, mats, by_locus, return_num_dem):
    pairs1 = np.ascontiguousarray(np.asarray(pairs1, dtype=np.int32).T)  # (P, 2) rows = (pop1, pop2)
    P = pairs1.shape[0]
    tot = np.zeros(P)
    a = np.zeros((m, P), order="F") if by_locus else np.zeros((0, 0))
    b = np.zeros((m, P), order="F") if return_num_dem else np.zeros((0, 0))
    fn(_p(pairs1, C.c_int32), C.c_int(P), C.c_int(m), *[_d(np.asfortranarray(x)) for x in mats],
       C.c_int(int(by_locus)), C.c_int(int(return_num_dem)), _d(tot), _d(a), _d(b))
    if not return_num_dem:
        return dict(fst_locus=a, fst_tot=tot)
    return dict(Fst_by_locus_num=a, Fst_by_locus_den=b)


def pairwise_fst_hudson_loop(pairwise_combn, n, freq_alt, freq_ref, by_locus=False, return_num_dem=False):
    """src/pairwise_fst_hudson_loop.cpp:5-63; pairwise_combn is 2 x P, 1-based"""
    return _fst_loop(lib().orc_pairwise_fst_hudson_loop, pairwise_combn, n.shape[0],
                     [n, freq_alt, freq_ref], by_locus, return_num_dem)


def pairwise_fst_wc84_loop(pairwise_combn, n, freq_alt, het_obs, by_locus=False, return_num_dem=False):
    """src/pairwise_fst_wc84_loop.cpp:5-121"""
    return _fst_loop(lib().orc_pairwise_fst_wc84_loop, pairwise_combn, n.shape[0],
                     [n, freq_alt, het_obs], by_locus, return_num_dem)


def pairwise_fst_nei87_loop(pairwise_combn, n, het_obs, freq_alt, freq_ref, by_locus=False,
                            return_num_dem=False):
    """src/pairwise_fst_nei87_loop.cpp:5-115"""
    return _fst_loop(lib().orc_pairwise_fst_nei87_loop, pairwise_combn, n.shape[0],
                     [n, het_obs, freq_alt, freq_ref], by_locus, return_num_dem)


def fbm256_prod_and_rowSumsSq(fbm, ind_row, ind_col, center, scale, V, code256=CODE_012):
    """src/fbm_prod_and_rowSumSq.cpp:10-47 -> (XV (n,K), rowSumsSq (n,))"""
    fbm, r, c = _view(fbm, ind_row, ind_col)
    V = np.asfortranarray(V, dtype=float)
    assert V.shape[0] == len(c)
    XV = np.zeros((len(r), V.shape[1]), order="F")
    rss = np.zeros(len(r))
    lib().orc_fbm256_prod_and_rowSumsSq(*_args(fbm, r, c), _d(np.ascontiguousarray(code256)),
                                        _d(np.ascontiguousarray(center, dtype=float)),
                                        _d(np.ascontiguousarray(scale, dtype=float)), _d(V),
                                        C.c_int(V.shape[1]), _d(XV), _d(rss))
    return XV, rss


# --------------------------------------------------------------------------
# R-level drivers (block loop + epilogue)

def _blocks(m, block_size):
    lo, up = cut_by_size(m, block_size)
    return [(int(a) - 1, int(b)) for a, b in zip(lo, up)]  # python slices


def snp_ibs(fbm, ind_row=None, ind_col=None, type="proportion", block_size=None):
    """R/snp_ibs.R:42-104"""
    fbm, r, c = _view(fbm, ind_row, ind_col)
    n, m = len(r), len(c)
    block_size = block_size or block_size_default(fbm.shape[0])
    IBS = np.zeros((n, n), order="F")
    valid = np.zeros((n, n), order="F")
    for a, b in _blocks(m, block_size):
        increment_ibs_counts(IBS, valid, fbm, r, c[a:b])
    if type == "raw_counts":
        return dict(ibs=IBS, valid_n=valid)
    with np.errstate(invalid="ignore", divide="ignore"):
        prop = IBS / valid
    return prop if type == "proportion" else prop * m


def snp_king(fbm, ind_row=None, ind_col=None, block_size=None):
    """R/snp_king.R:32-103"""
    fbm, r, c = _view(fbm, ind_row, ind_col)
    n, m = len(r), len(c)
    block_size = block_size or block_size_default(fbm.shape[0]) * 4
    K = np.zeros((n, n), order="F")
    Ni = np.zeros((n, n), order="F")
    for a, b in _blocks(m, block_size):
        increment_king_numerator(K, Ni, fbm, r, c[a:b])
    return king_epilogue(K, Ni)


def king_epilogue(K, Ni):
    """R/snp_king.R:79-101"""
    Nj = Ni.T
    mn = np.minimum(Ni, Nj)
    with np.errstate(invalid="ignore", divide="ignore"):
        return K / (2 * mn) + 0.5 - 0.25 * (Ni + Nj) / mn


def snp_allele_sharing(fbm, ind_row=None, ind_col=None, block_size=None, emulate_as_pad_quirk=False):
    """R/snp_allele_sharing.R:33-82.  emulate_as_pad_quirk=True reproduces what the
    reference BINARY does at unequal block sizes (quirk Q1); the default is the
    mathematically intended value that the reference's own test asserts."""
    fbm, r, c = _view(fbm, ind_row, ind_col)
    n, m = len(r), len(c)
    block_size = block_size or block_size_default(fbm.shape[0])
    num = np.zeros((n, n), order="F")
    den = np.zeros((n, n), order="F")
    blocks = _blocks(m, block_size)
    widest = max(b - a for a, b in blocks)
    for a, b in blocks:
        increment_as_counts(num, den, fbm, r, c[a:b],
                            pad_quirk=emulate_as_pad_quirk and (b - a) < widest)
    return as_epilogue(num, den)


def as_epilogue(num, den):
    """R/snp_allele_sharing.R:77-81"""
    with np.errstate(invalid="ignore", divide="ignore"):
        res = 0.5 * (1 + num / den)
    res[den == 0] = np.nan
    return res


def pairwise_grm(allele_sharing_mat):
    """R/pairwise_grm.R:42-50"""
    M = np.array(allele_sharing_mat, dtype=float, copy=True)
    off = M.copy()
    np.fill_diagonal(off, np.nan)
    mb = np.nanmean(off)
    return (M - mb) / (1 - mb) * 2


def loci_alt_freq(fbm, ind_row=None, ind_col=None, ploidy=None, as_counts=False, block_size=None,
                  code256=CODE_012):
    """R/loci_alt_freq.R:328-379 (diploid / pseudohaploid path, >1 individual)"""
    fbm, r, c = _view(fbm, ind_row, ind_col)
    ploidy = np.full(len(r), 2.0) if ploidy is None else np.asarray(ploidy, dtype=float)
    block_size = block_size or len(c)
    parts = [alt_freq_dip_pseudo_cpp(fbm, r, c[a:b], ploidy, as_counts, code256)
             for a, b in _chunks(len(c), block_size)]
    freq = np.vstack(parts)
    return freq if as_counts else freq[:, 0]


def _chunks(m, block_size):
    """bigstatsr::big_apply splits `ind` with CutBySize semantics (recalled)."""
    return _blocks(m, block_size)


def loci_missingness(fbm, ind_row=None, ind_col=None, as_counts=False, code256=CODE_012):
    """R/loci_missingness.R:97-134; the count itself is bigstatsr::big_counts
    (third-party, recalled: per column, number of entries decoding to NA)."""
    fbm, r, c = _view(fbm, ind_row, ind_col)
    gid = np.zeros(len(r), dtype=np.int32)
    n_na = grouped_missingness_cpp(fbm, r, c, gid, 1, code256)[:, 0]
    return n_na if as_counts else n_na / len(r)


def combn2(G: int) -> np.ndarray:
    """utils::combn(G, 2): 2 x P, 1-based, column order (1,2),(1,3),...,(G-1,G)
    (R/pairwise_pop_fst.R:119)."""
    cols = [(a, b) for a in range(1, G + 1) for b in range(a + 1, G + 1)]
    return np.array(cols, dtype=np.int32).T.reshape(2, -1)


def pairwise_pop_fst(fbm, ind_row, ind_col, groupIds, ngroups, ploidy=None, method="Hudson",
                     by_locus=False, return_num_dem=False, code256=CODE_012):
    """R/pairwise_pop_fst.R:116-161"""
    fbm, r, c = _view(fbm, ind_row, ind_col)
    ploidy = np.full(len(r), 2.0) if ploidy is None else np.asarray(ploidy, dtype=float)
    if return_num_dem:
        by_locus = True
    pf = grouped_summaries_dip_pseudo_cpp(fbm, r, c, groupIds, ngroups, ploidy, code256)
    pairs = combn2(ngroups)
    if method == "Hudson":
        return pairwise_fst_hudson_loop(pairs, pf["n"], pf["freq_alt"], pf["freq_ref"], by_locus, return_num_dem)
    if method == "Nei87":
        return pairwise_fst_nei87_loop(pairs, pf["n"], pf["het_obs"], pf["freq_alt"], pf["freq_ref"], by_locus,
                                       return_num_dem)
    if method == "WC84":
        return pairwise_fst_wc84_loop(pairs, pf["n"], pf["freq_alt"], pf["het_obs"], by_locus, return_num_dem)
    raise ValueError(method)


def pca_gram(fbm, ind_row=None, ind_col=None, code256=CODE_IMPUTE_PRED):
    """center / scale (bigsnpr::snp_scaleBinom, recalled) and K = Z Z'
    (bigstatsr::big_SVD, recalled); call site R/gt_pca_partialSVD.R:82-89."""
    fbm, r, c = _view(fbm, ind_row, ind_col)
    n, m = len(r), len(c)
    center = np.zeros(m)
    scale = np.zeros(m)
    K = np.zeros((n, n), order="F")
    rc = lib().orc_pca_center_scale_gram(*_args(fbm, r, c), _d(np.ascontiguousarray(code256)),
                                         _d(center), _d(scale), _d(K))
    if rc != 0:
        raise ValueError("missing values or zero scale: big_SVD would stop")
    return center, scale, K


def square_frobenius(fbm, ind_row, ind_col, center, scale, code256=CODE_IMPUTE_PRED):
    """R/square_frobenius.R:19-35"""
    fbm, r, c = _view(fbm, ind_row, ind_col)
    return float(lib().orc_square_frobenius(*_args(fbm, r, c), _d(np.ascontiguousarray(code256)),
                                            _d(np.ascontiguousarray(center, dtype=float)),
                                            _d(np.ascontiguousarray(scale, dtype=float))))


def gt_pca_partialSVD(fbm, ind_row=None, ind_col=None, k=10, total_var=True, code256=CODE_IMPUTE_PRED):
    """R/gt_pca_partialSVD.R:67-108 around bigstatsr::big_SVD (recalled):
    eigen(K) -> d = sqrt(lambda_1..k), u = eigenvectors, v = Z'u/d."""
    fbm, r, c = _view(fbm, ind_row, ind_col)
    center, scale, K = pca_gram(fbm, r, c, code256)
    w, U = np.linalg.eigh(K)
    order = np.argsort(w)[::-1][:k]
    d = np.sqrt(w[order])
    u = np.asfortranarray(U[:, order])
    v = np.zeros((len(c), k), order="F")
    lib().orc_pca_loadings(*_args(fbm, r, c), _d(np.ascontiguousarray(code256)), _d(center), _d(scale),
                           _d(u), _d(d), C.c_int(k), _d(v))
    out = dict(d=d, u=u, v=v, center=center, scale=scale)
    if total_var:
        out["square_frobenius"] = square_frobenius(fbm, r, c, center, scale, code256)
    return out


# --------------------------------------------------------------------------
# "as the reference does it" CPU baseline: dense FP64 one-hot blocks and the
# same 6 / 4 / 2 products per block through the threaded BLAS numpy links
# (src/snp_ibs.cpp:67-72, src/snp_king.cpp:70-72, src/snp_as.cpp:64-65).

def _onehots(fbm, r, c):
    sub = fbm[np.ix_(r - 1, c - 1)]
    return [(sub == v).astype(np.float64) for v in (0, 1, 2)]


def blas_increment_ibs(K, K2, fbm, r, c):
    g0, g1, g2 = _onehots(fbm, r, c)
    K += 2 * (g2 @ g2.T + g1 @ g1.T + g0 @ g0.T) + g1 @ (g0 + g2).T + (g0 + g2) @ g1.T
    v = g0 + g1 + g2
    K2 += 2 * v @ v.T


def blas_increment_king(K, Ni, fbm, r, c):
    g0, g1, g2 = _onehots(fbm, r, c)
    v = g0 + g1 + g2
    K += g1 @ g1.T - 2 * (g0 @ g2.T + g2 @ g0.T)
    Ni += g1 @ v.T


def blas_increment_as(K, K2, fbm, r, c):
    sub = fbm[np.ix_(r - 1, c - 1)]
    na = (sub < 3).astype(np.float64)
    dos = (sub.astype(np.float64) - 1.0) * na
    K += dos @ dos.T
    K2 += na @ na.T


# --------------------------------------------------------------------------
# relatedness filter and PCA projection (host-side consumers of the N x N / loadings outputs)

def r_mean(x) -> float:
    """R's mean() of a double vector after na.rm (src/main/summary.c, recalled): long double sum / n, then one
    refinement pass sum(x - mean) / n, rounded to double at the end."""
    x = np.asarray(x, dtype=np.float64)
    x = x[~np.isnan(x)]
    n = len(x)
    if n == 0:
        return float("nan")
    xl = x.astype(np.longdouble)
    s = np.longdouble(0)
    for v in xl:          # sequential long double accumulation, as the C loop
        s += v
    s /= n
    t = np.longdouble(0)
    for v in xl:
        t += v - s
    return float(s + t / n)


def filter_high_relatedness(matrix, kings_threshold):
    """R/filter_high_relatedness.R:26-145, statement by statement (names are 1..n as strings when the matrix has
    none).  -> (passed_filter ids in the new order, to_remove ids, logical keep vector in the original order).
    Quadratic work per comparison, as the reference: small matrices only."""
    M = np.array(matrix, dtype=np.float64, copy=True)
    var_num = M.shape[0]
    var_names = np.arange(1, var_num + 1)
    if var_num == 1:                                                       # :46-52
        return var_names.copy(), np.zeros(0, dtype=int), np.ones(1, dtype=bool)
    M = np.abs(M)                                                          # :55
    tmp = M.copy()
    np.fill_diagonal(tmp, np.nan)                                          # :69
    col_means = np.array([r_mean(tmp[:, j]) for j in range(var_num)])      # :72-73 apply(tmp, 2, mean, na.rm)
    # order(decreasing = TRUE): stable, NaN last
    key = np.where(np.isnan(col_means), -np.inf, col_means)
    order = np.argsort(-key, kind="stable")
    M = M[np.ix_(order, order)]                                            # :76
    new_order = order                                                      # :79 (0-based here)
    col_to_delete = np.zeros(var_num, dtype=bool)                          # :83
    M2 = M.copy()
    np.fill_diagonal(M2, np.nan)                                           # :85
    for i in range(var_num - 1):                                           # :90
        with np.errstate(invalid="ignore"):
            if not np.any(M2[~np.isnan(M2)] > kings_threshold):            # :91-96
                break
        if col_to_delete[i]:
            continue
        for j in range(i + 1, var_num):                                    # :100
            if not col_to_delete[i] and not col_to_delete[j]:
                if np.isnan(M[i, j]):
                    raise ValueError("missing value where TRUE/FALSE needed")  # R: if (NA)
                if M[i, j] > kings_threshold:                              # :102
                    mn1 = r_mean(M2[i, :])                                 # :103
                    mn2 = r_mean(np.delete(M2, j, axis=0).ravel(order="F"))  # :104 mean(matrix2[-j, ]): ALL other rows
                    if mn1 > mn2:                                          # :119
                        col_to_delete[i] = True
                        M2[i, :] = np.nan
                        M2[:, i] = np.nan
                    else:
                        col_to_delete[j] = True
                        M2[j, :] = np.nan
                        M2[:, j] = np.nan
    passed = var_names[new_order][~col_to_delete]                          # :138
    keep = np.isin(var_names, passed)
    return passed, var_names[~keep], keep


def predict_gt_pca(pca, fbm=None, ind_row=None, ind_col=None, project_method="none", lsq_pcs=(1, 2),
                   code256=CODE_IMPUTE_PRED):
    """R/predict_gt_pca.R:73-236 (numeric part; the loci-name matching is done by the caller through ind_col).
    pca = dict(d, u, v, center, scale).  "OADP" needs bigsnpr's OADP_proj, which is not in the reference checkout."""
    if fbm is None:
        return pca["u"] * pca["d"]                                         # :103 sweep(u, 2, d, "*")
    fbm, r, c = _view(fbm, ind_row, ind_col)
    if project_method in ("none", "simple"):
        # "none": bigstatsr::big_prodMat on the imputed code (third-party, recalled: (X - center) / scale %*% V);
        # "simple": fbm256_prod_and_rowSumsSq, missing -> 0 (:157-170, src/fbm_prod_and_rowSumSq.cpp:30-44)
        XV, _ = fbm256_prod_and_rowSumsSq(fbm, r, c, pca["center"], pca["scale"], pca["v"], code256)
        return XV
    if project_method == "least_squares":                                  # :187-232
        lsq = np.asarray(lsq_pcs)
        if len(lsq) == 0 or np.any(lsq < 1) or np.any(lsq > pca["v"].shape[1]) or np.any(lsq != lsq.astype(int)):
            raise ValueError("lsq_pcs should be a vector of valid component indices")
        if len(set(lsq.tolist())) != len(lsq):
            raise ValueError("lsq_pcs should not contain duplicate values")
        g = np.asarray(code256)[fbm[np.ix_(r - 1, c - 1)]]
        out = np.zeros((len(r), len(lsq)))
        for i in range(len(r)):
            gs = (g[i] - pca["center"]) / pca["scale"]
            ok = ~np.isnan(gs)
            vs = pca["v"][np.ix_(np.where(ok)[0], lsq - 1)]
            out[i] = np.linalg.solve(vs.T @ vs, vs.T @ gs[ok])             # solve(crossprod(v_sub), crossprod(v_sub, g))
        return out
    raise ValueError("project_method 'OADP' calls bigsnpr's OADP_proj (third-party, not in the reference checkout)")


# --------------------------------------------------------------------------
# synthetic panel (bit-identical to the HIP generator)

def synth_fbm(seed: int, n: int, m: int, j0: int = 0, npop: int = 51, miss: float = 0.02,
              imputed_bytes: bool = False) -> np.ndarray:
    out = np.zeros((n, m), dtype=np.uint8, order="F")
    thr = int(round(miss * 2 ** 32))
    thr = min(thr, 2 ** 32 - 1)
    lib().orc_synth_fbm(C.c_uint64(seed), C.c_int64(n), C.c_int64(m), C.c_int64(j0), C.c_int(npop),
                        C.c_uint32(thr), C.c_int(int(imputed_bytes)), _p(out, C.c_uint8))
    return out


def synth_rows(seed: int, rows0, m: int, j0: int = 0, npop: int = 51, miss: float = 0.02,
               imputed_bytes: bool = False) -> np.ndarray:
    """rows `rows0` (0-based) of the panel synth_fbm(seed, n, m, ...) would give, without generating the rest"""
    rows0 = np.ascontiguousarray(rows0, dtype=np.int64)
    out = np.zeros((len(rows0), m), dtype=np.uint8, order="F")
    thr = min(int(round(miss * 2 ** 32)), 2 ** 32 - 1)
    lib().orc_synth_rows(C.c_uint64(seed), _p(rows0, C.c_int64), C.c_int64(len(rows0)), C.c_int64(m), C.c_int64(j0),
                         C.c_int(npop), C.c_uint32(thr), C.c_int(int(imputed_bytes)), _p(out, C.c_uint8))
    return out
