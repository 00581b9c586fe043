/*
 * tpg_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the arithmetic of tidypopgen's per-SNP genotype
 * matrix hot path, following the reference statement by statement.  Every
 * function cites the reference file:line it restates (paths relative to the
 * reference checkout).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker.
 *
 * Pinning: this restatement is checked (tests/test_oracle_golden.py) against
 * the golden vectors the reference's own tests hold for this path:
 *   inst/extdata/related/families.bed + test_plinkIBS.mibs  (IBS, 6 dp)
 *   inst/extdata/related/test_king.kin0                      (KING)
 *   tests/testthat/testdata/fst_scikit-allel/ (6 files)      (Hudson, WC84)
 *   literal matrices of tests/testthat/test_{snp_ibs,snp_king,loci_freq,
 *   loci_missingness,pairwise_allele_sharing,pairwise_grm,pairwise_pop_fst}.R
 * The reference itself (R + Rcpp + bigstatsr) cannot be built here (no R,
 * no bigstatsr/Rcpp/Armadillo headers), so there is no oracle/_ref.
 *
 * Conventions (identical to the reference's .Call boundary):
 *   - the genotype store is a bigstatsr FBM: uint8, column-major, element
 *     (i,j) at fbm[i + j*nrow]  (bigstatsr BMAcc, recalled);
 *   - rowInd / colInd are 1-based (src/snp_ibs.cpp:35 passes `1` to the
 *     accessor, which subtracts it);
 *   - code256 is double[256]; NA is any NaN (`x > -1` is false for it,
 *     src/alt_freq_dip_pseudo_cpp.cpp:35);
 *   - all matrices are column-major doubles, as R stores them.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off so that no FMA contraction
 * changes the reference's operation order).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FBM(i, j) fbm[(size_t)(rowInd[(i)] - 1) + (size_t)(colInd[(j)] - 1) * (size_t)nrow]

static inline int is_na(double x) { return isnan(x); } /* NumericVector::is_na == R_isnancpp */

/* ------------------------------------------------------------------------- */
/* bigparallelr::split_len (third-party, recalled; used by CutBySize,
 * R/local_reimplementations.R:13-15).  Fills lower/upper (1-based, inclusive)
 * for nb blocks.  R's round() is round-half-even, as rint() under the default
 * rounding mode. */
void orc_split_len(int total_len, int nb, int* lower, int* upper) {
  double step = (double)total_len / (double)nb;
  for (int b = 0; b < nb; b++) {
    upper[b] = (int)rint((double)(b + 1) * step);
    lower[b] = (b == 0) ? 1 : upper[b - 1] + 1;
  }
}

/* ------------------------------------------------------------------------- */
/* src/snp_ibs.cpp:22-74  increment_ibs_counts.
 * The reference fills three one-hot n x B double matrices from the RAW bytes
 * (:45-55: value==0 / ==1 / ==2, anything else is missing) and forms
 *   K  += 2*(G2 G2' + G1 G1' + G0 G0') + G1 (G0+G2)' + (G0+G2) G1'   (:67-68)
 *   K2 += 2 * V V',  V = G0+G1+G2                                    (:71-72)
 * All terms are integer valued, so the sum over loci is exact in double in any
 * order; we accumulate per pair in int64 and add once. */
void orc_increment_ibs_counts(const uint8_t* fbm, int64_t nrow, const int32_t* rowInd, int n,
                              const int32_t* colInd, int m, double* K, double* K2) {
  uint8_t* g = (uint8_t*)malloc((size_t)n * (size_t)m);
  for (int j = 0; j < m; j++)
    for (int i = 0; i < n; i++) g[(size_t)i * m + j] = FBM(i, j);
#pragma omp parallel for schedule(dynamic, 1)
  for (int a = 0; a < n; a++) {
    const uint8_t* ga = g + (size_t)a * m;
    for (int b = 0; b < n; b++) {
      const uint8_t* gb = g + (size_t)b * m;
      int64_t k = 0, k2 = 0;
      for (int j = 0; j < m; j++) {
        int va = ga[j], vb = gb[j];
        if (va > 2 || vb > 2) continue;
        k2 += 2;
        if (va == vb) k += 2;                                   /* 2 * sum_a Ga Ga' */
        else if (va == 1 || vb == 1) k += 1;                    /* G1 (G0+G2)' + (G0+G2) G1' */
      }
      K[(size_t)a + (size_t)b * n] += (double)k;
      K2[(size_t)a + (size_t)b * n] += (double)k2;
    }
  }
  free(g);
}

/* ------------------------------------------------------------------------- */
/* src/snp_king.cpp:21-74  increment_king_numerator.
 *   K      += G1 G1' - 2 (G0 G2' + G2 G0')     (:70)
 *   N_Aa_i += G1 V'                            (:72)   (row = i is the het one) */
void orc_increment_king_numerator(const uint8_t* fbm, int64_t nrow, const int32_t* rowInd, int n,
                                  const int32_t* colInd, int m, double* K, double* N_Aa_i) {
  uint8_t* g = (uint8_t*)malloc((size_t)n * (size_t)m);
  for (int j = 0; j < m; j++)
    for (int i = 0; i < n; i++) g[(size_t)i * m + j] = FBM(i, j);
#pragma omp parallel for schedule(dynamic, 1)
  for (int a = 0; a < n; a++) {
    const uint8_t* ga = g + (size_t)a * m;
    for (int b = 0; b < n; b++) {
      const uint8_t* gb = g + (size_t)b * m;
      int64_t k = 0, naa = 0;
      for (int j = 0; j < m; j++) {
        int va = ga[j], vb = gb[j];
        if (va == 1 && vb <= 2) naa += 1;
        if (va == 1 && vb == 1) k += 1;
        else if ((va == 0 && vb == 2) || (va == 2 && vb == 0)) k -= 2;
      }
      K[(size_t)a + (size_t)b * n] += (double)k;
      N_Aa_i[(size_t)a + (size_t)b * n] += (double)naa;
    }
  }
  free(g);
}

/* ------------------------------------------------------------------------- */
/* src/snp_as.cpp:22-67  increment_as_counts.
 *   dos = value-1 if value<3 else 0 ; na = 1 if value<3 else 0     (:44-53)
 *   K += dos dos' ; K2 += na na'                                    (:64-65)
 * pad_quirk != 0 reproduces reference quirk Q1: when the R
 * driver's scratch matrices are one column wider than this block
 * (R/snp_allele_sharing.R:55-56), :57-63 fill the spare column with dos=1,
 * na=0, which adds +1 to EVERY element of K and nothing to K2. */
void orc_increment_as_counts(const uint8_t* fbm, int64_t nrow, const int32_t* rowInd, int n,
                             const int32_t* colInd, int m, int pad_quirk, double* K, double* K2) {
  int8_t* d = (int8_t*)malloc((size_t)n * (size_t)m);
  uint8_t* v = (uint8_t*)malloc((size_t)n * (size_t)m);
  for (int j = 0; j < m; j++)
    for (int i = 0; i < n; i++) {
      int value = FBM(i, j);
      if (value < 3) { d[(size_t)i * m + j] = (int8_t)(value - 1); v[(size_t)i * m + j] = 1; }
      else           { d[(size_t)i * m + j] = 0;                   v[(size_t)i * m + j] = 0; }
    }
#pragma omp parallel for schedule(dynamic, 1)
  for (int a = 0; a < n; a++)
    for (int b = 0; b < n; b++) {
      int64_t k = 0, k2 = 0;
      const int8_t *da = d + (size_t)a * m, *db = d + (size_t)b * m;
      const uint8_t *va = v + (size_t)a * m, *vb = v + (size_t)b * m;
      for (int j = 0; j < m; j++) { k += da[j] * db[j]; k2 += va[j] & vb[j]; }
      if (pad_quirk) k += 1;
      K[(size_t)a + (size_t)b * n] += (double)k;
      K2[(size_t)a + (size_t)b * n] += (double)k2;
    }
  free(d); free(v);
}

/* ------------------------------------------------------------------------- */
/* src/alt_freq_dip_pseudo_cpp.cpp:8-58.  out is m x 2 column-major:
 * col 0 = n_alt (as_counts) or freq, col 1 = n_valid.  Same loop order. */
void orc_alt_freq_dip_pseudo(const uint8_t* fbm, int64_t nrow, const int32_t* rowInd, int n,
                             const int32_t* colInd, int m, const double* code256,
                             const double* ploidy, int as_counts, double* out) {
  double* mult = (double*)malloc(sizeof(double) * (size_t)n);
  for (int i = 0; i < n; i++) mult[i] = 1 / (3 - ploidy[i]);                   /* :25-27 */
  for (int j = 0; j < m; j++) { out[j] = 0; out[(size_t)m + j] = 0; }
#pragma omp parallel for schedule(static)
  for (int j = 0; j < m; j++)                                                   /* :33-41 */
    for (int i = 0; i < n; i++) {
      double x = code256[FBM(i, j)];
      if (x > -1) { out[j] += x * mult[i]; out[(size_t)m + j] += ploidy[i]; }
    }
  if (!as_counts)                                                               /* :48-54 */
    for (int j = 0; j < m; j++) {
      if (out[(size_t)m + j] > 0) out[j] = out[j] / out[(size_t)m + j];
      else out[j] = NAN; /* NA_REAL */
    }
  free(mult);
}

/* src/grouped_alt_freq_dip_pseudo_cpp.cpp:8-58.  out is m x 2G column-major
 * (G alt/freq columns then G valid columns); no NA guard (0/0 -> NaN). */
void orc_grouped_alt_freq_dip_pseudo(const uint8_t* fbm, int64_t nrow, const int32_t* rowInd, int n,
                                     const int32_t* colInd, int m, const double* code256,
                                     const int32_t* groupIds, int ngroups, const double* ploidy,
                                     int as_counts, double* out) {
  double* mult = (double*)malloc(sizeof(double) * (size_t)n);
  for (int i = 0; i < n; i++) mult[i] = 1 / (3 - ploidy[i]);
  memset(out, 0, sizeof(double) * (size_t)m * 2 * (size_t)ngroups);
#pragma omp parallel for schedule(static)
  for (int j = 0; j < m; j++)
    for (int i = 0; i < n; i++) {
      double x = code256[FBM(i, j)];
      if (x > -1) {
        out[(size_t)j + (size_t)groupIds[i] * m] += x * mult[i];
        out[(size_t)j + (size_t)(ngroups + groupIds[i]) * m] += ploidy[i];
      }
    }
  if (!as_counts)
    for (int j = 0; j < m; j++)
      for (int g = 0; g < ngroups; g++)
        out[(size_t)j + (size_t)g * m] = out[(size_t)j + (size_t)g * m] / out[(size_t)j + (size_t)(ngroups + g) * m];
  free(mult);
}

/* src/grouped_missingness_cpp.cpp:8-33.  out is m x G column-major. */
void orc_grouped_missingness(const uint8_t* fbm, int64_t nrow, const int32_t* rowInd, int n,
                             const int32_t* colInd, int m, const double* code256,
                             const int32_t* groupIds, int ngroups, double* out) {
  memset(out, 0, sizeof(double) * (size_t)m * (size_t)ngroups);
#pragma omp parallel for schedule(static)
  for (int j = 0; j < m; j++)
    for (int i = 0; i < n; i++) {
      double x = code256[FBM(i, j)];
      if (!(x > -1)) out[(size_t)j + (size_t)groupIds[i] * m] += 1;
    }
}

/* src/grouped_summaries_dip_pseudo_cpp.cpp:11-63.  Four m x G column-major
 * outputs: freq_alt, freq_ref, n (valid alleles), het_obs. */
void orc_grouped_summaries_dip_pseudo(const uint8_t* fbm, int64_t nrow, const int32_t* rowInd, int n,
                                      const int32_t* colInd, int m, const double* code256,
                                      const int32_t* groupIds, int ngroups, const double* ploidy,
                                      double* freq, double* ref_freq, double* valid_alleles,
                                      double* heterozygotes) {
  size_t sz = sizeof(double) * (size_t)m * (size_t)ngroups;
  memset(freq, 0, sz); memset(ref_freq, 0, sz); memset(valid_alleles, 0, sz); memset(heterozygotes, 0, sz);
  double* mult = (double*)malloc(sizeof(double) * (size_t)n);
  for (int i = 0; i < n; i++) mult[i] = 1 / (3 - ploidy[i]);
#pragma omp parallel for schedule(static)
  for (int j = 0; j < m; j++) {
    for (int i = 0; i < n; i++) {
      double x = code256[FBM(i, j)];
      if (x > -1) {
        size_t o = (size_t)j + (size_t)groupIds[i] * m;
        freq[o] += x * mult[i];
        valid_alleles[o] += ploidy[i];
        if (x == 1) heterozygotes[o] += 2;
      }
    }
    for (int g = 0; g < ngroups; g++) {
      size_t o = (size_t)j + (size_t)g * m;
      freq[o] = freq[o] / valid_alleles[o];
      ref_freq[o] = 1 - freq[o];
      heterozygotes[o] = heterozygotes[o] / valid_alleles[o];
    }
  }
  free(mult);
}

/* src/gt_ind_hetero.cpp:11-42.  out is 2 x n column-major ints: row 0 het, row 1 na */
void orc_gt_ind_hetero(const uint8_t* fbm, int64_t nrow, const int32_t* rowInd, int n, const int32_t* colInd,
                       int m, const double* code256, int32_t* out) {
  memset(out, 0, sizeof(int32_t) * 2 * (size_t)n);
  for (int j = 0; j < m; j++)
    for (int i = 0; i < n; i++) {
      double x = code256[FBM(i, j)];
      if (x > -1) { if (x == 1) out[2 * (size_t)i] += 1; }
      else out[2 * (size_t)i + 1] += 1;
    }
}

/* src/gt_pi_diploid.cpp:7-38 */
void orc_gt_pi_diploid(const uint8_t* fbm, int64_t nrow, const int32_t* rowInd, int n, const int32_t* colInd,
                       int m, const double* code256, double* pi) {
  for (int j = 0; j < m; j++) {
    double cnt = 0, valid = 0;
    for (int i = 0; i < n; i++) {
      double x = code256[FBM(i, j)];
      if (x > -1) { cnt += x; valid += 2; }
    }
    pi[j] = (valid > 0) ? (cnt * (valid - cnt) / (valid * (valid - 1) / 2)) : NAN;
  }
}

/* src/gt_grouped_pi_diploid.cpp:7-42.  pi and n are m x G column-major */
void orc_gt_grouped_pi_diploid(const uint8_t* fbm, int64_t nrow, const int32_t* rowInd, int n,
                               const int32_t* colInd, int m, const double* code256, const int32_t* groupIds,
                               int ngroups, double* pi, double* valid) {
  memset(pi, 0, sizeof(double) * (size_t)m * (size_t)ngroups);
  memset(valid, 0, sizeof(double) * (size_t)m * (size_t)ngroups);
  for (int j = 0; j < m; j++) {
    for (int i = 0; i < n; i++) {
      double x = code256[FBM(i, j)];
      if (x > -1) { pi[(size_t)j + (size_t)groupIds[i] * m] += x; valid[(size_t)j + (size_t)groupIds[i] * m] += 2; }
    }
    for (int g = 0; g < ngroups; g++) {
      size_t o = (size_t)j + (size_t)g * m;
      pi[o] = (pi[o] * (valid[o] - pi[o]) / (valid[o] * (valid[o] - 1) / 2));
    }
  }
}

/* ------------------------------------------------------------------------- */
/* Fst pair loops.  pairs1 is 2 x P column-major, 1-based doubles in R; here
 * int32 1-based.  Inputs are m x G column-major.  fst_tot has P entries.
 * If by_locus: out_a is m x P (ratio, or numerator when return_num_dem);
 * if return_num_dem: out_b is m x P (denominator). */

/* src/pairwise_fst_hudson_loop.cpp:5-63 */
void orc_pairwise_fst_hudson_loop(const int32_t* pairs1, int P, int m, const double* n,
                                  const double* freq_alt, const double* freq_ref, int by_locus,
                                  int return_num_dem, double* fst_tot, double* out_a, double* out_b) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int c = 0; c < P; c++) {
    const double *p1 = freq_alt + (size_t)(pairs1[2 * c] - 1) * m, *p2 = freq_alt + (size_t)(pairs1[2 * c + 1] - 1) * m;
    const double *q1 = freq_ref + (size_t)(pairs1[2 * c] - 1) * m, *q2 = freq_ref + (size_t)(pairs1[2 * c + 1] - 1) * m;
    const double *n1 = n + (size_t)(pairs1[2 * c] - 1) * m, *n2 = n + (size_t)(pairs1[2 * c + 1] - 1) * m;
    double mean_num = 0.0, mean_den = 0.0;
    for (int i = 0; i < m; i++) {
      double d = p1[i] - p2[i];
      double num = pow(d, 2) - (p1[i] * q1[i]) / (n1[i] - 1) - (p2[i] * q2[i]) / (n2[i] - 1);   /* :27-29 */
      double den = p1[i] * q2[i] + p2[i] * q1[i];                                               /* :31-32 */
      if (by_locus) {
        if (!return_num_dem) out_a[(size_t)i + (size_t)c * m] = num / den;
        else { out_a[(size_t)i + (size_t)c * m] = num; out_b[(size_t)i + (size_t)c * m] = den; }
      }
      if (!is_na(num) && !is_na(den)) { mean_num += num; mean_den += den; }                      /* :45-51 */
    }
    fst_tot[c] = mean_num / mean_den;
  }
}

/* src/pairwise_fst_wc84_loop.cpp:5-121 (r = 2 populations per pair) */
void orc_pairwise_fst_wc84_loop(const int32_t* pairs1, int P, int m, const double* n,
                                const double* freq_alt, const double* het_obs, int by_locus,
                                int return_num_dem, double* fst_tot, double* out_a, double* out_b) {
  const int r = 2;
#pragma omp parallel for schedule(dynamic, 4)
  for (int c = 0; c < P; c++) {
    const double* an[2]; const double* p[2]; const double* h[2];
    for (int j = 0; j < r; j++) {
      size_t o = (size_t)(pairs1[2 * c + j] - 1) * m;
      an[j] = n + o; p[j] = freq_alt + o; h[j] = het_obs + o;
    }
    double mean_num = 0.0, mean_den = 0.0;
    for (int i = 0; i < m; i++) {
      double n_ind[2];
      for (int j = 0; j < r; j++) n_ind[j] = an[j][i] / 2.0;                      /* :41-44 */
      double sum_n = 0.0, sum_sq = 0.0;
      for (int j = 0; j < r; j++) { sum_n += n_ind[j]; sum_sq += pow(n_ind[j], 2); }
      double n_total = sum_n, n_bar = sum_n / r;
      double n_c = (sum_n - sum_sq / sum_n) / (r - 1);                            /* :58 */
      double sum_pn = 0.0, sum_sq_diff = 0.0, sum_h = 0.0;
      for (int j = 0; j < r; j++) { sum_pn += p[j][i] * n_ind[j]; sum_h += h[j][i] * n_ind[j]; }
      double p_bar = sum_pn / n_total, h_bar = sum_h / n_total;
      for (int j = 0; j < r; j++) sum_sq_diff += pow(p[j][i] - p_bar, 2) * n_ind[j];
      double s2 = sum_sq_diff / (n_bar * (r - 1));                                /* :76 */
      double a = n_bar / n_c * (s2 - (1.0 / (n_bar - 1.0)) *
                 (p_bar * (1 - p_bar) - ((r - 1.0) / r) * s2 - h_bar / 4.0));     /* :81-83 */
      double b = n_bar / (n_bar - 1.0) *
                 (p_bar * (1 - p_bar) - ((r - 1.0) / r) * s2 -
                  ((2 * n_bar - 1.0) / (4.0 * n_bar)) * h_bar);                   /* :84-86 */
      double cc = h_bar / 2.0;
      double num = a, den = a + b + cc;
      if (by_locus) {
        if (!return_num_dem) out_a[(size_t)i + (size_t)c * m] = num / den;
        else { out_a[(size_t)i + (size_t)c * m] = num; out_b[(size_t)i + (size_t)c * m] = den; }
      }
      if (!is_na(num) && !is_na(den)) { mean_num += num; mean_den += den; }
    }
    fst_tot[c] = mean_num / mean_den;
  }
}

/* src/pairwise_fst_nei87_loop.cpp:5-115 */
void orc_pairwise_fst_nei87_loop(const int32_t* pairs1, int P, int m, const double* n,
                                 const double* het_obs, const double* freq_alt, const double* freq_ref,
                                 int by_locus, int return_num_dem, double* fst_tot, double* out_a,
                                 double* out_b) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int c = 0; c < P; c++) {
    size_t o1 = (size_t)(pairs1[2 * c] - 1) * m, o2 = (size_t)(pairs1[2 * c + 1] - 1) * m;
    double mean_num = 0.0, mean_den = 0.0;
    for (int i = 0; i < m; i++) {
      double n_pair[2] = {n[o1 + i] / 2.0, n[o2 + i] / 2.0};
      double sHo[2] = {het_obs[o1 + i], het_obs[o2 + i]};
      double fA[2] = {freq_alt[o1 + i], freq_alt[o2 + i]};
      double fR[2] = {freq_ref[o1 + i], freq_ref[o2 + i]};
      int valid = 0; double nsum = 0.0, inv_nsum = 0.0, ho_sum = 0.0;
      for (int j = 0; j < 2; j++)
        if (!is_na(n_pair[j])) { valid++; ho_sum += sHo[j]; nsum += 1.0; inv_nsum += 1.0 / n_pair[j]; }
      double np = valid;
      double mn = (inv_nsum > 0) ? nsum / inv_nsum : NAN;
      double mHo = ho_sum / valid;
      double sp2a = pow(fA[0], 2) + pow(fA[1], 2);
      double sp2r = pow(fR[0], 2) + pow(fR[1], 2);
      double sp2 = sp2a + sp2r;
      double msp2 = sp2 / 2.0;
      double fAm = (fA[0] + fA[1]) / 2.0, fRm = (fR[0] + fR[1]) / 2.0;
      double mp2 = pow(fAm, 2) + pow(fRm, 2);
      double mHs = mn / (mn - 1.0) * (1.0 - msp2 - mHo / (2.0 * mn));
      double Ht = 1.0 - mp2 + mHs / (mn * np) - mHo / (2.0 * mn * np);
      double Dst = Ht - mHs;
      double Dstp = (np / (np - 1.0)) * Dst;
      double Htp = mHs + Dstp;
      if (by_locus) {
        if (!return_num_dem) out_a[(size_t)i + (size_t)c * m] = Dstp / Htp;
        else { out_a[(size_t)i + (size_t)c * m] = Dstp; out_b[(size_t)i + (size_t)c * m] = Htp; }
      }
      if (!is_na(Dstp) && !is_na(Htp)) { mean_num += Dstp; mean_den += Htp; }
    }
    fst_tot[c] = mean_num / mean_den;
  }
}

/* ------------------------------------------------------------------------- */
/* src/fbm_prod_and_rowSumSq.cpp:10-47.  V is m x K column-major, XV is n x K
 * column-major, rss has n entries.  Same accumulation order (j ascending). */
void orc_fbm256_prod_and_rowSumsSq(const uint8_t* fbm, int64_t nrow, const int32_t* rowInd, int n,
                                   const int32_t* colInd, int m, const double* code256,
                                   const double* center, const double* scale, const double* V, int K,
                                   double* XV, double* rss) {
  memset(XV, 0, sizeof(double) * (size_t)n * (size_t)K);
  memset(rss, 0, sizeof(double) * (size_t)n);
  for (int j = 0; j < m; j++)
    for (int i = 0; i < n; i++) {
      double x = code256[FBM(i, j)];
      if (x > -1) x = (x - center[j]) / scale[j];
      else x = 0;
      rss[i] += x * x;
      for (int k = 0; k < K; k++) XV[(size_t)i + (size_t)k * n] += x * V[(size_t)j + (size_t)k * m];
    }
}

/* ------------------------------------------------------------------------- */
/* PCA Gram (a10).  Third-party arithmetic (bigstatsr::big_SVD with
 * bigsnpr::snp_scaleBinom, both absent from the reference checkout; recalled):
 *   center_j = sum_i x_ij / n ; p = center/2 ; scale_j = sqrt(2 p (1-p))
 *   K = sum_j z_j z_j' ,  z_ij = (x_ij - center_j)/scale_j
 * Call site: R/gt_pca_partialSVD.R:82-89.  Returns 1 if a missing value or a
 * zero scale is met (big_SVD errors in both cases), else 0.  K is n x n. */
int orc_pca_center_scale_gram(const uint8_t* fbm, int64_t nrow, const int32_t* rowInd, int n,
                              const int32_t* colInd, int m, const double* code256, double* center,
                              double* scale, double* K) {
  memset(K, 0, sizeof(double) * (size_t)n * (size_t)n);
  double* z = (double*)malloc(sizeof(double) * (size_t)n);
  for (int j = 0; j < m; j++) {
    double s = 0;
    for (int i = 0; i < n; i++) {
      double x = code256[FBM(i, j)];
      if (!(x > -1)) { free(z); return 1; }
      s += x;
    }
    center[j] = s / n;
    double p = center[j] / 2;
    scale[j] = sqrt(2 * p * (1 - p));
    if (!(scale[j] > 0)) { free(z); return 1; }
    for (int i = 0; i < n; i++) z[i] = (code256[FBM(i, j)] - center[j]) / scale[j];
    for (int b = 0; b < n; b++) {
      double zb = z[b];
      double* Kb = K + (size_t)b * n;
      for (int a = 0; a < n; a++) Kb[a] += z[a] * zb;
    }
  }
  free(z);
  return 0;
}

/* R/square_frobenius.R:19-35 with third-party bigstatsr::big_colstats
 * (recalled: per column `sum` and `var` = sample variance with n-1):
 *   sum_j ((n-1) var_j + n (sum_j/n - center_j)^2) / scale_j^2 */
double orc_square_frobenius(const uint8_t* fbm, int64_t nrow, const int32_t* rowInd, int n,
                            const int32_t* colInd, int m, const double* code256, const double* center,
                            const double* scale) {
  double tot = 0;
  for (int j = 0; j < m; j++) {
    double s = 0, ss = 0;
    for (int i = 0; i < n; i++) { double x = code256[FBM(i, j)]; s += x; ss += x * x; }
    double var = (ss - s * s / n) / (n - 1);
    double dm = s / n - center[j];
    tot += ((n - 1) * var + n * dm * dm) / (scale[j] * scale[j]);
  }
  return tot;
}

/* v = Z' u / d  (second sweep of big_SVD, recalled).  U is n x k col-major,
 * d has k entries, Vout is m x k col-major. */
void orc_pca_loadings(const uint8_t* fbm, int64_t nrow, const int32_t* rowInd, int n,
                      const int32_t* colInd, int m, const double* code256, const double* center,
                      const double* scale, const double* U, const double* d, int k, double* Vout) {
  for (int j = 0; j < m; j++)
    for (int c = 0; c < k; c++) {
      double s = 0;
      for (int i = 0; i < n; i++) s += (code256[FBM(i, j)] - center[j]) / scale[j] * U[(size_t)i + (size_t)c * n];
      Vout[(size_t)j + (size_t)c * m] = s / d[c];
    }
}

/* ------------------------------------------------------------------------- */
/* Deterministic synthetic panel generator, integer-only so the
 * HIP generator (tidypopgen_amd/csrc/synth.hip) reproduces it bit for bit.
 * See tidypopgen_amd/csrc/synth_common.h for the shared definition. */
#include "../tidypopgen_amd/csrc/synth_common.h"

void orc_synth_fbm(uint64_t seed, int64_t n, int64_t m, int64_t j0, int npop, uint32_t miss_thresh,
                   int imputed_bytes, uint8_t* out /* n x m column-major */) {
  uint32_t* pjg = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)npop);
  for (int64_t j = 0; j < m; j++) {
    for (int g = 0; g < npop; g++) pjg[g] = tpg_synth_pjg(seed, (uint64_t)(j0 + j), (uint32_t)g, (uint32_t)npop);
    for (int64_t i = 0; i < n; i++)
      out[(size_t)i + (size_t)j * (size_t)n] =
          tpg_synth_geno(seed, (uint64_t)i, (uint64_t)(j0 + j), pjg[i % npop], miss_thresh, imputed_bytes);
  }
  free(pjg);
}

/* The same panel for a SUBSET of the individuals (rows0: 0-based indices into the full panel) over m loci from j0:
 * lets a test check a 5 000 x 1 000 000 device result against this oracle on a few dozen individuals over ALL loci
 * without generating 5 GB on the host.  out is nrows x m column-major. */
void orc_synth_rows(uint64_t seed, const int64_t* rows0, int64_t nrows, int64_t m, int64_t j0, int npop,
                    uint32_t miss_thresh, int imputed_bytes, uint8_t* out) {
#pragma omp parallel
  {
    uint32_t* pjg = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)npop);
    uint8_t* need = (uint8_t*)calloc((size_t)npop, 1);
    for (int64_t r = 0; r < nrows; r++) need[rows0[r] % npop] = 1;
#pragma omp for schedule(static)
    for (int64_t j = 0; j < m; j++) {
      for (int g = 0; g < npop; g++)
        if (need[g]) pjg[g] = tpg_synth_pjg(seed, (uint64_t)(j0 + j), (uint32_t)g, (uint32_t)npop);
      for (int64_t r = 0; r < nrows; r++)
        out[(size_t)r + (size_t)j * (size_t)nrows] = tpg_synth_geno(seed, (uint64_t)rows0[r], (uint64_t)(j0 + j),
                                                                    pjg[rows0[r] % npop], miss_thresh, imputed_bytes);
    }
    free(pjg);
    free(need);
  }
}
