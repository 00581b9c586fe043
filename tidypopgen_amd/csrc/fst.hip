// fst.hip -- pairwise population Fst (Hudson / WC84 / Nei87).
//
// Replaces pairwise_fst_hudson_loop (src/pairwise_fst_hudson_loop.cpp:5-63),
// pairwise_fst_wc84_loop (src/pairwise_fst_wc84_loop.cpp:5-121), pairwise_fst_nei87_loop
// (src/pairwise_fst_nei87_loop.cpp:5-115) and, in the fused entry point, the m x G summary
// matrices of grouped_summaries_dip_pseudo_cpp that feed them (R/pairwise_pop_fst.R:123-161).
//
// The per-locus per-pair arithmetic is the reference's, statement for statement, in FP64 with
// FMA contraction off (this file is compiled with -ffp-contract=off), so by-locus values match
// bit for bit; only the order of the sum over loci differs (fixed: locus chunks are summed per
// workgroup in ascending order, workgroup partials are summed in ascending order by one thread).
//
// Work decomposition: a workgroup stages the (n, p, h) doubles of LB loci x G populations in LDS
// (layout [locus][population]: threads of a wave hold consecutive pairs, i.e. the same pop1
// (broadcast) and consecutive pop2 (conflict-free)), then every thread owns one population pair
// and walks the LB loci.  FP64-VALU bound (about 12 / 45 / 60 flops per pair-locus for
// Hudson / WC84 / Nei87), not HBM bound: the counts it reads are 12 B per locus-population.
#include <math.h>

#include <algorithm>

#include "common.h"
#include "devfrag.h"
#include "host/host_fsttiles.h"
typedef double v2d __attribute__((ext_vector_type(2)));

#define FST_NAN __longlong_as_double(0x7FF8000000000000ll)

// e1 / e2: the per-(locus, population) term staged once per locus chunk -- Hudson: p q / (n - 1), hoisted out of
// the pair loop with its operation order unchanged (src/pairwise_fst_hudson_loop.cpp:28-29), so values stay
// bit-identical.  FAST (only when no by-locus output is requested): WC84 with 3 divisions instead of 8
// (reciprocals reused); the ratio of sums moves by ~1e-15 relative.
// 1 / x to ~1 ulp: v_rcp_f64 (about 2^-23 relative) and two Newton steps.  x = 0 or inf gives NaN, callers guard.
__device__ __forceinline__ double fst_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(e, r, r);
  e = fma(-x, r, 1.0);
  return fma(e, r, r);
}

template <int METHOD, bool FAST>
__device__ __forceinline__ void fst_terms(double n1, double p1, double h1, double e1, double n2, double p2, double h2,
                                          double e2, double& num, double& den) {
  if (METHOD == TPG_FST_HUDSON) {
    // src/pairwise_fst_hudson_loop.cpp:27-32
    const double q1 = 1 - p1, q2 = 1 - p2;  // freq_ref = 1 - freq_alt (grouped_summaries :53)
    const double d = p1 - p2;
    num = d * d - e1 - e2;
    den = p1 * q2 + p2 * q1;
  } else if (METHOD == TPG_FST_WC84 && FAST) {
    // Sums only (no per-locus output): the same estimator with the two-population algebra done by hand.
    // Staged per (locus, population): n1 = individuals (allele count / 2), h1 = het_obs * individuals, e1 = 1 /
    // individuals.  With r = 2:  n_c = 2 n1 n2 / nt,  s2 = 2 d^2 n1 n2 / nt^2  (d = p1 - p2), so
    // n_bar / n_c * s2 = d^2 / 2 and n_bar / n_c = nt^2 / (4 n1 n2): two reciprocals per pair and locus (by
    // v_rcp_f64 + two Newton steps) instead of three IEEE divisions, ~50 FP64 instructions instead of ~100.
    // Differs from the statement-order path by rounding only (tests: <= 1e-12 relative on the sums).
    const double nt = n1 + n2;
    const double inv_nt = fst_rcp(nt);
    const double nb1 = 0.5 * nt - 1.0;
    double inv_nb1 = fst_rcp(nb1);
    if (nb1 == 0.0) inv_nb1 = HUGE_VAL;  // one individual per population (nb1 = +0): the reference's 1 / 0 = +inf
    const double p_bar = (p1 * n1 + p2 * n2) * inv_nt, h_bar = (h1 + h2) * inv_nt;
    const double d = p1 - p2, hd2 = 0.5 * (d * d);
    const double half_s2 = (2.0 * hd2) * (n1 * n2) * (inv_nt * inv_nt);
    const double core = p_bar * (1.0 - p_bar) - half_s2;
    const double X = (0.25 * (nt * nt)) * (e1 * e2) * inv_nb1;
    const double a = hd2 - X * (core - 0.25 * h_bar);
    const double b = (0.5 * nt) * inv_nb1 * (core - (0.5 * (nt - 1.0)) * inv_nt * h_bar);
    num = a;
    den = a + b + 0.5 * h_bar;
  } else if (METHOD == TPG_FST_WC84) {
    // src/pairwise_fst_wc84_loop.cpp:41-99 with r = 2
    const double r = 2.0;
    const double ni1 = n1 / 2.0, ni2 = n2 / 2.0;
    double sum_n = 0.0, sum_sq = 0.0;
    sum_n += ni1; sum_sq += ni1 * ni1;
    sum_n += ni2; sum_sq += ni2 * ni2;
    const double n_total = sum_n, n_bar = sum_n / 2;
    const double n_c = (sum_n - sum_sq / sum_n) / (2 - 1);
    double sum_pn = 0.0, sum_h = 0.0;
    sum_pn += p1 * ni1; sum_h += h1 * ni1;
    sum_pn += p2 * ni2; sum_h += h2 * ni2;
    const double p_bar = sum_pn / n_total, h_bar = sum_h / n_total;
    double sum_sq_diff = 0.0;
    sum_sq_diff += (p1 - p_bar) * (p1 - p_bar) * ni1;
    sum_sq_diff += (p2 - p_bar) * (p2 - p_bar) * ni2;
    const double s2 = sum_sq_diff / (n_bar * (2 - 1));
    const double a = n_bar / n_c * (s2 - (1.0 / (n_bar - 1.0)) * (p_bar * (1 - p_bar) - ((r - 1.0) / r) * s2 - h_bar / 4.0));
    const double b = n_bar / (n_bar - 1.0) *
                     (p_bar * (1 - p_bar) - ((r - 1.0) / r) * s2 - ((2 * n_bar - 1.0) / (4.0 * n_bar)) * h_bar);
    const double c = h_bar / 2.0;
    num = a;
    den = a + b + c;
  } else {
    // src/pairwise_fst_nei87_loop.cpp:45-80
    const double q1 = 1 - p1, q2 = 1 - p2;
    const double np1 = n1 / 2.0, np2 = n2 / 2.0;
    int valid = 0;
    double nsum = 0.0, inv_nsum = 0.0, ho_sum = 0.0;
    if (np1 == np1) { valid++; ho_sum += h1; nsum += 1.0; inv_nsum += 1.0 / np1; }
    if (np2 == np2) { valid++; ho_sum += h2; nsum += 1.0; inv_nsum += 1.0 / np2; }
    const double np = valid;
    const double mn = (inv_nsum > 0) ? nsum / inv_nsum : FST_NAN;
    const double mHo = ho_sum / valid;
    const double sp2a = p1 * p1 + p2 * p2;
    const double sp2r = q1 * q1 + q2 * q2;
    const double sp2 = sp2a + sp2r;
    const double msp2 = sp2 / 2.0;
    const double fAm = (p1 + p2) / 2.0, fRm = (q1 + q2) / 2.0;
    const double mp2 = fAm * fAm + fRm * fRm;
    const double mHs = mn / (mn - 1.0) * (1.0 - msp2 - mHo / (2.0 * mn));
    const double Ht = 1.0 - mp2 + mHs / (mn * np) - mHo / (2.0 * mn * np);
    const double Dst = Ht - mHs;
    const double Dstp = (np / (np - 1.0)) * Dst;
    num = Dstp;
    den = mHs + Dstp;
  }
}

struct FstSrc {
  // either class counts (fused path) ...
  const int32_t* cnt;
  int64_t Mpad;
  int Cpad;
  int has_hap;
  // ... or the reference's m x G matrices (loop mirror)
  const double* n;
  const double* p;
  const double* q;  // freq_ref; only used to honour a caller-supplied matrix in the mirror
  const double* h;
};

// n (valid alleles), freq_alt and het_obs of population g at locus j, as grouped_summaries_dip_pseudo_cpp forms them
// (src/grouped_summaries_dip_pseudo_cpp.cpp:40-56), from the class counts of the fused path or from the caller's matrices
__device__ __forceinline__ void fst_stage(const FstSrc& src, int64_t m, int64_t j, int g, double& vn, double& vp, double& vh) {
  if (src.cnt) {
    const int64_t plane = src.Mpad * src.Cpad;
    if (!src.has_hap) {
      const int64_t o = j * src.Cpad + g;
      const int n1 = src.cnt[o], n2 = src.cnt[plane + o], nv = src.cnt[2 * plane + o];
      vn = (double)(2 * nv);
      vp = (double)(n1 + 2 * n2) / vn;
      vh = (double)(2 * n1) / vn;
    } else {
      const int64_t o = j * src.Cpad + 2 * g;
      const int n1d = src.cnt[o], n2d = src.cnt[plane + o], nvd = src.cnt[2 * plane + o];
      const int n1h = src.cnt[o + 1], n2h = src.cnt[plane + o + 1], nvh = src.cnt[2 * plane + o + 1];
      vn = (double)(2 * nvd + nvh);
      vp = ((double)(n1d + 2 * n2d) + 0.5 * (double)(n1h + 2 * n2h)) / vn;
      vh = (double)(2 * (n1d + n1h)) / vn;
    }
  } else {
    vn = src.n[j + (int64_t)g * m];
    vp = src.p[j + (int64_t)g * m];
    vh = src.h ? src.h[j + (int64_t)g * m] : 0.0;
  }
}

// PPT population pairs per thread (1 or 8): with P > 256 pairs the (n, p, h) staging of a locus chunk is done
// once per workgroup instead of once per 256 pairs.
template <int METHOD, bool FAST, int PPT>
__global__ __launch_bounds__(256) void tpg_fst_kernel(FstSrc src, int64_t m, int G, int LB,
                                                      const int32_t* __restrict__ pairs0, int P, int by_locus,
                                                      int return_num_dem, double* __restrict__ part,
                                                      double* __restrict__ out_a, double* __restrict__ out_b) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double* sh_n = sh;
  double* sh_p = sh + (size_t)LB * G;
  double* sh_h = sh + 2 * (size_t)LB * G;
  double* sh_e = sh + 3 * (size_t)LB * G;
  int pidx[PPT], g1[PPT], g2[PPT];
  double sum_num[PPT], sum_den[PPT];
#pragma unroll
  for (int k = 0; k < PPT; k++) {
    pidx[k] = (blockIdx.y * PPT + k) * 256 + threadIdx.x;
    g1[k] = 0; g2[k] = 0;
    if (pidx[k] < P) { g1[k] = pairs0[2 * pidx[k]]; g2[k] = pairs0[2 * pidx[k] + 1]; }
    sum_num[k] = 0.0; sum_den[k] = 0.0;
  }
  const int64_t nchunks = (m + LB - 1) / LB;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t j0 = ch * LB;
    __syncthreads();
    for (int idx = threadIdx.x; idx < LB * G; idx += 256) {
      const int l = idx / G, g = idx % G;
      const int64_t j = j0 + l;
      double vn = FST_NAN, vp = FST_NAN, vh = FST_NAN;
      if (j < m) fst_stage(src, m, j, g, vn, vp, vh);
      sh_n[idx] = vn; sh_p[idx] = vp; sh_h[idx] = vh;
      if (METHOD == TPG_FST_HUDSON) sh_e[idx] = (vp * (1 - vp)) / (vn - 1);  // (p q) / (n - 1), once per population
      if (METHOD == TPG_FST_WC84 && FAST) {  // individuals, het_obs * individuals, 1 / individuals
        const double ni = 0.5 * vn;
        sh_n[idx] = ni; sh_h[idx] = vh * ni; sh_e[idx] = 1.0 / ni;
      }
    }
    __syncthreads();
    const int lmax = (int)((m - j0) < LB ? (m - j0) : LB);
#pragma unroll
    for (int k = 0; k < PPT; k++) {
      if (pidx[k] >= P) continue;
#pragma unroll 2
      for (int l = 0; l < lmax; l++) {
        const int o1 = l * G + g1[k], o2 = l * G + g2[k];
        double num, den;
        constexpr bool use_e = METHOD == TPG_FST_HUDSON || (METHOD == TPG_FST_WC84 && FAST);
        fst_terms<METHOD, FAST>(sh_n[o1], sh_p[o1], sh_h[o1], use_e ? sh_e[o1] : 0.0, sh_n[o2], sh_p[o2], sh_h[o2],
                                use_e ? sh_e[o2] : 0.0, num, den);
        if (!FAST && by_locus) {  // FAST instantiations are launched for sums only
          const int64_t o = (j0 + l) + (int64_t)pidx[k] * m;
          if (!return_num_dem) out_a[o] = num / den;
          else { out_a[o] = num; out_b[o] = den; }
        }
        if (num == num && den == den) { sum_num[k] += num; sum_den[k] += den; }  // !is_na(num) && !is_na(den)
      }
    }
  }
#pragma unroll
  for (int k = 0; k < PPT; k++)
    if (pidx[k] < P) {
      part[((int64_t)blockIdx.x * P + pidx[k]) * 2] = sum_num[k];
      part[((int64_t)blockIdx.x * P + pidx[k]) * 2 + 1] = sum_den[k];
    }
}

// sum the per-workgroup partials of every pair: 16 pairs x 16 slices of the partial list per workgroup (the
// slices are summed in a fixed order: deterministic), so ~P/16 workgroups share the read instead of P/256
__global__ __launch_bounds__(256) void tpg_fst_reduce_kernel(const double* __restrict__ part, int nblocks, int P,
                                                             double* __restrict__ fst_tot, double* __restrict__ sum_num,
                                                             double* __restrict__ sum_den) {
  __shared__ double shn[16][17], shd[16][17];
  const int px = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int pi = blockIdx.x * 16 + px;
  double sn = 0.0, sd = 0.0;
  if (pi < P)
    for (int b = sl; b < nblocks; b += 16) { sn += part[((int64_t)b * P + pi) * 2]; sd += part[((int64_t)b * P + pi) * 2 + 1]; }
  shn[sl][px] = sn; shd[sl][px] = sd;
  __syncthreads();
  if (sl == 0 && pi < P) {
    sn = 0.0; sd = 0.0;
    for (int t = 0; t < 16; t++) { sn += shn[t][px]; sd += shd[t][px]; }
    if (fst_tot) fst_tot[pi] = sn / sd;
    if (sum_num) { sum_num[pi] = sn; sum_den[pi] = sd; }
  }
}

// ---------------------------------------------------------------------------
// Hudson, totals only (no per-locus output), as three masked matrix products over the loci.  With m_g = 1 where the
// reference keeps the locus for population g (freq and p q / (n - 1) not NaN, src/pairwise_fst_hudson_loop.cpp:43-52 drops a
// locus for a pair when its numerator or denominator is NaN) and a_g = m_g (p_g^2 - e_g), b_g = m_g, c_g = m_g p_g:
//     sum_j num = sum_j m_1 m_2 [(p_1 - p_2)^2 - e_1 - e_2] = (A B')[g1,g2] + (A B')[g2,g1] - 2 (C C')[g1,g2]
//     sum_j den = sum_j m_1 m_2 [p_1 q_2 + p_2 q_1]         = (C B')[g1,g2] + (C B')[g2,g1] - 2 (C C')[g1,g2]
// i.e. 3 x 2 G^2 flops per locus instead of ~12 per population pair and locus with 8 LDS reads each: 64 x 64 populations per
// workgroup, a 4 x 4 block of (g1, g2) and 48 FP64 sums per thread, operands staged once per chunk of loci in LDS.  The
// sums are formed in another order than the reference's (and than the by-locus path's, which stays statement for
// statement): the totals agree to ~1e-14 relative.
#define FSTG_T 64   // populations per tile side
#define FSTG_LB 16  // loci per staged chunk (32 KiB of LDS: four workgroups per CU hide one another's staging)
// MF: the three products on the FP64 matrix cores (v_mfma_f64_16x16x4_f64: 16 x 16 pairs x 4 loci in 32 cycles, twice the
// FP64 VALU rate, and a lane feeds it ONE double per operand where the VALU form reads 16 doubles for 48 FMAs -- that form
// is bound by LDS bandwidth: 27 M ds_read_b128 + 40 M conflict cycles against 1.32 M cycles per launch at 51 populations).
// A wave owns 16 row populations x all 64 column populations: 4 column tiles x 3 products = 12 accumulators, per 4 loci
// 2 + 8 ds_read_b64 and 12 MFMAs.  Rows of the staged arrays are FSTG_RS = 80 doubles apart (32 dwords modulo the 64
// banks), so that the four loci a wave instruction reads fall on disjoint banks.
#define FSTG_RS_MFMA 80
template <bool MF>
__global__ __launch_bounds__(256) void tpg_fst_hudson_gemm_kernel(FstSrc src, int64_t m, int G, int ntile, int kmax,
                                                                  double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  constexpr int RS = MF ? FSTG_RS_MFMA : FSTG_T;  // row stride of the staged arrays
  double* ra = sh;                      // [l][RS] a of the row populations
  double* rc = sh + FSTG_LB * RS;       // c of the row populations
  double* cb = sh + 2 * FSTG_LB * RS;   // b of the column populations
  const int tR = blockIdx.y / ntile, tC = blockIdx.y % ntile;
  const bool diag = tR == tC;
  double* cc = diag ? rc : sh + 3 * FSTG_LB * RS;  // c of the column populations
  // 1 / A and 1 / (A - 1) for A valid alleles (class counts: A is a small integer), by IEEE divisions once per workgroup:
  // the per-population staging then needs no division (freq = alt * (1 / A) differs from alt / A by at most an ulp; the
  // by-locus path, which must be bit-identical to the reference, divides)
  double* inv = sh + 4 * FSTG_LB * RS;
  for (int A = threadIdx.x; A <= kmax; A += 256) { inv[2 * A] = 1.0 / (double)A; inv[2 * A + 1] = 1.0 / ((double)A - 1.0); }
  // thread grid TG x TG of 4 x 4 blocks, TG = populations of this tile / 4 rounded up: with 51 populations 13 x 13 = 169
  // threads work (three waves) instead of 256 on a padded 64 x 64 tile
  const int gR = min(FSTG_T, G - tR * FSTG_T), gC = min(FSTG_T, G - tC * FSTG_T);
  const int TGy = (gR + 3) / 4, TGx = (gC + 3) / 4;
  const bool work = MF || (int)threadIdx.x < TGy * TGx;
  const int ty = work ? threadIdx.x / TGx : 0, tx = work ? threadIdx.x % TGx : 0;
  double AB[4][4], CB[4][4], CC[4][4];  // VALU form: a 4 x 4 block of pairs; MFMA form: [column tile][C/D register]
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int c = 0; c < 4; c++) { AB[r][c] = 0.0; CB[r][c] = 0.0; CC[r][c] = 0.0; }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r16 = lane & 15, kq = lane >> 4;
  const int64_t nchunks = (m + FSTG_LB - 1) / FSTG_LB;
  // class counts of diploids (kmax > 0): the three counts of this thread's (locus, population) slots of the NEXT chunk are
  // fetched into registers before the products of the current one, so that their latency hides behind the FMAs
  constexpr int NSLOT = 2 * FSTG_LB * FSTG_T / 256;
  const int nslot = (diag ? 1 : 2) * FSTG_LB * FSTG_T / 256;
  int pn1[NSLOT], pn2[NSLOT], pnv[NSLOT];
  auto fetch = [&](int64_t ch) {
#pragma unroll
    for (int i = 0; i < NSLOT; i++) {
      pn1[i] = 0; pn2[i] = 0; pnv[i] = 0;
      if (i < nslot && ch < nchunks) {
        const int idx = threadIdx.x + 256 * i;
        const int side = idx / (FSTG_LB * FSTG_T), rem = idx % (FSTG_LB * FSTG_T);
        const int g = (side ? tC : tR) * FSTG_T + rem % FSTG_T;
        const int64_t j = ch * FSTG_LB + rem / FSTG_T;
        if (j < m && g < G) {
          const int64_t plane = src.Mpad * src.Cpad, o = j * src.Cpad + g;
          pn1[i] = src.cnt[o]; pn2[i] = src.cnt[plane + o]; pnv[i] = src.cnt[2 * plane + o];
        }
      }
    }
  };
  if (kmax > 0) fetch(blockIdx.x);
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t j0 = ch * FSTG_LB;
    __syncthreads();
    if (kmax > 0) {
#pragma unroll
      for (int i = 0; i < NSLOT; i++) {
        if (i >= nslot) break;
        const int idx = threadIdx.x + 256 * i;
        const int side = idx / (FSTG_LB * FSTG_T), rem = idx % (FSTG_LB * FSTG_T);
        double a = 0.0, b = 0.0, c = 0.0;
        const int A = min(2 * pnv[i], kmax);
        if (A > 0) {  // no valid genotype (or a slot past the data): freq is NaN, the population drops out
          const double vp = (double)(pn1[i] + 2 * pn2[i]) * inv[2 * A];
          const double e = (vp * (1 - vp)) * inv[2 * A + 1];
          if (e == e) { a = vp * vp - e; b = 1.0; c = vp; }
        }
        const int srem = (rem / FSTG_T) * RS + rem % FSTG_T;
        if (side) { cb[srem] = b; cc[srem] = c; }
        else {
          ra[srem] = a; rc[srem] = c;
          if (diag) cb[srem] = b;
        }
      }
    } else {
      // a tile on the diagonal (always, up to 64 populations) stages its populations once: rows and columns are the same
      for (int idx = threadIdx.x; idx < (diag ? 1 : 2) * FSTG_LB * FSTG_T; idx += 256) {
        const int side = idx / (FSTG_LB * FSTG_T), rem = idx % (FSTG_LB * FSTG_T);
        const int l = rem / FSTG_T, gl = rem % FSTG_T;
        const int g = (side ? tC : tR) * FSTG_T + gl;
        const int64_t j = j0 + l;
        double a = 0.0, b = 0.0, c = 0.0;
        if (j < m && g < G) {
          double vn, vp, vh;
          fst_stage(src, m, j, g, vn, vp, vh);
          const double e = (vp * (1 - vp)) / (vn - 1);  // src/pairwise_fst_hudson_loop.cpp:28-29
          if (vp == vp && e == e) { a = vp * vp - e; b = 1.0; c = vp; }
        }
        const int srem = l * RS + gl;
        if (side) { cb[srem] = b; cc[srem] = c; }
        else {
          ra[srem] = a; rc[srem] = c;
          if (diag) cb[srem] = b;
        }
      }
    }
    __syncthreads();
    if (kmax > 0) fetch(ch + gridDim.x);
    if constexpr (MF) {
      // A = x[locus l + kq][row population 16 wv + r16], B = y[locus l + kq][column population 16 ct + r16]
#pragma unroll
      for (int l = 0; l < FSTG_LB; l += 4) {
        const double av = ra[(l + kq) * RS + 16 * wv + r16], cv = rc[(l + kq) * RS + 16 * wv + r16];
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
          const double bv = cb[(l + kq) * RS + 16 * ct + r16], dv = cc[(l + kq) * RS + 16 * ct + r16];
          *(v4d*)AB[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, *(v4d*)AB[ct], 0, 0, 0);
          *(v4d*)CB[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(cv, bv, *(v4d*)CB[ct], 0, 0, 0);
          *(v4d*)CC[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(cv, dv, *(v4d*)CC[ct], 0, 0, 0);
        }
      }
    } else if (work) {
#pragma unroll 2
      for (int l = 0; l < FSTG_LB; l++) {
        const v4d a4 = *(const v4d*)&ra[l * FSTG_T + 4 * ty], c4 = *(const v4d*)&rc[l * FSTG_T + 4 * ty];
        const v4d b4 = *(const v4d*)&cb[l * FSTG_T + 4 * tx], d4 = *(const v4d*)&cc[l * FSTG_T + 4 * tx];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int c = 0; c < 4; c++) {
            AB[r][c] = fma(a4[r], b4[c], AB[r][c]);
            CB[r][c] = fma(c4[r], b4[c], CB[r][c]);
            CC[r][c] = fma(c4[r], d4[c], CC[r][c]);
          }
      }
    }
  }
  // partial sums of this workgroup: [block x][tile][product][row][col]
  if (!work) return;
  double* o = part + (((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 3) * FSTG_T * FSTG_T;
  if constexpr (MF) {  // C/D: column = lane & 15, row = (lane >> 4) + 4 reg
#pragma unroll
    for (int ct = 0; ct < 4; ct++)
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int q = (16 * wv + kq + 4 * reg) * FSTG_T + 16 * ct + r16;
        o[q] = AB[ct][reg];
        o[FSTG_T * FSTG_T + q] = CB[ct][reg];
        o[2 * FSTG_T * FSTG_T + q] = CC[ct][reg];
      }
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int q = (4 * ty + r) * FSTG_T + 4 * tx + c;
      o[q] = AB[r][c];
      o[FSTG_T * FSTG_T + q] = CB[r][c];
      o[2 * FSTG_T * FSTG_T + q] = CC[r][c];
    }
}

// full[slice][tile][product][row][col] = sum over every FSTG_SL-th workgroup's partials, in workgroup order; the final
// kernel adds the FSTG_SL slices in slice order (the same sums on every run)
#define FSTG_SL 16
__global__ __launch_bounds__(256) void tpg_fst_hudson_gemm_reduce_kernel(const double* __restrict__ part, int nbx, int64_t cells,
                                                                         double* __restrict__ full) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= cells) return;
  double s = 0.0;
  for (int b = blockIdx.y; b < nbx; b += FSTG_SL) s += part[(int64_t)b * cells + q];
  full[(int64_t)blockIdx.y * cells + q] = s;
}

__global__ void tpg_fst_hudson_gemm_final_kernel(const double* __restrict__ full, int64_t cells, int ntile,
                                                 const int32_t* __restrict__ pairs0, int P, double* __restrict__ fst_tot,
                                                 double* __restrict__ sum_num, double* __restrict__ sum_den) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= P) return;
  const int g1 = pairs0[2 * pi], g2 = pairs0[2 * pi + 1];
  auto at = [&](int prod, int r, int c) {
    const int t = (r / FSTG_T) * ntile + c / FSTG_T;
    const int64_t q = (((int64_t)t * 3 + prod) * FSTG_T + r % FSTG_T) * FSTG_T + c % FSTG_T;
    double v = 0.0;
    for (int sl = 0; sl < FSTG_SL; sl++) v += full[(int64_t)sl * cells + q];
    return v;
  };
  const double cc2 = 2 * at(2, g1, g2);
  const double sn = at(0, g1, g2) + at(0, g2, g1) - cc2, sd = at(1, g1, g2) + at(1, g2, g1) - cc2;
  if (fst_tot) fst_tot[pi] = sn / sd;
  if (sum_num) { sum_num[pi] = sn; sum_den[pi] = sd; }
}

// ---------------------------------------------------------------------------
// WC84, totals only, on the class counts of the fused path.  The estimator is the FAST form of fst_terms above; what
// is new is where the reciprocals come from.  Everything in it that depends on the sample sizes ALONE depends on the
// number of valid alleles of the pair, A = A_1 + A_2 -- a small integer (at most four times the largest group) -- so
// 1 / nt, (nt - 1) / (2 nt), nt / (2 (nt / 2 - 1)) and nt^2 / (4 (nt / 2 - 1)) (nt = A / 2 individuals) are tabulated
// once per workgroup by IEEE divisions (closer to the reference than v_rcp_f64 + Newton steps) and a pair and locus
// costs ~30 FP64 instructions and one 32-byte table read instead of ~60 instructions.
// Staged per (locus, population) as four doubles {individuals, freq_alt, het_obs * individuals, 1 / individuals}.
#define FSTW_TAB 4  // doubles per table entry: {1 / nt, (nt - 1) / (2 nt), nt / (2 nb1), nt^2 / (4 nb1)}
// GS = stride of the staged arrays in populations: 64 (a compile-time constant: every LDS address of the pair loop is
// then a per-pair base register + an immediate offset, no integer arithmetic inside it) when G <= 64, else 0 = G itself.
template <int PPT, int GS>
__global__ __launch_bounds__(256) void tpg_fst_wc84_tab_kernel(FstSrc src, int64_t m, int G, int LB, int kmax,
                                                               const int32_t* __restrict__ pairs0, int P,
                                                               double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int gs = GS ? GS : G;
  // byte offsets inside the dynamic LDS block: staged {n, p} [l][g] and {H, e} [l][g] (16-byte entries: the pairs of a wave
  // read consecutive populations, conflict-free at a 16-byte stride, two-way conflicts at 32), then the two tables [A]
  const uint32_t sb_b = (uint32_t)LB * gs * 16u, tab_b = (uint32_t)LB * gs * 32u, tab2_b = tab_b + (uint32_t)(kmax + 1) * 16u;
  char* shb = (char*)sh;
  for (int A = threadIdx.x; A <= kmax; A += 256) {
    const double nt = 0.5 * (double)A, nb1 = 0.5 * nt - 1.0;
    const double r = 1.0 / nt, sv = 1.0 / nb1;  // nb1 = 0 (one individual per population): +inf, as the reference's 1 / 0
    // two tables of 16-byte entries, {t0, t1} and {t2, t3}: a 32-byte entry put the 30-odd entries a wave touches (the
    // pairs' valid-allele counts differ by their missing genotypes) on FOUR bank positions, three conflict cycles per
    // LDS instruction (rocprofv3 --pmc: SQ_LDS_BANK_CONFLICT 3.9e8 against SQ_ACTIVE_INST_LDS 1.3e8)
    double* t = (double*)(shb + tab_b) + (size_t)A * 2;
    t[0] = r;
    t[1] = (0.5 * (nt - 1.0)) * r;
    double* t2 = (double*)(shb + tab2_b) + (size_t)A * 2;
    t2[0] = (0.5 * nt) * sv;
    t2[1] = (0.25 * (nt * nt)) * sv;
  }
  int pidx[PPT];
  uint32_t o1[PPT], o2[PPT];  // the pair's two populations
  double sum_num[PPT], sum_den[PPT];
#pragma unroll
  for (int k = 0; k < PPT; k++) {
    pidx[k] = (blockIdx.y * PPT + k) * 256 + threadIdx.x;
    o1[k] = 0; o2[k] = 0;
    if (pidx[k] < P) { o1[k] = (uint32_t)pairs0[2 * pidx[k]]; o2[k] = (uint32_t)pairs0[2 * pidx[k] + 1]; }
    sum_num[k] = 0.0; sum_den[k] = 0.0;
  }
  const int64_t nchunks = (m + LB - 1) / LB;
  // the three class counts of this thread's (locus, population) slots of the NEXT chunk are fetched into registers before
  // the pair loop of the current one (their latency hides behind it); FSTW_SLOTS x 256 >= LB * G
  constexpr int FSTW_SLOTS = 8;
  int pn1[FSTW_SLOTS], pn2[FSTW_SLOTS], pnv[FSTW_SLOTS];
  auto fetch = [&](int64_t ch) {
    const int64_t plane = src.Mpad * src.Cpad;
#pragma unroll
    for (int i = 0; i < FSTW_SLOTS; i++) {
      pn1[i] = 0; pn2[i] = 0; pnv[i] = 0;
      const int idx = threadIdx.x + 256 * i;
      if (idx < LB * G && ch < nchunks) {
        const int64_t j = ch * LB + idx / G;
        if (j < m) {
          const int64_t o = j * src.Cpad + idx % G;
          pn1[i] = src.cnt[o]; pn2[i] = src.cnt[plane + o]; pnv[i] = src.cnt[2 * plane + o];
        }
      }
    }
  };
  fetch(blockIdx.x);
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t j0 = ch * LB;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FSTW_SLOTS; i++) {
      const int idx = threadIdx.x + 256 * i;
      if (idx < LB * G) {
        // src/grouped_summaries_dip_pseudo_cpp.cpp:40-56 on the counts (diploids): n = 2 valid, freq, het_obs
        const double vn = (double)(2 * pnv[i]);
        const double vp = (double)(pn1[i] + 2 * pn2[i]) / vn, vh = (double)(2 * pn1[i]) / vn;  // 0 / 0 = NaN: no valid genotype
        const double ni = 0.5 * vn;
        const uint32_t q = (uint32_t)((idx / G) * gs + idx % G);
        *(v2d*)(shb + q * 16u) = v2d{ni, vp};
        *(v2d*)(shb + sb_b + q * 16u) = v2d{vh * ni, 1.0 / ni};
      }
    }
    __syncthreads();
    fetch(ch + gridDim.x);
    const int lmax = (int)((m - j0) < LB ? (m - j0) : LB);
#pragma unroll
    for (int k = 0; k < PPT; k++) {
      if (pidx[k] >= P) continue;
      const uint32_t a1 = o1[k] * 16u, a2 = o2[k] * 16u;
      double sn = sum_num[k], sd = sum_den[k];
#pragma unroll 4
      for (int l = 0; l < lmax; l++) {
#pragma clang fp contract(fast)  // totals only (1e-11 of the exact sum is the contract here, not the reference's rounding)
        const uint32_t lo = (uint32_t)l * (uint32_t)gs;
        const v2d s1a = *(const v2d*)(shb + a1 + lo * 16u), s2a = *(const v2d*)(shb + a2 + lo * 16u);                  // {n, p}
        const v2d s1b = *(const v2d*)(shb + sb_b + a1 + lo * 16u), s2b = *(const v2d*)(shb + sb_b + a2 + lo * 16u);    // {H, e}
        const v4d s1 = v4d{s1a[0], s1a[1], s1b[0], s1b[1]}, s2 = v4d{s2a[0], s2a[1], s2b[0], s2b[1]};
        // valid alleles of the pair = 2 (n1 + n2): an exact small integer in FP64 (an empty population has n = 0)
        const int A = min(__double2int_rz(2.0 * (s1[0] + s2[0])), kmax);
        const v2d ta = *(const v2d*)(shb + tab_b + (uint32_t)A * 16u), tb = *(const v2d*)(shb + tab2_b + (uint32_t)A * 16u);
        const v4d t = v4d{ta[0], ta[1], tb[0], tb[1]};
        const double p_bar = fma(s2[1], s2[0], s1[1] * s1[0]) * t[0], h_bar = (s1[2] + s2[2]) * t[0];
        const double d = s1[1] - s2[1], d2 = d * d;
        const double half_s2 = (d2 * (s1[0] * s2[0])) * (t[0] * t[0]);
        const double core = fma(-p_bar, p_bar, p_bar) - half_s2;
        const double X = t[3] * (s1[3] * s2[3]);
        const double a = fma(-X, fma(-0.25, h_bar, core), 0.5 * d2);
        const double b = t[2] * fma(-t[1], h_bar, core);
        const double den = fma(0.5, h_bar, a + b);
        if (den == den) { sn += a; sd += den; }  // a NaN numerator makes the denominator NaN too
      }
      sum_num[k] = sn; sum_den[k] = sd;
    }
  }
#pragma unroll
  for (int k = 0; k < PPT; k++)
    if (pidx[k] < P) {
      part[((int64_t)blockIdx.x * P + pidx[k]) * 2] = sum_num[k];
      part[((int64_t)blockIdx.x * P + pidx[k]) * 2 + 1] = sum_den[k];
    }
}

// The same for MANY pairs (more than 256) of at most 64 populations: a thread owns a TILE of FSTW_TR x FSTW_TC populations =
// up to 6 pairs instead of 8 unrelated pairs.  The kernel above is bound by LDS bandwidth AND close to its VALU bound
// (rocprofv3 --pmc at 51 populations: 1.2e8 ds_read_b128 x 8 cycles / 256 CUs = 3.8 of the kernel's 3.97 million cycles, 7.6e8
// VALU x 4 / 1 024 SIMDs = 3.0): six 16-byte reads per pair and locus, four of them the pair's two populations, and 38 VALU
// instructions of which 24 are the FP64 arithmetic.  Here
//   * a tile reads its 3 + 2 populations ONCE per locus -- 1.67 reads per pair -- and only the two table reads stay per
//     pair: 3.67 reads instead of 6.  The tiles come from the host (any list of pairs: a tile holds the listed pairs that fall
//     into it, -1 elsewhere; all pairs of 51 populations: 242 tiles = one workgroup of four full waves, 88 % of their slots
//     used; 4 x 2 tiles need 194 registers and leave a CU with six waves: 2.4 ms);
//   * every LDS address is a per-lane pointer that advances by two loci per iteration + an IMMEDIATE offset (the layout is a
//     compile-time constant: tables of 512 entries first, 16 loci x 64 populations after them), where the kernel above
//     adds a loop-dependent offset to every one of its addresses in the VALU;
//   * a population is staged as {32 n, p} and {32 H, 1 / n}: 32 (n_1 + n_2) IS the byte offset of the pair's table entry
//     (16 bytes per valid allele), and the powers of two cancel exactly against the table's 1 / (32 nt): no doubling, no
//     shift, the same roundings;
//   * "is the denominator a number" is decided per WAVE and locus (one ballot of the six comparisons): the common case
//     adds without v_cndmask (4 per pair).
// 26 VALU instructions per pair and locus instead of 38.  Per pair the arithmetic, its order, and the order of the loci are
// those of the kernel above; the staged frequency is a product with the table's 1 / n instead of a quotient (one ulp), so the
// sums agree with that kernel's to 1e-16, not bit for bit.
// FSTW_TR x FSTW_TC = 3 x 2 populations; a task record = {first row population, first column population, pair index x 6}:
// host/host_fsttiles.h (fst_wc84_tiles cuts the pair list into them)
#define FSTT_KMAX 511                            // table entries 0 .. 511 valid alleles of a pair
#define FSTT_LB 16
#define FSTT_TAB2 (512 * 16)
#define FSTT_SA (2 * 512 * 16)                   // {32 n, p} [locus][population]
#define FSTT_SB (FSTT_SA + FSTT_LB * 64 * 16)    // {32 H, 1 / n}
#define FSTT_LDS (FSTT_SB + FSTT_LB * 64 * 16)
__global__ __launch_bounds__(256, 3) void tpg_fst_wc84_tile_kernel(FstSrc src, int64_t m, int G, int kmax,
                                                                   const int32_t* __restrict__ tasks, int ntask, int P,
                                                                   double* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) char shb[FSTT_LDS];
  constexpr int NPT = FSTW_TR * FSTW_TC, LB = FSTT_LB, GS = 64;
  for (int A = threadIdx.x; A <= kmax; A += 256) {
    const double nt = 0.5 * (double)A, nb1 = 0.5 * nt - 1.0;
    const double r = 1.0 / nt, sv = 1.0 / nb1;
    double* t = (double*)shb + (size_t)A * 2;
    t[0] = r * 0.03125;  // against the staged 32 n, 32 H
    t[1] = (0.5 * (nt - 1.0)) * r;
    double* t2 = (double*)(shb + FSTT_TAB2) + (size_t)A * 2;
    t2[0] = (0.5 * nt) * sv;
    t2[1] = (0.25 * (nt * nt)) * sv;
  }
  const int task = blockIdx.y * 256 + threadIdx.x;
  const bool active = task < ntask;
  int pidx[NPT];
  uint32_t ro[FSTW_TR], co[FSTW_TC];  // byte offsets of the tile's populations inside a locus' staged row
  double sum_num[NPT], sum_den[NPT];
  {
    const int r0 = active ? tasks[task * FSTW_TASK_INTS] : 0, c0 = active ? tasks[task * FSTW_TASK_INTS + 1] : 0;
#pragma unroll
    for (int i = 0; i < FSTW_TR; i++) ro[i] = (uint32_t)min(r0 + i, G - 1) * 16u;
#pragma unroll
    for (int j = 0; j < FSTW_TC; j++) co[j] = (uint32_t)min(c0 + j, G - 1) * 16u;
#pragma unroll
    for (int k = 0; k < NPT; k++) {
      pidx[k] = active ? tasks[task * FSTW_TASK_INTS + 2 + k] : -1;
      sum_num[k] = 0.0; sum_den[k] = 0.0;
    }
  }
  const int64_t nchunks = (m + LB - 1) / LB;
  constexpr int SLOTS = LB * GS / 256;  // staged (locus, population) slots per thread
  int pn1[SLOTS], pn2[SLOTS], pnv[SLOTS];
  auto fetch = [&](int64_t ch) {
    const int64_t plane = src.Mpad * src.Cpad;
#pragma unroll
    for (int i = 0; i < SLOTS; i++) {
      pn1[i] = 0; pn2[i] = 0; pnv[i] = 0;
      // slot = (locus (threadIdx.x >> 6) + 4 i, population threadIdx.x & 63): no division by G, and a wave reads one locus' row
      if ((int)(threadIdx.x & 63u) < G && ch < nchunks) {
        const int64_t j = ch * LB + (threadIdx.x >> 6) + 4 * i;
        if (j < m) {
          const int64_t o = j * src.Cpad + (threadIdx.x & 63u);
          pn1[i] = src.cnt[o]; pn2[i] = src.cnt[plane + o]; pnv[i] = src.cnt[2 * plane + o];
        }
      }
    }
  };
  // one locus of the tile: the row populations at pr[] + OFF, the column populations at pc[] + OFF
  // (LDS pointers by their address space: as generic pointers kept in arrays across the loop they become 64-bit FLAT loads)
  typedef __attribute__((address_space(3))) const char lchar;
  typedef __attribute__((address_space(3))) const v2d lv2d;
  auto locus = [&](lchar* const* pr, lchar* const* pc, auto off) {
#pragma clang fp contract(fast)  // (as in the kernel above)
    constexpr int OFF = decltype(off)::value;
    v4d R[FSTW_TR], Cc[FSTW_TC];  // {32 n, p, 32 H, 1 / n}
#pragma unroll
    for (int i = 0; i < FSTW_TR; i++) {
      const v2d a = *(lv2d*)(pr[i] + OFF), b = *(lv2d*)(pr[i] + OFF + (FSTT_SB - FSTT_SA));
      R[i] = v4d{a[0], a[1], b[0], b[1]};
    }
#pragma unroll
    for (int j = 0; j < FSTW_TC; j++) {
      const v2d a = *(lv2d*)(pc[j] + OFF), b = *(lv2d*)(pc[j] + OFF + (FSTT_SB - FSTT_SA));
      Cc[j] = v4d{a[0], a[1], b[0], b[1]};
    }
    double av[NPT], dv[NPT];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < FSTW_TR; i++)
#pragma unroll
      for (int j = 0; j < FSTW_TC; j++) {
        const int k = i * FSTW_TC + j;
        const v4d s1 = R[i], s2 = Cc[j];
        // 16 bytes per valid allele of the pair: 32 (n1 + n2), an exact small integer in FP64
        const int A16 = __double2int_rz(s1[0] + s2[0]);  // <= 16 kmax (the staging clamps n)
        const v2d ta = *(const v2d*)(shb + A16), tb = *(const v2d*)(shb + A16 + FSTT_TAB2);
        const v4d t = v4d{ta[0], ta[1], tb[0], tb[1]};
        const double p_bar = fma(s2[1], s2[0], s1[1] * s1[0]) * t[0], h_bar = (s1[2] + s2[2]) * t[0];
        const double d = s1[1] - s2[1], d2 = d * d;
        const double half_s2 = (d2 * (s1[0] * s2[0])) * (t[0] * t[0]);
        const double core = fma(-p_bar, p_bar, p_bar) - half_s2;
        const double X = t[3] * (s1[3] * s2[3]);
        const double a = fma(-X, fma(-0.25, h_bar, core), 0.5 * d2);
        const double b = t[2] * fma(-t[1], h_bar, core);
        const double den = fma(0.5, h_bar, a + b);
        av[k] = a; dv[k] = den;
        ok = ok && den == den;
      }
    if (__all(ok)) {  // (an unused slot of the tile sums what nobody reads)
#pragma unroll
      for (int k = 0; k < NPT; k++) { sum_num[k] += av[k]; sum_den[k] += dv[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < NPT; k++)
        if (dv[k] == dv[k]) { sum_num[k] += av[k]; sum_den[k] += dv[k]; }  // a NaN numerator makes the denominator NaN too
    }
  };
  fetch(blockIdx.x);
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t j0 = ch * LB;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SLOTS; i++) {
      if ((int)(threadIdx.x & 63u) < G) {
        // src/grouped_summaries_dip_pseudo_cpp.cpp:40-56 on the counts (diploids): n = 2 valid, freq, het_obs -- without a
        // division: 1 / n is the table's own 1 / nt at 2 n valid alleles (32 t[0], exact), freq = alleles * (0.5 / n) (one ulp
        // from the quotient: these are the totals, 1e-11 is the contract), and H = het_obs * n IS the heterozygote count
        // (kmax = 4 x the largest population: n <= kmax / 4 whenever the counts belong to these populations; the clamp is what
        // keeps the pair's table offset 32 (n1 + n2) inside the table if they do not, instead of a test per pair and locus)
        const int nv = min(pnv[i], kmax >> 2);
        const double rn = 32.0 * *(const double*)(shb + (uint32_t)nv * 32u);  // 1 / n; n = 0: inf, and 0 * inf = NaN below, as 0 / 0 was
        const double ni = (double)nv;
        const double vp = (double)(pn1[i] + 2 * pn2[i]) * (0.5 * rn);
        const uint32_t q = threadIdx.x + 256u * (uint32_t)i;
        *(v2d*)(shb + FSTT_SA + q * 16u) = v2d{32.0 * ni, vp};
        *(v2d*)(shb + FSTT_SB + q * 16u) = v2d{32.0 * (double)pn1[i], rn};
      }
    }
    __syncthreads();
    fetch(ch + gridDim.x);
    const int lmax = (int)((m - j0) < LB ? (m - j0) : LB);
    if (!active) continue;
    lchar *pr[FSTW_TR], *pc[FSTW_TC];
#pragma unroll
    for (int i = 0; i < FSTW_TR; i++) pr[i] = (lchar*)(shb + FSTT_SA + ro[i]);
#pragma unroll
    for (int j = 0; j < FSTW_TC; j++) pc[j] = (lchar*)(shb + FSTT_SA + co[j]);
    int l = 0;
    for (; l + 2 <= lmax; l += 2) {
      locus(pr, pc, std::integral_constant<int, 0>{});
      locus(pr, pc, std::integral_constant<int, GS * 16>{});
#pragma unroll
      for (int i = 0; i < FSTW_TR; i++) pr[i] += 2 * GS * 16;
#pragma unroll
      for (int j = 0; j < FSTW_TC; j++) pc[j] += 2 * GS * 16;
    }
    if (l < lmax) locus(pr, pc, std::integral_constant<int, 0>{});
  }
#pragma unroll
  for (int k = 0; k < NPT; k++)
    if (pidx[k] >= 0) {
      part[((int64_t)blockIdx.x * P + pidx[k]) * 2] = sum_num[k];
      part[((int64_t)blockIdx.x * P + pidx[k]) * 2 + 1] = sum_den[k];
    }
}

// flag[0] = 1 if some freq_ref entry is not exactly 1 - freq_alt (NaN matches NaN)
__global__ void tpg_freq_ref_check_kernel(const double* __restrict__ p, const double* __restrict__ q, int64_t total,
                                          int* __restrict__ flag) {
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const double want = 1 - p[idx], got = q[idx];
    if (!(want == got || (want != want && got != got))) flag[0] = 1;
  }
}

static int run_fst(tpg_ctx* ctx, int method, FstSrc src, int64_t m, int G, const int32_t* pairs1, int P, int by_locus,
                   int return_num_dem, double* fst_tot, double* out_a, double* out_b, double* sum_num = nullptr,
                   double* sum_den = nullptr, int kmax = 0 /* valid alleles of a pair at most (0: unknown) */) {
  TPG_REQUIRE(method == TPG_FST_HUDSON || method == TPG_FST_NEI87 || method == TPG_FST_WC84, TPG_EINVAL,
              "unknown Fst method %d", method);
  TPG_REQUIRE(P > 0 && pairs1, TPG_EINVAL, "no population pairs");
  if (return_num_dem) by_locus = 1;  // R/pairwise_pop_fst.R:103-106
  TPG_REQUIRE(!by_locus || out_a, TPG_EINVAL, "by_locus output requested but out_a is NULL");
  TPG_REQUIRE(!return_num_dem || out_b, TPG_EINVAL, "return_num_dem requested but out_b is NULL");
  TPG_REQUIRE(return_num_dem || fst_tot || sum_num, TPG_EINVAL, "fst_tot is NULL");
  TPG_REQUIRE((sum_num == nullptr) == (sum_den == nullptr), TPG_EINVAL, "sum_num and sum_den go together");
  std::vector<int32_t> p0((size_t)2 * P);
  for (int k = 0; k < 2 * P; k++) {
    TPG_REQUIRE(pairs1[k] >= 1 && pairs1[k] <= G, TPG_EINVAL, "pairwise_combn[%d] = %d out of [1,%d]", k, pairs1[k], G);
    p0[(size_t)k] = pairs1[k] - 1;
  }
  InBuf pb;
  TPG_TRY(pb.init(ctx, p0.data(), sizeof(int32_t) * 2 * (size_t)P));
  int LB = 32;
  while (LB > 1 && (size_t)LB * G * 4 * sizeof(double) > 96 * 1024) LB /= 2;
  TPG_REQUIRE((size_t)LB * G * 4 * sizeof(double) <= 150 * 1024, TPG_EUNSUPPORTED, "too many populations (%d)", G);
  const size_t shmem = (size_t)LB * G * 4 * sizeof(double);
  const bool fast = !by_locus;  // exact statement order whenever per-locus values are returned
  if (fast && method == TPG_FST_HUDSON) {  // totals only: three masked matrix products over the loci
    const int ntile = (int)ceil_div(G, FSTG_T);
    const int64_t cells = (int64_t)ntile * ntile * 3 * FSTG_T * FSTG_T;
    const int64_t nch = ceil_div(m, FSTG_LB);
    int nbx = (int)std::min<int64_t>(nch, std::max(1, 4 * ctx->num_cu / (ntile * ntile)));
    OutBuf ot, osn, osd;
    if (fst_tot) TPG_TRY(ot.init(fst_tot, sizeof(double) * (size_t)P));
    if (sum_num) { TPG_TRY(osn.init(sum_num, sizeof(double) * (size_t)P)); TPG_TRY(osd.init(sum_den, sizeof(double) * (size_t)P)); }
    double* d_gp = nullptr;
    TPG_HIP(tpg_pmalloc((void**)&d_gp, sizeof(double) * (size_t)cells * (size_t)(nbx + FSTG_SL)));
    double* d_full = d_gp + (size_t)cells * (size_t)nbx;
    // reciprocals by table when the source is the class counts of diploids and the table fits (kmax valid alleles at most)
    const int kq = (src.cnt && !src.has_hap && kmax > 0 && kmax <= 4096) ? kmax : 0;
    // TPG_FST_HUDSON_VALU=1: the three products on the FP64 VALU (A/B; rounds 3 and 4)
    static const bool valu = getenv("TPG_FST_HUDSON_VALU") && atoi(getenv("TPG_FST_HUDSON_VALU")) != 0;
    const size_t shg = sizeof(double) * (4 * FSTG_LB * (valu ? FSTG_T : FSTG_RS_MFMA) + 2 * ((size_t)kq + 1));
    if (valu) {
      (void)hipFuncSetAttribute((const void*)tpg_fst_hudson_gemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shg);
      TPG_LAUNCH(ctx, "fst_hudson", tpg_fst_hudson_gemm_kernel<false>, dim3((unsigned)nbx, (unsigned)(ntile * ntile)), dim3(256), shg,
                 src, m, G, ntile, kq, d_gp);
    } else {
      (void)hipFuncSetAttribute((const void*)tpg_fst_hudson_gemm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shg);
      TPG_LAUNCH(ctx, "fst_hudson", tpg_fst_hudson_gemm_kernel<true>, dim3((unsigned)nbx, (unsigned)(ntile * ntile)), dim3(256), shg,
                 src, m, G, ntile, kq, d_gp);
    }
    TPG_LAUNCH(ctx, "fst_reduce", tpg_fst_hudson_gemm_reduce_kernel, dim3((unsigned)ceil_div(cells, 256), FSTG_SL), dim3(256), 0,
               (const double*)d_gp, nbx, cells, d_full);
    TPG_LAUNCH(ctx, "fst_reduce", tpg_fst_hudson_gemm_final_kernel, dim3((unsigned)ceil_div(P, 256)), dim3(256), 0, (const double*)d_full,
               cells, ntile, pb.dev<int32_t>(), P, ot.dev<double>(), osn.dev<double>(), osd.dev<double>());
    hipError_t e = hipGetLastError();
    tpg_pfree(d_gp);  // stream-ordered
    if (e != hipSuccess) { tpg_set_error("fst kernels: %s", hipGetErrorString(e)); return TPG_EHIP; }
    if (fst_tot) TPG_TRY(ot.commit(ctx));
    if (sum_num) { TPG_TRY(osn.commit(ctx)); TPG_TRY(osd.commit(ctx)); }
    return TPG_OK;
  }
  const int64_t nchunks = ceil_div(m, LB);
  const int ppt = P > 256 ? 8 : 1;
  const int ypass = (int)ceil_div(P, 256 * ppt);
  int nblocks = (int)(nchunks < 4 * ctx->num_cu ? nchunks : 4 * ctx->num_cu);
  if (nblocks < 1) nblocks = 1;
  double* d_part = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_part, sizeof(double) * 2 * (size_t)nblocks * (size_t)P));
  OutBuf ot, oa, ob, osn, osd;
  int rc = TPG_OK;
  if (sum_num) { rc = osn.init(sum_num, sizeof(double) * (size_t)P); if (rc == TPG_OK) rc = osd.init(sum_den, sizeof(double) * (size_t)P); }
  const size_t mp = sizeof(double) * (size_t)m * (size_t)P;
  if (rc == TPG_OK && fst_tot) rc = ot.init(fst_tot, sizeof(double) * (size_t)P);
  if (rc == TPG_OK && by_locus) rc = oa.init(out_a, mp);
  if (rc == TPG_OK && return_num_dem) rc = ob.init(out_b, mp);
  // the table kernel stages 32 bytes per (locus, population), at a stride of 64 populations when G <= 64
  const int gs_tab = G <= 64 ? 64 : G, lb_tab = G <= 64 ? 16 : LB;
  const size_t sh_tab = (size_t)lb_tab * gs_tab * 32 + (size_t)(kmax + 1) * FSTW_TAB * 8;
  const bool wc84_tab = fast && method == TPG_FST_WC84 && src.cnt && !src.has_hap && kmax > 0 && sh_tab <= 150 * 1024 &&
                        (int64_t)lb_tab * G <= 8 * 256;  // a thread prefetches at most 8 (locus, population) slots
  if (rc == TPG_OK && wc84_tab) {  // reciprocals by table: see tpg_fst_wc84_tab_kernel
    const int64_t nch = ceil_div(m, lb_tab);
    const int nbt = (int)std::max<int64_t>(1, std::min<int64_t>(nch, (int64_t)nblocks));
    dim3 grid((unsigned)nbt, (unsigned)ypass);
    // many pairs: a thread takes a tile of populations (tpg_fst_wc84_tile_kernel); TPG_FST_TILES=0: 8 unrelated pairs (A/B)
    static const bool no_tiles = getenv("TPG_FST_TILES") && atoi(getenv("TPG_FST_TILES")) == 0;
    InBuf tb;
    int ntask = 0;
    const bool tiles = ppt == 8 && !no_tiles && G <= 64 && kmax <= FSTT_KMAX && lb_tab == FSTT_LB;
    if (tiles) {
      std::vector<int32_t> tasks;
      fst_wc84_tiles(p0, P, tasks);
      ntask = (int)(tasks.size() / FSTW_TASK_INTS);
      rc = tb.init(ctx, tasks.data(), sizeof(int32_t) * tasks.size());
    }
    if (rc == TPG_OK && ntask > 0) {
      const int nbt2 = (int)std::max<int64_t>(1, std::min<int64_t>(nch, (int64_t)nblocks));
      dim3 tgrid((unsigned)nbt2, (unsigned)ceil_div(ntask, 256));
      TPG_LAUNCH(ctx, "fst_wc84", tpg_fst_wc84_tile_kernel, tgrid, dim3(256), 0, src, m, G, kmax, tb.dev<int32_t>(), ntask, P,
                 d_part);
    } else if (rc == TPG_OK) {
#define FSTW_LAUNCH(PP, GSV)                                                                                                    \
  do {                                                                                                                          \
    (void)hipFuncSetAttribute((const void*)tpg_fst_wc84_tab_kernel<PP, GSV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_tab); \
    TPG_LAUNCH(ctx, "fst_wc84", (tpg_fst_wc84_tab_kernel<PP, GSV>), grid, dim3(256), sh_tab, src, m, G, lb_tab, kmax, pb.dev<int32_t>(), P, \
               d_part);                                                                                                         \
  } while (0)
    if (ppt == 8 && G <= 64) FSTW_LAUNCH(8, 64);
    else if (ppt == 8) FSTW_LAUNCH(8, 0);
    else if (G <= 64) FSTW_LAUNCH(1, 64);
    else FSTW_LAUNCH(1, 0);
#undef FSTW_LAUNCH
    }
    if (rc == TPG_OK)
      TPG_LAUNCH(ctx, "fst_reduce", tpg_fst_reduce_kernel, dim3((unsigned)ceil_div(P, 16)), dim3(256), 0, d_part, nbt, P,
                 ot.dev<double>(), osn.dev<double>(), osd.dev<double>());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { tpg_set_error("fst kernels: %s", hipGetErrorString(e)); rc = TPG_EHIP; }
  } else if (rc == TPG_OK) {
    dim3 grid((unsigned)nblocks, (unsigned)ypass);
#define FST_LAUNCH1(M, F, PP, NAME)                                                                              \
  do {                                                                                                           \
    (void)hipFuncSetAttribute((const void*)tpg_fst_kernel<M, F, PP>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                              (int)shmem);                                                                       \
    TPG_LAUNCH(ctx, NAME, (tpg_fst_kernel<M, F, PP>), grid, dim3(256), shmem, src, m, G, LB, pb.dev<int32_t>(), P, \
               by_locus, return_num_dem, d_part, oa.dev<double>(), ob.dev<double>());                            \
  } while (0)
#define FST_LAUNCH(M, F, NAME) do { if (ppt == 8) FST_LAUNCH1(M, F, 8, NAME); else FST_LAUNCH1(M, F, 1, NAME); } while (0)
    if (method == TPG_FST_HUDSON && fast) FST_LAUNCH(TPG_FST_HUDSON, true, "fst_hudson");  // same arithmetic, no per-locus stores
    else if (method == TPG_FST_HUDSON) FST_LAUNCH(TPG_FST_HUDSON, false, "fst_hudson");
    else if (method == TPG_FST_WC84 && fast) FST_LAUNCH(TPG_FST_WC84, true, "fst_wc84");
    else if (method == TPG_FST_WC84) FST_LAUNCH(TPG_FST_WC84, false, "fst_wc84");
    else FST_LAUNCH(TPG_FST_NEI87, false, "fst_nei87");
#undef FST_LAUNCH1
#undef FST_LAUNCH
    if (fst_tot || sum_num)
      TPG_LAUNCH(ctx, "fst_reduce", tpg_fst_reduce_kernel, dim3((unsigned)ceil_div(P, 16)), dim3(256), 0, d_part,
                 nblocks, P, ot.dev<double>(), osn.dev<double>(), osd.dev<double>());
    hipError_t e = hipGetLastError();  // no wait: d_part returns to the pool in stream order, commit() waits for host outputs
    if (e != hipSuccess) { tpg_set_error("fst kernels: %s", hipGetErrorString(e)); rc = TPG_EHIP; }
  }
  tpg_pfree(d_part);
  TPG_TRY(rc);
  if (fst_tot) TPG_TRY(ot.commit(ctx));
  if (sum_num) { TPG_TRY(osn.commit(ctx)); TPG_TRY(osd.commit(ctx)); }
  if (by_locus) TPG_TRY(oa.commit(ctx));
  if (return_num_dem) TPG_TRY(ob.commit(ctx));
  return TPG_OK;
}

extern "C" int tpg_pairwise_fst_loop(tpg_ctx* ctx, int method, const int32_t* pairs1, int P, int64_t m, int G,
                                     const double* n, const double* freq_alt, const double* freq_ref,
                                     const double* het_obs, int by_locus, int return_num_dem, double* fst_tot,
                                     double* out_a, double* out_b) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && n && freq_alt, TPG_EINVAL, "null argument");
  TPG_REQUIRE(m > 0 && G > 0, TPG_EINVAL, "empty input");
  TPG_REQUIRE(method == TPG_FST_HUDSON || het_obs, TPG_EINVAL, "het_obs is required for WC84 / Nei87");
  const size_t bytes = sizeof(double) * (size_t)m * (size_t)G;
  InBuf bn, bp, bh;
  TPG_TRY(bn.init(ctx, n, bytes));
  TPG_TRY(bp.init(ctx, freq_alt, bytes));
  if (het_obs) TPG_TRY(bh.init(ctx, het_obs, bytes));
  // freq_ref is 1 - freq_alt by construction (src/grouped_summaries_dip_pseudo_cpp.cpp:53) and the device code
  // recomputes it.  The reference's Hudson and Nei87 loops READ the caller's matrix
  // (src/pairwise_fst_hudson_loop.cpp:28-32, src/pairwise_fst_nei87_loop.cpp:66-72), so a matrix that is anything
  // else would silently give other numbers there: refuse it instead.
  if (freq_ref && method != TPG_FST_WC84) {
    InBuf bq;
    TPG_TRY(bq.init(ctx, freq_ref, bytes));
    int* d_bad = nullptr;
    TPG_HIP(tpg_pmalloc((void**)&d_bad, sizeof(int)));
    int bad = 0;
    hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(tpg_freq_ref_check_kernel, dim3(1024), dim3(256), 0, ctx->stream, bp.dev<double>(),
                         bq.dev<double>(), (int64_t)m * G, d_bad);
      e = tpg_fetch_small(ctx, &bad, d_bad, sizeof(int));
    }
    tpg_pfree(d_bad);
    TPG_HIP(e);
    TPG_REQUIRE(!bad, TPG_EINVAL, "freq_ref is not 1 - freq_alt: the device path recomputes it and would not match the reference");
  }
  FstSrc src{nullptr, 0, 0, 0, bn.dev<double>(), bp.dev<double>(), nullptr, het_obs ? bh.dev<double>() : nullptr};
  return run_fst(ctx, method, src, m, G, pairs1, P, by_locus, return_num_dem, fst_tot, out_a, out_b);
}

static int fused_fst(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups, const double* ploidy,
                     int method, const int32_t* pairs1, int P, int by_locus, int return_num_dem, double* fst_tot,
                     double* out_a, double* out_b, double* sum_num, double* sum_den) {
  TPG_REQUIRE(ctx && v && groupIds0, TPG_EINVAL, "null argument");
  TPG_REQUIRE(ngroups > 0, TPG_EINVAL, "ngroups must be positive");
  int has_hap = 0;
  if (ploidy)
    for (int64_t i = 0; i < v->n; i++) {
      TPG_REQUIRE(ploidy[i] == 1.0 || ploidy[i] == 2.0, TPG_EUNSUPPORTED, "ploidy[%lld] = %g unsupported",
                  (long long)i, ploidy[i]);
      if (ploidy[i] == 1.0) has_hap = 1;
    }
  // R/pairwise_pop_fst.R:110-115: pseudohaploids only with Hudson
  TPG_REQUIRE(!has_hap || method == TPG_FST_HUDSON, TPG_EINVAL,
              "only method = Hudson is valid when the data include pseudohaploids");
  std::vector<int32_t> cls((size_t)v->n);
  for (int64_t i = 0; i < v->n; i++) {
    TPG_REQUIRE(groupIds0[i] >= 0 && groupIds0[i] < ngroups, TPG_EINVAL, "groupIds[%lld] = %d out of [0,%d)",
                (long long)i, groupIds0[i], ngroups);
    cls[(size_t)i] = has_hap ? 2 * groupIds0[i] + (ploidy[i] == 1.0 ? 1 : 0) : groupIds0[i];
  }
  GroupedCounts gc;
  TPG_TRY(tpg_grouped_counts(ctx, v, cls.data(), ngroups * (has_hap ? 2 : 1), &gc));
  FstSrc src{gc.cnt, gc.Mpad, gc.Cpad, has_hap, nullptr, nullptr, nullptr, nullptr};
  // the valid alleles of a pair of populations never exceed four times the largest group
  std::vector<int64_t> gsize((size_t)ngroups, 0);
  for (int64_t i = 0; i < v->n; i++) gsize[(size_t)groupIds0[i]]++;
  const int64_t kmax = 4 * *std::max_element(gsize.begin(), gsize.end());
  return run_fst(ctx, method, src, v->m, ngroups, pairs1, P, by_locus, return_num_dem, fst_tot, out_a, out_b, sum_num,
                 sum_den, kmax < (1 << 20) ? (int)kmax : 0);
}

extern "C" int tpg_pairwise_pop_fst(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups,
                                    const double* ploidy, int method, const int32_t* pairs1, int P, int by_locus,
                                    int return_num_dem, double* fst_tot, double* out_a, double* out_b) {
  TpgEnter _enter(ctx);
  return fused_fst(ctx, v, groupIds0, ngroups, ploidy, method, pairs1, P, by_locus, return_num_dem, fst_tot, out_a,
                   out_b, nullptr, nullptr);
}

extern "C" int tpg_pairwise_pop_fst_sums(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups,
                                         const double* ploidy, int method, const int32_t* pairs1, int P,
                                         double* sum_num, double* sum_den) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(sum_num && sum_den, TPG_EINVAL, "null output");
  return fused_fst(ctx, v, groupIds0, ngroups, ploidy, method, pairs1, P, 0, 0, nullptr, nullptr, nullptr, sum_num,
                   sum_den);
}

// ---------------------------------------------------------------------------
// Population branch statistic from by-locus (or by-window) pairwise Fst: pbs_one_triplet of R/nwise_pop_pbs.R:118-156.
// For triplet t with Fst columns (c12, c13, c23): six output columns {pbs_1, pbs_2, pbs_3, pbsn1_1, pbsn1_2, pbsn1_3}.
__global__ void tpg_pbs_kernel(const double* __restrict__ fst, int64_t m, const int32_t* __restrict__ trip, int ntrip,
                               double* __restrict__ out) {
  const int64_t total = m * ntrip;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = idx % m;
    const int t = (int)(idx / m);
    const double f12 = fst[j + (int64_t)trip[3 * t] * m], f13 = fst[j + (int64_t)trip[3 * t + 1] * m],
                 f23 = fst[j + (int64_t)trip[3 * t + 2] * m];
    const double t12 = -log(1 - f12), t13 = -log(1 - f13), t23 = -log(1 - f23);  // :138-140
    const double p1 = (t12 + t13 - t23) / 2, p2 = (t12 + t23 - t13) / 2, p3 = (t13 + t23 - t12) / 2;
    double* o = out + (int64_t)t * 6 * m + j;
    o[0] = p1; o[m] = p2; o[2 * m] = p3;
    o[3 * m] = p1 / (1 + p1 + p2 + p3);  // :147-149
    o[4 * m] = p2 / (1 + p1 + p2 + p3);
    o[5 * m] = p3 / (1 + p1 + p2 + p3);
  }
}

extern "C" int tpg_pbs_from_fst(tpg_ctx* ctx, const double* fst, int64_t m, int P, const int32_t* trip_cols0, int ntrip,
                                double* out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && fst && trip_cols0 && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(m > 0 && P > 0 && ntrip > 0, TPG_EINVAL, "bad sizes");
  for (int k = 0; k < 3 * ntrip; k++)
    TPG_REQUIRE(trip_cols0[k] >= 0 && trip_cols0[k] < P, TPG_EINVAL, "triplet column %d out of [0,%d)", trip_cols0[k], P);
  InBuf ifst, itr;
  TPG_TRY(ifst.init(ctx, fst, sizeof(double) * (size_t)m * (size_t)P));
  TPG_TRY(itr.init(ctx, trip_cols0, sizeof(int32_t) * 3 * (size_t)ntrip));
  OutBuf o;
  TPG_TRY(o.init(out, sizeof(double) * (size_t)m * 6 * (size_t)ntrip));
  TPG_LAUNCH(ctx, "pbs", tpg_pbs_kernel, dim3(2048), dim3(256), 0, ifst.dev<double>(), m, itr.dev<int32_t>(), ntrip,
             o.dev<double>());
  TPG_CHECK_LAUNCH();
  TPG_HIP(hipStreamSynchronize(ctx->stream));
  return o.commit(ctx);
}

