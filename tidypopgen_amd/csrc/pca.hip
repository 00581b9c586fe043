// pca.hip -- gt_pca_partialSVD on the device: center/scale, Gram matrix, top-k eigenpairs,
// loadings, projections and the squared Frobenius norm.
//
// Replaces the arithmetic behind R/gt_pca_partialSVD.R:82-89 (bigstatsr::big_SVD with
// bigsnpr::snp_scaleBinom -- third-party, recalled: center_j = sum_i x_ij / n, p = center/2,
// scale_j = sqrt(2 p (1-p)), K = sum_j z_j z_j', eigen(K), d = sqrt(lambda), v = Z'u/d),
// R/square_frobenius.R:19-35 and fbm256_prod_and_rowSumsSq (src/fbm_prod_and_rowSumSq.cpp:10-47).
//
// Gram matrix.  z_ij = (g_ij - c_j)/s_j with g in {0,1,2} (no missing values), so
//     K_ik = sum_j w_j g_ij g_kj  -  r_i  -  r_k  +  C,      w_j = 1/s_j^2,
//     r_i  = sum_j w_j c_j g_ij,   C = sum_j w_j c_j^2.
// S' = sum_j w_j g_i g_k' comes from gramcls.hip (weight classes on the FP4 matrix cores) whenever the weights take few
// enough distinct values for that to pay -- always so under the binomial scaling on a long panel -- and from the
// digit-split int8 MFMA kernel of this file otherwise.  For the latter the weights are rounded to fixed point, W_j = round(w_j 2^F), and written in balanced base-128
// digits W_j = sum_t D_t[j] 128^t, D_t in [-64, 63].  Then sum_j W_j g_ij g_kj is a sum of T exact
// int8 x int8 -> int32 MFMA contractions with A = D_t[j] * g_ij (|.| <= 128) and B = g_kj.  The only
// approximation is the rounding of w_j (relative error <= 2^-(F+1)/w_min <= 2^-(F+2)): K-hat is the
// exact Gram matrix of data whose per-locus scale is perturbed by that relative amount, so every
// eigenvalue moves by at most that relative amount.  F is chosen as large as 4 digits allow
// (F = 22 for w_max < 32, i.e. relative 6e-8); more digits are added when rare alleles need them.
// r and C are FP64 sweeps that use the same rounded weights, which keeps K-hat exactly symmetric PSD
// up to FP64 rounding.
//
// A-side fragment of digit t: one v_perm_b32 per register selects, per byte, D_t (genotype 1),
// 2 D_t (genotype 2) or 0 (genotype 0) with a selector computed once per register from the codes.
//
// Eigen step: Chebyshev-filtered subspace iteration (Zhou & Saad) on the N x N matrix in HBM; all
// N-sized work (K Q products, projections, residuals) runs in FP64 kernels here, only b x b
// (b = 2k + 12) factorizations run on the host.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>

#include "common.h"
#include <type_traits>
#include "devfrag.h"

#define MFMA_I8(a, b, c) __builtin_amdgcn_mfma_i32_32x32x32_i8((a), (b), (c), 0, 0, 0)
#define PCA_NAN __longlong_as_double(0x7FF8000000000000ll)

// ---------------------------------------------------------------------------
// center / scale from genotype counts; flags[0] = missing value met, flags[1] = zero scale met
__global__ void tpg_pca_center_scale_kernel(const int4* __restrict__ counts, int64_t m, int64_t n,
                                            double* __restrict__ center, double* __restrict__ scale,
                                            int* __restrict__ flags) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = counts[j];
    if (c.w != 0) flags[0] = 1;
    const double mean = (double)(c.y + 2 * c.z) / (double)n;  // big_colstats()$sum / n
    const double p = mean / 2;
    const double sd = sqrt(2 * p * (1 - p));
    if (!(sd > 0)) flags[1] = 1;
    center[j] = mean;
    scale[j] = sd;
  }
}

// R/square_frobenius.R:32-34 from the counts: sum_j ((n-1) var_j + n (mean_j - center_j)^2) / scale_j^2,
// var_j the sample variance (bigstatsr::big_colstats, recalled).  Per-block partial sums.
__global__ __launch_bounds__(256) void tpg_pca_frobenius_kernel(const int4* __restrict__ counts, int64_t m, int64_t n,
                                                                const double* __restrict__ center,
                                                                const double* __restrict__ scale,
                                                                double* __restrict__ part) {
  __shared__ double sh[256];
  double acc = 0;
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = counts[j];
    const double s = (double)(c.y + 2 * c.z), ss = (double)(c.y + 4 * c.z);
    const double nn = (double)n;
    const double var = (ss - s * s / nn) / (nn - 1);
    const double dm = s / nn - center[j];
    acc += ((nn - 1) * var + nn * dm * dm) / (scale[j] * scale[j]);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

static int pca_counts_center_scale(tpg_ctx* ctx, const tpg_view* v, int32_t* d_counts, double* d_center,
                                   double* d_scale) {
  TPG_TRY(tpg_launch_loci_counts(ctx, v, d_counts));
  int* d_flags = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_flags, 2 * sizeof(int)));
  hipError_t e = hipMemsetAsync(d_flags, 0, 2 * sizeof(int), ctx->stream);
  int flags[2] = {0, 0};
  if (e == hipSuccess) {
    TPG_LAUNCH(ctx, "pca_center_scale", tpg_pca_center_scale_kernel, dim3(1024), dim3(256), 0, (const int4*)d_counts,
               v->m, v->n, d_center, d_scale, d_flags);
    e = tpg_fetch_small(ctx, flags, d_flags, sizeof(flags));
  }
  tpg_pfree(d_flags);
  if (e != hipSuccess) { tpg_set_error("pca center/scale: %s", hipGetErrorString(e)); return TPG_EHIP; }
  // bigstatsr::big_SVD stops on missing values and on a zero scale
  TPG_REQUIRE(!flags[0], TPG_ENUMERIC, "You can't have missing values in 'X'.");
  TPG_REQUIRE(!flags[1], TPG_ENUMERIC, "Some variables have a zero scaling; remove them before attempting to scale variables.");
  return TPG_OK;
}

extern "C" int tpg_pca_center_scale(tpg_ctx* ctx, const tpg_view* v, double* center, double* scale) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && center && scale, TPG_EINVAL, "null argument");
  OutBuf oc, os;
  TPG_TRY(oc.init(center, sizeof(double) * (size_t)v->m));
  TPG_TRY(os.init(scale, sizeof(double) * (size_t)v->m));
  int32_t* d_counts = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_counts, sizeof(int32_t) * 4 * (size_t)v->m));
  int rc = pca_counts_center_scale(ctx, v, d_counts, oc.dev<double>(), os.dev<double>());
  tpg_pfree(d_counts);
  TPG_TRY(rc);
  TPG_TRY(oc.commit(ctx));
  return os.commit(ctx);
}

static int frobenius_from_counts(tpg_ctx* ctx, const tpg_view* v, const int32_t* d_counts, const double* d_center,
                                 const double* d_scale, double* out_host) {
  const int NB = 512;
  double* d_part = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_part, sizeof(double) * NB));
  TPG_LAUNCH(ctx, "pca_frobenius", tpg_pca_frobenius_kernel, dim3(NB), dim3(256), 0, (const int4*)d_counts, v->m, v->n,
             d_center, d_scale, d_part);
  std::vector<double> hp(NB);
  hipError_t e = tpg_fetch_small(ctx, hp.data(), d_part, sizeof(double) * NB);
  tpg_pfree(d_part);
  if (e != hipSuccess) { tpg_set_error("frobenius: %s", hipGetErrorString(e)); return TPG_EHIP; }
  long double s = 0;
  for (int b = 0; b < NB; b++) s += hp[b];
  *out_host = (double)s;
  return TPG_OK;
}

extern "C" int tpg_square_frobenius(tpg_ctx* ctx, const tpg_view* v, const double* center, const double* scale,
                                    double* out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && center && scale && out, TPG_EINVAL, "null argument");
  InBuf ic, is;
  TPG_TRY(ic.init(ctx, center, sizeof(double) * (size_t)v->m));
  TPG_TRY(is.init(ctx, scale, sizeof(double) * (size_t)v->m));
  int32_t* d_counts = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_counts, sizeof(int32_t) * 4 * (size_t)v->m));
  int rc = tpg_launch_loci_counts(ctx, v, d_counts);
  if (rc == TPG_OK) rc = frobenius_from_counts(ctx, v, d_counts, ic.dev<double>(), is.dev<double>(), out);
  tpg_pfree(d_counts);
  return rc;
}

// ---------------------------------------------------------------------------
// Skinny FP64 sweeps over a fragment layout (T: rows = individuals, columns = loci;
// L: rows = loci, columns = individuals):
//     out[row][k] = sum_col z(row,col) * Tab[col][k],   rss[row] = sum_col z(row,col)^2
// with z = (g - center) * inv_scale for typed genotypes and 0 for missing ones.
//   COLSCALE: center / inv_scale belong to the column (T layout: XV = Z V, src/fbm_prod_and_rowSumSq.cpp:30-44)
//   ROWSCALE: they belong to the row (L layout: Z'U of the big_SVD second sweep)
//   RAW:      z = g, no centering (the r_i sums of the Gram correction)
// One wave per 32-row tile; the 128-column slice of Tab for the current block is staged in LDS and
// shared by the workgroup's 4 waves.  KC (<= 8) output columns per pass.
enum { SW_COLSCALE = 0, SW_ROWSCALE = 1, SW_RAW = 2, SW_VALID = 3 };  // SW_VALID: z = 1 for a typed genotype (0 if missing)

template <int KC, int MODE>
__global__ __launch_bounds__(256) void tpg_sweep_kernel(const uint4* __restrict__ P, int64_t nrowtiles, int64_t nblocks,
                                                        int S, int64_t nrows, int64_t ncols,
                                                        const double* __restrict__ center,
                                                        const double* __restrict__ inv_scale,
                                                        const double* __restrict__ Tab, int64_t ldtab, int kc,
                                                        double* __restrict__ part, double* __restrict__ rss_part) {
  __shared__ double tab[128][KC];
  __shared__ double cs[128][2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t rt = (int64_t)blockIdx.x * 4 + wv;
  const int split = blockIdx.y;
  const int64_t b0 = nblocks * split / S, b1 = nblocks * (split + 1) / S;
  const int r = lane & 31, h = lane >> 5;
  const int64_t row = rt * 32 + r;
  double acc[KC];
#pragma unroll
  for (int k = 0; k < KC; k++) acc[k] = 0;
  double rss = 0;
  double rc = 0, ris = 1;
  if (MODE == SW_ROWSCALE && row < nrows) { rc = center[row]; ris = inv_scale[row]; }
  const uint4* p = P + (rt * nblocks) * 64 + lane;
  for (int64_t b = b0; b < b1; b++) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 128 * KC; idx += 256) {
      const int cl = idx / KC, k = idx % KC;
      const int64_t col = b * 128 + cl;
      tab[cl][k] = (col < ncols && k < kc) ? Tab[col + (int64_t)k * ldtab] : 0.0;
    }
    if (MODE == SW_COLSCALE && threadIdx.x < 128) {
      const int64_t col = b * 128 + threadIdx.x;
      cs[threadIdx.x][0] = col < ncols ? center[col] : 0.0;
      cs[threadIdx.x][1] = col < ncols ? inv_scale[col] : 0.0;
    }
    __syncthreads();
    if (rt < nrowtiles) {
      const uint4 a = p[b * 64];
      const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int s = 0; s < 4; s++) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int code = (w[s] >> tpg_elem_shift(e)) & 3;
          const int cl = 32 * s + 16 * h + e;
          double z;
          if (MODE == SW_COLSCALE) z = ((double)code - cs[cl][0]) * cs[cl][1];
          else if (MODE == SW_ROWSCALE) z = ((double)code - rc) * ris;
          else if (MODE == SW_VALID) z = 1.0;
          else z = (double)code;
          if (code == 3) z = 0;
          rss += z * z;
#pragma unroll
          for (int k = 0; k < KC; k++) acc[k] += z * tab[cl][k];
        }
      }
    }
  }
  // the two halves of a row
#pragma unroll
  for (int k = 0; k < KC; k++) acc[k] += __shfl_xor(acc[k], 32);
  rss += __shfl_xor(rss, 32);
  if (rt < nrowtiles && lane < 32) {
    const int64_t rows_pad = nrowtiles * 32;
#pragma unroll
    for (int k = 0; k < KC; k++) part[((int64_t)split * KC + k) * rows_pad + row] = acc[k];
    if (rss_part) rss_part[(int64_t)split * rows_pad + row] = rss;
  }
}

// out[row + (k0+k)*ldout] = scale_k * sum_split part[split][k][row]
__global__ void tpg_sweep_reduce_kernel(const double* __restrict__ part, const double* __restrict__ rss_part, int S,
                                        int KC, int kc, int64_t rows_pad, int64_t nrows, double* __restrict__ out,
                                        int64_t ldout, int k0, const double* __restrict__ col_div,
                                        double* __restrict__ rss_out) {
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < nrows * (kc + 1);
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx % nrows;
    const int k = (int)(idx / nrows);
    if (k < kc) {
      double s = 0;
      for (int sp = 0; sp < S; sp++) s += part[((int64_t)sp * KC + k) * rows_pad + row];
      if (col_div) s = s / col_div[k0 + k];
      out[row + (int64_t)(k0 + k) * ldout] = s;
    } else if (rss_out && rss_part) {
      double s = 0;
      for (int sp = 0; sp < S; sp++) s += rss_part[(int64_t)sp * rows_pad + row];
      rss_out[row] = s;
    }
  }
}

// out (nrows x K, column-major, ld = nrows) ; Tab is ncols x K column-major (ld = ldtab)
static int run_sweep(tpg_ctx* ctx, int mode, const uint4* P, int64_t nrowtiles, int64_t nblocks, int64_t nrows,
                     int64_t ncols, const double* d_center, const double* d_inv_scale, const double* d_Tab,
                     int64_t ldtab, int K, double* d_out, const double* d_col_div, double* d_rss) {
  const int KC = K <= 4 ? 4 : 20;
  const int64_t rows_pad = nrowtiles * 32;
  // split the column range so that the grid fills the chip (rows alone may be few: N/32 tiles)
  int64_t S = 1;
  const int64_t row_blocks = ceil_div(nrowtiles, 4);
  while (row_blocks * S < 4 * ctx->num_cu && S * 2 <= nblocks && S < 64) S *= 2;
  double *d_part = nullptr, *d_rsp = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_part, sizeof(double) * (size_t)S * KC * (size_t)rows_pad));
  hipError_t e = tpg_pmalloc((void**)&d_rsp, sizeof(double) * (size_t)S * (size_t)rows_pad);
  if (e != hipSuccess) { tpg_pfree(d_part); tpg_set_error("hipMalloc sweep: %s", hipGetErrorString(e)); return TPG_EHIP; }
  dim3 grid((unsigned)row_blocks, (unsigned)S);
  for (int k0 = 0; k0 < K; k0 += KC) {
    const int kc = K - k0 < KC ? K - k0 : KC;
    const double* tabp = d_Tab + (int64_t)k0 * ldtab;
    double* rsp = (k0 == 0 && d_rss) ? d_rsp : nullptr;
#define SWEEP_LAUNCH(KCV, MODEV, NAME)                                                                              \
  TPG_LAUNCH(ctx, NAME, (tpg_sweep_kernel<KCV, MODEV>), grid, dim3(256), 0, P, nrowtiles, nblocks, (int)S, nrows,   \
             ncols, d_center, d_inv_scale, tabp, ldtab, kc, d_part, rsp)
    if (KC == 4) {
      if (mode == SW_COLSCALE) SWEEP_LAUNCH(4, SW_COLSCALE, "sweep_colscale");
      else if (mode == SW_ROWSCALE) SWEEP_LAUNCH(4, SW_ROWSCALE, "sweep_rowscale");
      else if (mode == SW_VALID) SWEEP_LAUNCH(4, SW_VALID, "sweep_valid");
      else SWEEP_LAUNCH(4, SW_RAW, "sweep_raw");
    } else {
      if (mode == SW_COLSCALE) SWEEP_LAUNCH(20, SW_COLSCALE, "sweep_colscale");
      else if (mode == SW_ROWSCALE) SWEEP_LAUNCH(20, SW_ROWSCALE, "sweep_rowscale");
      else if (mode == SW_VALID) SWEEP_LAUNCH(20, SW_VALID, "sweep_valid");
      else SWEEP_LAUNCH(20, SW_RAW, "sweep_raw");
    }
#undef SWEEP_LAUNCH
    TPG_LAUNCH(ctx, "sweep_reduce", tpg_sweep_reduce_kernel, dim3(512), dim3(256), 0, d_part, rsp, (int)S, KC, kc,
               rows_pad, nrows, d_out, nrows, k0, d_col_div, rsp ? d_rss : nullptr);
  }
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  tpg_pfree(d_part);
  tpg_pfree(d_rsp);
  if (e != hipSuccess) { tpg_set_error("sweep: %s", hipGetErrorString(e)); return TPG_EHIP; }
  return TPG_OK;
}

__global__ void tpg_inv_kernel(const double* __restrict__ x, int64_t n, double* __restrict__ y) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = 1.0 / x[i];
}

extern "C" int tpg_fbm256_prod_and_rowSumsSq(tpg_ctx* ctx, const tpg_view* v, const double* center,
                                             const double* scale, const double* V, int K, double* XV, double* rss) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && center && scale && V && XV && rss, TPG_EINVAL, "null argument");
  TPG_REQUIRE(K > 0, TPG_EINVAL, "V has no columns");
  InBuf ic, is, iv;
  TPG_TRY(ic.init(ctx, center, sizeof(double) * (size_t)v->m));
  TPG_TRY(is.init(ctx, scale, sizeof(double) * (size_t)v->m));
  TPG_TRY(iv.init(ctx, V, sizeof(double) * (size_t)v->m * (size_t)K));
  OutBuf oxv, orss;
  TPG_TRY(oxv.init(XV, sizeof(double) * (size_t)v->n * (size_t)K));
  TPG_TRY(orss.init(rss, sizeof(double) * (size_t)v->n));
  double* d_inv = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_inv, sizeof(double) * (size_t)v->m));
  TPG_LAUNCH(ctx, "inv_scale", tpg_inv_kernel, dim3(1024), dim3(256), 0, is.dev<double>(), v->m, d_inv);
  TPG_TRY(tpg_view_need_T(ctx, v));
  int rc = run_sweep(ctx, SW_COLSCALE, v->T, v->Q * 4, v->KG, v->n, v->m, ic.dev<double>(), d_inv, iv.dev<double>(),
                     v->m, K, oxv.dev<double>(), nullptr, orss.dev<double>());
  tpg_pfree(d_inv);
  TPG_TRY(rc);
  TPG_TRY(oxv.commit(ctx));
  return orss.commit(ctx);
}

// out[i][k] = sum over the loci j where individual i is typed of Tab[j][k]: the per-individual masked sums behind
// predict(project_method = "least_squares") (R/predict_gt_pca.R:221-228), where every individual solves its own
// normal equations over its non-missing loci: crossprod(v_sub) = sum_j typed v_j v_j' is this sweep with
// Tab = the pairwise products of the columns of v.
extern "C" int tpg_fbm256_valid_prod(tpg_ctx* ctx, const tpg_view* v, const double* Tab, int K, double* out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && Tab && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(K > 0, TPG_EINVAL, "Tab has no columns");
  InBuf it;
  TPG_TRY(it.init(ctx, Tab, sizeof(double) * (size_t)v->m * (size_t)K));
  OutBuf oo;
  TPG_TRY(oo.init(out, sizeof(double) * (size_t)v->n * (size_t)K));
  TPG_TRY(tpg_view_need_T(ctx, v));
  TPG_TRY(run_sweep(ctx, SW_VALID, v->T, v->Q * 4, v->KG, v->n, v->m, nullptr, nullptr, it.dev<double>(), v->m, K,
                    oo.dev<double>(), nullptr, nullptr));
  return oo.commit(ctx);
}

// ---------------------------------------------------------------------------
// Gram matrix: weight digits
//
// DG layout (per pass of <= 4 digits, TD digits in the pass): K group kg owns TD * 256 contiguous bytes:
// for K step s, lane half h, digit t: 8 dwords = D (4 dwords) then 2D (4 dwords); byte b of dword k
// belongs to locus 128 kg + 32 s + 16 h + 4 k + b.  dword index = (((kg*4 + s)*2 + h)*TD + t)*8 + which*4 + k.
__global__ void tpg_pca_digits_kernel(const double* __restrict__ scale, const double* __restrict__ center, int64_t m,
                                      int64_t KG, int F, int T, uint32_t* __restrict__ DG, double* __restrict__ what,
                                      double* __restrict__ wc) {
  const int64_t total = KG * 4 * 2 * 4;  // (kg, s, h, k)
  const int npass = (T + 3) / 4;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx & 3), h = (int)((idx >> 2) & 1), s = (int)((idx >> 3) & 3);
    const int64_t kg = idx >> 5;
    uint32_t d[8], d2[8];
    for (int t = 0; t < 8; t++) { d[t] = 0; d2[t] = 0; }
    for (int b = 0; b < 4; b++) {
      const int64_t j = kg * 128 + 32 * s + 16 * h + 4 * k + b;
      if (j >= m) continue;
      const double w = 1.0 / (scale[j] * scale[j]);
      long long W = llrint(ldexp(w, F));
      const double wh = ldexp((double)W, -F);
      what[j] = wh;
      wc[j] = wh * center[j];
      for (int t = 0; t < T; t++) {
        long long dig = W & 127;
        if (dig >= 64) dig -= 128;
        W = (W - dig) >> 7;
        d[t] |= (uint32_t)((int)dig & 0xFF) << (8 * b);
        d2[t] |= (uint32_t)((int)(2 * dig) & 0xFF) << (8 * b);
      }
    }
    int64_t pass_base = 0;  // dwords before this pass's table
    for (int p = 0; p < npass; p++) {
      const int td = T - 4 * p < 4 ? T - 4 * p : 4;
      const int64_t base = pass_base + (((kg * 4 + s) * 2 + h) * td) * 8;
      for (int t = 0; t < td; t++) {
        DG[base + t * 8 + k] = d[4 * p + t];
        DG[base + t * 8 + 4 + k] = d2[4 * p + t];
      }
      pass_base += KG * 4 * 2 * td * 8;
    }
  }
}

// unit (ia, jb): A row tile ia (32 individuals, weighted digit planes) x B super-tile jb (PCA_TB row tiles,
// plain dosage), jb >= ia / PCA_TB.  Combined int64 slab per unit: [tb][reg][lane].
#define PCA_TB 4  // B row tiles per unit: 32 x 128 wave tile, 16 accumulator tiles (256 AGPRs), 1 wave per SIMD
#define PCA_SLAB_INTS (PCA_TB * 16 * 64)


// One wave per SIMD, 16 accumulator tiles in AGPRs.  The operands of K step s+1 (one weighted A fragment per
// digit, PCA_TB dosage fragments) are decoded into a second register set while the TD * PCA_TB MFMAs of step s
// issue (sched_group_barrier: one MFMA, then five VALU), because an MFMA that reads registers written by VALU
// instructions just before it stalls (tools/ubench_mfma_dep.hip: 47.6 / 41.6 / 36.4 cycles at distance 0 / 1 / 2).
// The digit table of a 128-locus group (TD * 256 B) goes through wave-private, double-buffered LDS.
#define SGB_MFMA 0x008
#define SGB_VALU 0x002
#define SGB_DSR 0x100
#define SGB_VMEMR 0x020
template <int TD>  // digits handled by this pass (<= 4); DG holds exactly these TD digits per locus
__global__ __launch_bounds__(256, 1) void tpg_pca_gram_kernel(const uint4* __restrict__ Tl, int64_t KG,
                                                              int64_t kg_begin, int64_t kg_end,
                                                              const uint4* __restrict__ DG, int t0,
                                                              const int2* __restrict__ order, int nun, int S,
                                                              long long* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) uint4 dgs[4][2][TD * 16];
  constexpr int TB = PCA_TB;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int h = lane >> 5;
  const int64_t kgs = kg_end - kg_begin;
  // Work distribution.  `order` (host-built, pca_gram_device) lists the units (ia, jb) -- A row tile ia against the
  // B super-tile jb -- in PATCH order: four consecutive entries are the A row tiles 4a .. 4a+3 against the same B
  // super-tile (the four waves of a workgroup fetch the B stream into the CU once), and a run of 128 consecutive
  // entries is ~16 A tiles x 8 B super-tiles.  Workgroups are dispatched round-robin over the 8 XCDs
  // (blockIdx % 8), each XCD with its own L2: in every round XCD x takes such a run for its 128 waves (same K
  // range), ~16 A and ~32 B tiles per 128 loci instead of up to ~190 distinct ones with a plain strided
  // assignment -- the re-reads then hit that XCD's L2.  The slab of a unit is its position in the table.
  const int xcd = blockIdx.x & 7, cidx = blockIdx.x >> 3, cpx = gridDim.x >> 3;
  for (int64_t round = 0;; round++) {
    const int64_t un = ((round * 8 + xcd) * cpx + cidx) * 4 + wv;
    if (un >= (int64_t)nun * S) break;
    const int ks = (int)(un / nun);
    const int64_t u0 = un % nun;
    const int2 ijb = order[u0];
    const int ia = ijb.x, jb = ijb.y;
    const int64_t k0 = kg_begin + (kgs * ks) / S, k1 = kg_begin + (kgs * (ks + 1)) / S;

    const uint4* pa = Tl + ((int64_t)ia * KG) * 64 + lane;
    const uint4* pb[TB];
#pragma unroll
    for (int tb = 0; tb < TB; tb++) pb[tb] = Tl + ((int64_t)(TB * jb + tb) * KG) * 64 + lane;

    v16i acc[TD][TB];
#pragma unroll
    for (int t = 0; t < TD; t++)
#pragma unroll
      for (int tb = 0; tb < TB; tb++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[t][tb][q] = 0;

    if (k0 < k1) {
      const int64_t kl = k1 - 1;
      const bool dlane = lane < TD * 16;  // lanes that carry a piece of the digit table
      // decode of one K step: word wA of the A tile, words wB[tb] of the B tiles, digits from LDS buffer `dbuf`, step s
      v4i PA[2][TD], PB[2][TB];
      uint4 d1[TD], d2[TD];  // digits of the step being decoded
      // first half of a decode: issue all digit reads of the step, then the B side (needs no digits: its ~44
      // VALU instructions and the MFMAs interleaved with them cover the LDS latency)
      auto decode_B = [&](int set, const uint32_t* wB, int dbuf, int sidx) {
        const uint4* dl = &dgs[wv][dbuf][((sidx * 2 + h) * TD) * 2];
#pragma unroll
        for (int t = 0; t < TD; t++) { d1[t] = dl[t * 2]; d2[t] = dl[t * 2 + 1]; }
#pragma unroll
        for (int tb = 0; tb < TB; tb++)
#pragma unroll
          // the view has no missing value (both callers check): codes 0 / 1 / 2 ARE the dosage bytes.  Code 3 only
          // pads rows past n (their columns of K are never read) and loci past m (zero digits on the A side).
          for (int k = 0; k < 4; k++) PB[set][tb][k] = (int)tpg_codes(wB[tb], k);
      };
      // second half: selectors from the A word, then one v_perm per digit and register
      auto decode_A = [&](int set, uint32_t wA) {
        uint32_t sel[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t c = tpg_codes(wA, k);
          const uint32_t base = (uint32_t)tpg_lut(0x0C04000Cu, c);   // genotype 1 -> D (S1), 2 -> 2D (S0), else 0
          const uint32_t mv = (uint32_t)tpg_lut(0x00FFFF00u, c);
          sel[k] = (0x03020100u & mv) | base;
        }
#pragma unroll
        for (int t = 0; t < TD; t++) {
          PA[set][t][0] = (int)__builtin_amdgcn_perm(d2[t].x, d1[t].x, sel[0]);
          PA[set][t][1] = (int)__builtin_amdgcn_perm(d2[t].y, d1[t].y, sel[1]);
          PA[set][t][2] = (int)__builtin_amdgcn_perm(d2[t].z, d1[t].z, sel[2]);
          PA[set][t][3] = (int)__builtin_amdgcn_perm(d2[t].w, d1[t].w, sel[3]);
        }
      };
      auto lds_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      };

      // three rotating register slots (current group, next, the one being loaded): the K loop is unrolled by
      // three so that no slot is ever copied -- a copy at the end of a group would wait for loads issued at its
      // start, i.e. cut the prefetch distance from two groups to one
      uint4 RA[3], RB[3][TB], RD[3];
      RD[0] = RD[1] = RD[2] = make_uint4(0, 0, 0, 0);
      RA[0] = pa[k0 * 64];
#pragma unroll
      for (int tb = 0; tb < TB; tb++) RB[0][tb] = pb[tb][k0 * 64];
      if (dlane) dgs[wv][0][lane] = DG[k0 * (TD * 16) + lane];
      const int64_t i1 = k0 + 1 < k1 ? k0 + 1 : kl;
      if (dlane) RD[1] = DG[i1 * (TD * 16) + lane];
      RA[1] = pa[i1 * 64];
#pragma unroll
      for (int tb = 0; tb < TB; tb++) RB[1][tb] = pb[tb][i1 * 64];
      lds_sync();
      {
        const uint32_t wB0[TB] = {RB[0][0].x, RB[0][1].x, RB[0][2].x, RB[0][3].x};
        decode_B(0, wB0, 0, 0);
        decode_A(0, RA[0].x);
      }
      int buf = 0;
      auto group = [&](auto Cc, auto Nn, auto Mm, int64_t kg) {
        constexpr int C = decltype(Cc)::value, N = decltype(Nn)::value, M = decltype(Mm)::value;
        // digits of the next group into the other LDS buffer (its last readers were this wave's previous group)
        if (dlane) dgs[wv][buf ^ 1][lane] = RD[N];
        lds_sync();
        // the loads of the group after next are spread over steps 0..2 (at most two back to back): six in a row
        // hold this wave's instruction issue long enough for the MFMA pipe to drain (one wave per SIMD)
        const int64_t i2 = kg + 2 < k1 ? kg + 2 : kl;
        // groups past k1 (the loop runs in triples): all-missing A words, i.e. zero planes
        const uint32_t dead = kg < k1 ? 0u : ~0u, dead1 = kg + 1 < k1 ? 0u : ~0u;
        const uint32_t wA[5] = {RA[C].x | dead, RA[C].y | dead, RA[C].z | dead, RA[C].w | dead, RA[N].x | dead1};
        uint32_t wB[5][TB];
#pragma unroll
        for (int tb = 0; tb < TB; tb++) {
          wB[0][tb] = RB[C][tb].x; wB[1][tb] = RB[C][tb].y; wB[2][tb] = RB[C][tb].z; wB[3][tb] = RB[C][tb].w;
          wB[4][tb] = RB[N][tb].x;
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
          const int cur = s & 1, nx = cur ^ 1;
          // next step's operands: steps 1..3 of this group, or step 0 of the next group (other digit buffer).
          // Two scheduling regions so that the digit reads are long done when the A-side perms need them.
          constexpr int NQ = TD * TB, Q1 = NQ / 2 > 0 ? NQ / 2 : 1;  // half the MFMAs per region: ~3.5 and ~4.4 VALU per MFMA
          if (s == 0) {
            if (dlane) RD[M] = DG[i2 * (TD * 16) + lane];
            RA[M] = pa[i2 * 64];
          } else if (s <= (TB + 1) / 2) {
#pragma unroll
            for (int tb = 2 * (s - 1); tb < 2 * s && tb < TB; tb++) RB[M][tb] = pb[tb][i2 * 64];
          }
          decode_B(nx, wB[s + 1], s < 3 ? buf : (buf ^ 1), s < 3 ? s + 1 : 0);
#pragma unroll
          for (int q = 0; q < Q1; q++) acc[q / TB][q % TB] = MFMA_I8(PA[cur][q / TB], PB[cur][q % TB], acc[q / TB][q % TB]);
          __builtin_amdgcn_sched_group_barrier(SGB_DSR, 2 * TD, 0);
#pragma unroll
          for (int q = 0; q < Q1; q++) {
            __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SGB_VALU, 5, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          decode_A(nx, wA[s + 1]);
#pragma unroll
          for (int q = Q1; q < NQ; q++) acc[q / TB][q % TB] = MFMA_I8(PA[cur][q / TB], PB[cur][q % TB], acc[q / TB][q % TB]);
#pragma unroll
          for (int q = Q1; q < NQ; q++) {
            __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SGB_VALU, 5, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        buf ^= 1;
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      // single-exit loop over triples: groups past k1 run with an all-missing A word (zero planes, adds nothing)
      for (int64_t kg = k0; kg < k1; kg += 3) {
        group(I0{}, I1{}, I2{}, kg);
        group(I1{}, I2{}, I0{}, kg + 1);
        group(I2{}, I0{}, I1{}, kg + 2);
      }
    }
    long long* slab = slabs + u0 * PCA_SLAB_INTS + lane;
#pragma unroll
    for (int tb = 0; tb < TB; tb++)
#pragma unroll
      for (int q = 0; q < 16; q++) {
        long long c = 0;
#pragma unroll
        for (int t = 0; t < TD; t++) c += (long long)acc[t][tb][q] << (7 * t);
        c <<= 7 * t0;  // wrapping shift is fine: two's complement, |c| stays far below 2^62
        atomicAdd((unsigned long long*)(slab + (tb * 16 + q) * 64), (unsigned long long)c);
      }
  }
}

// Which unit holds element (i, k) of S = G W G', and where in its slab.  Full super-tiles (4 row tiles with data)
// are the columns of the regular units (ia, jb >= ia / 4); the 1..3 row tiles left over past the last full
// super-tile ("remainder" tiles, first one F4 = 4 nsbf) are kept as A ROW tiles against every full super-tile --
// element (i, k) with k in the remainder is read from the transposed unit -- plus one unit against the partial
// super-tile nsbf for the remainder x remainder corner.  (Treating the partial super-tile as a column of every
// row tile instead would spend 4 MFMAs per step on it for every row tile with 1..3 of them useful: 3.7 % of the
// kernel at N = 5 000.)
__device__ __forceinline__ void tpg_gram_locate(int nsbf, int i, int k, int& ia, int& jb, int& tb, int& row, int& col) {
  const int F4 = 4 * nsbf;
  int ti = i >> 5, tk = k >> 5;
  if (ti < F4 && tk < F4) {
    if ((tk >> 2) < (ti >> 2)) { int t = i; i = k; k = t; ti = i >> 5; tk = k >> 5; }
    ia = ti; jb = tk >> 2; tb = tk & 3;
  } else if (ti >= F4 && tk >= F4) {
    ia = ti; jb = nsbf; tb = tk - F4;
  } else {
    if (ti < F4) { int t = i; i = k; k = t; ti = i >> 5; tk = k >> 5; }  // i in the remainder, k in the full part
    ia = ti; jb = tk >> 2; tb = tk & 3;
  }
  row = i & 31; col = k & 31;
}

// K[i + k n] = 2^-F S[i][k] - r_i - r_k + C  (both triangles); lut[ia (nsbf + 1) + jb] = slab of unit (ia, jb)
__global__ void tpg_pca_assemble_kernel(const long long* __restrict__ slabs, const int32_t* __restrict__ lut, int nsbf,
                                        int n, int F, const double* __restrict__ rvec, double Cc,
                                        double* __restrict__ K) {
  const int64_t total = (int64_t)n * n;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int oi = (int)(idx % n), ok = (int)(idx / n);
    int ia, jb, tb, row, col;
    tpg_gram_locate(nsbf, oi, ok, ia, jb, tb, row, col);
    const int lane = col + 32 * ((row >> 2) & 1);
    const int reg = (row & 3) + 4 * (row >> 3);
    const long long s = slabs[(int64_t)lut[ia * (nsbf + 1) + jb] * PCA_SLAB_INTS + (tb * 16 + reg) * 64 + lane];
    double val = ldexp((double)s, -F);
    if (rvec) val = val - rvec[oi] - rvec[ok] + Cc;
    K[idx] = val;
  }
}

// Double centering K <- H K H, H = I - 11'/n.  With center_j = column mean of the SAME individuals,
// Z = H G W^(1/2), so the Gram matrix is exactly the double-centered weighted cross-product S' = G W G':
// r_i = row mean of S', C = grand mean.  Row means from the (symmetric) columns: coalesced.
__global__ __launch_bounds__(256) void tpg_colmean_kernel(const double* __restrict__ K, int n, double* __restrict__ r) {
  __shared__ double sh[256];
  const int col = blockIdx.x;
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += 256) acc += K[i + (int64_t)col * n];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) r[col] = sh[0] / n;
}

__global__ __launch_bounds__(256) void tpg_mean_kernel(const double* __restrict__ r, int n, double* __restrict__ out) {
  __shared__ double sh[256];
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += 256) acc += r[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0] / n;
}

__global__ void tpg_double_center_kernel(double* __restrict__ K, int n, const double* __restrict__ r,
                                         const double* __restrict__ grand) {
  const int64_t total = (int64_t)n * n;
  const double C0 = grand[0];
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx % n), k = (int)(idx / n);
    K[idx] = K[idx] - r[i] - r[k] + C0;
  }
}

// flags[0] |= (center_j != column mean from the counts)
__global__ void tpg_center_is_mean_kernel(const int4* __restrict__ counts, const double* __restrict__ center,
                                          int64_t m, int64_t n, int* __restrict__ flags) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = counts[j];
    if (c.w != 0) flags[1] = 1;
    if (center[j] != (double)(c.y + 2 * c.z) / (double)n) flags[0] = 1;
  }
}

__global__ __launch_bounds__(256) void tpg_dot_kernel(const double* __restrict__ a, const double* __restrict__ b,
                                                      int64_t n, double* __restrict__ part) {
  __shared__ double sh[256];
  double acc = 0;
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
    acc += a[j] * b[j];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

// Gram matrix into d_K (device, n x n column-major) given device center / scale.
// own_center: center_j is the column mean over the view's own individuals (big_SVD's case) -> double centering;
// otherwise the general form K = S' - r 1' - 1 r' + C with r_i = sum_j w_j c_j g_ij, C = sum_j w_j c_j^2.
// out[0] = bit pattern of the smallest scale (positive doubles order like their bit patterns), out[1] = 1 + index
// of some locus whose scale is not a positive number (0 if none)
__global__ void tpg_scale_range_kernel(const double* __restrict__ scale, int64_t m, unsigned long long* __restrict__ out) {
  unsigned long long mn = 0x7FF0000000000000ull, bad = 0;
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
    const double sc = scale[j];
    if (!(sc > 0)) bad = (unsigned long long)j + 1;
    else { const unsigned long long b = (unsigned long long)__double_as_longlong(sc); mn = b < mn ? b : mn; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long omn = __shfl_xor(mn, o), obad = __shfl_xor(bad, o);
    mn = omn < mn ? omn : mn;
    bad = obad > bad ? obad : bad;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(out, mn);
    if (bad) atomicMax(out + 1, bad);
  }
}

__global__ void tpg_pca_weights_kernel(const double* __restrict__ scale, int64_t m, double* __restrict__ w) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x)
    w[j] = 1.0 / (scale[j] * scale[j]);
}
__global__ void tpg_pca_wc_kernel(const double* __restrict__ what, const double* __restrict__ center, int64_t m,
                                  double* __restrict__ wc) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x)
    wc[j] = what[j] * center[j];
}

static int pca_gram_device(tpg_ctx* ctx, const tpg_view* v, const double* d_center, const double* d_scale,
                           double* d_K, bool own_center) {
  const int64_t n = v->n, m = v->m;
  // weight range decides the number of digits: smallest scale by a device reduction (16 bytes come back)
  unsigned long long* d_rng = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_rng, 2 * sizeof(unsigned long long)));
  const unsigned long long rng_init[2] = {0x7FF0000000000000ull, 0ull};  // +inf, no bad entry
  unsigned long long rng[2];
  hipError_t er = tpg_push_small(ctx, d_rng, rng_init, sizeof(rng_init));
  if (er == hipSuccess) {
    hipLaunchKernelGGL(tpg_scale_range_kernel, dim3(512), dim3(256), 0, ctx->stream, d_scale, m, d_rng);
    er = tpg_fetch_small(ctx, rng, d_rng, sizeof(rng));
  }
  tpg_pfree(d_rng);
  TPG_HIP(er);
  TPG_REQUIRE(rng[1] == 0, TPG_ENUMERIC, "zero or negative scale at locus %lld", (long long)(rng[1] - 1));
  double smin;
  memcpy(&smin, &rng[0], sizeof(double));
  const double wmax = 1.0 / (smin * smin);
  const int wbits = (int)ceil(log2(wmax + 1.0));
  int T = (wbits + ctx->pca_digit_fbits + 1 + 6) / 7;
  if (T > 8 && ctx->pca_digit_fbits > 22) T = std::max(8, (wbits + 22 + 1 + 6) / 7);  // (extra bits are a wish, 22 the requirement)
  if (T < 4) T = 4;
  TPG_REQUIRE(T <= 8, TPG_ENUMERIC, "per-locus weight range too wide (max 1/scale^2 = %g)", wmax);
  const int F = 7 * T - 1 - wbits;

  // units: see tpg_gram_locate.  nrtv row tiles hold data, nsbf full super-tiles, rem remainder tiles
  const int nrt = (int)(v->Q * 4);
  const int nrtv = (int)ceil_div(n, 32), nsbf = nrtv / PCA_TB, rem = nrtv - PCA_TB * nsbf;
  std::vector<int2> order;
  for (int pj = 0; pj * 8 < nsbf; pj++) {  // patch order: super-column blocks of 8, inside them the super-rows
    const int j1 = std::min(nsbf, pj * 8 + 8);
    for (int a = 0; a < j1; a++)
      for (int jb = std::max(pj * 8, a); jb < j1; jb++)
        for (int r = 0; r < PCA_TB; r++) order.push_back(make_int2(PCA_TB * a + r, jb));
  }
  for (int jb = 0; jb <= nsbf && rem > 0; jb++)
    for (int r = 0; r < rem; r++) order.push_back(make_int2(PCA_TB * nsbf + r, jb));
  const int64_t nun = (int64_t)order.size();
  std::vector<int32_t> lut((size_t)nrt * (size_t)(nsbf + 1), -1);
  for (int64_t u = 0; u < nun; u++) lut[(size_t)order[(size_t)u].x * (size_t)(nsbf + 1) + (size_t)order[(size_t)u].y] = (int32_t)u;
  uint32_t* d_DG = nullptr;
  double *d_what = nullptr, *d_wc = nullptr, *d_r = nullptr, *d_part = nullptr;
  long long* d_slabs = nullptr;
  int2* d_order = nullptr;
  int32_t* d_lut = nullptr;
  int rc = TPG_OK;
  hipError_t e = hipSuccess;
#define GHIP(call) do { if (e == hipSuccess) { e = (call); if (e != hipSuccess) tpg_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e)); } } while (0)
  GHIP(tpg_pmalloc((void**)&d_what, sizeof(double) * (size_t)m));
  GHIP(tpg_pmalloc((void**)&d_wc, sizeof(double) * (size_t)m));
  GHIP(tpg_pmalloc((void**)&d_r, sizeof(double) * (size_t)n));
  GHIP(tpg_pmalloc((void**)&d_part, sizeof(double) * 512));
  // S' = G W G' by weight classes on the FP4 matrix cores when the weights take few distinct values (gramcls.hip:
  // always so under the binomial scaling); the digit-split int8 kernel below otherwise
  bool by_classes = false;
  if (e == hipSuccess) {
    double* d_w = nullptr;
    GHIP(tpg_pmalloc((void**)&d_w, sizeof(double) * (size_t)m));
    if (e == hipSuccess) {
      TPG_LAUNCH(ctx, "pca_weights", tpg_pca_weights_kernel, dim3(1024), dim3(256), 0, d_scale, m, d_w);
      rc = tpg_gram_classes(ctx, v, d_w, d_what, d_K, &by_classes, own_center);
      if (rc == TPG_OK && by_classes)
        TPG_LAUNCH(ctx, "pca_weights", tpg_pca_wc_kernel, dim3(1024), dim3(256), 0, (const double*)d_what, d_center, m, d_wc);
    }
    if (d_w) tpg_pfree(d_w);
    if (rc != TPG_OK) { tpg_pfree(d_what); tpg_pfree(d_wc); tpg_pfree(d_r); tpg_pfree(d_part); return rc; }
  }
  if (!by_classes) {
  if (e == hipSuccess && tpg_view_need_T(ctx, v) != TPG_OK) { tpg_pfree(d_what); tpg_pfree(d_wc); tpg_pfree(d_r); tpg_pfree(d_part); return TPG_EHIP; }
  GHIP(tpg_pmalloc((void**)&d_DG, (size_t)v->KG * 4 * 2 * T * 8 * sizeof(uint32_t)));
  GHIP(tpg_pmalloc((void**)&d_slabs, sizeof(long long) * (size_t)nun * PCA_SLAB_INTS));
  GHIP(hipMemsetAsync(d_slabs, 0, sizeof(long long) * (size_t)nun * PCA_SLAB_INTS, ctx->stream));
  }
  if (e == hipSuccess && !by_classes) {
    TPG_LAUNCH(ctx, "pca_digits", tpg_pca_digits_kernel, dim3(1024), dim3(256), 0, d_scale, d_center, m, v->KG, F, T,
               d_DG, d_what, d_wc);
    GHIP(tpg_pmalloc((void**)&d_order, sizeof(int2) * (size_t)nun));
    GHIP(hipMemcpyAsync(d_order, order.data(), sizeof(int2) * (size_t)nun, hipMemcpyHostToDevice, ctx->stream));
    GHIP(tpg_pmalloc((void**)&d_lut, sizeof(int32_t) * lut.size()));
    GHIP(hipMemcpyAsync(d_lut, lut.data(), sizeof(int32_t) * lut.size(), hipMemcpyHostToDevice, ctx->stream));
    // K-split S: units x S wave-units over the resident waves (one workgroup per CU, one wave per SIMD; grid a
    // multiple of the 8 XCDs).  Cost model: a launch is rounds(S) = ceil(units S / waves) rounds, a round costs its K
    // range (about 1.17 us per 128-locus group: 64 MFMAs at ~37 cycles) plus the flush of the accumulators (256
    // 64-bit atomic wave-instructions per wave, all waves at once: ~65 us, fitted on the 1 000 x 650 000 launch).  A
    // small S leaves waves idle in the last round, a large one pays a flush per round.
    int nblk = ctx->num_cu / 8 * 8;
    if (nblk < 8) nblk = 8;
    int bestS = 1;
    double best = -1;
    const int64_t maxS = v->KG / 8 > 0 ? (v->KG / 8 < 96 ? v->KG / 8 : 96) : 1;
    for (int64_t S = 1; S <= maxS; S++) {
      const int64_t rounds = ceil_div(nun * S, 4 * (int64_t)nblk);
      const double cost = (double)rounds * ((double)ceil_div(v->KG, S) * 1.17 + 65.0);
      if (best < 0 || cost < best * 0.995) { best = cost; bestS = (int)S; }
    }
    const unsigned grid = (unsigned)nblk;
    int64_t pass_base = 0;  // dwords
    for (int t0 = 0; t0 < T; t0 += 4) {
      const int td = T - t0 < 4 ? T - t0 : 4;
      const uint4* dgp = (const uint4*)(d_DG + pass_base);
#define GRAM_LAUNCH(TD)                                                                                              \
  TPG_LAUNCH(ctx, "pca_gram_mfma", tpg_pca_gram_kernel<TD>, dim3(grid), dim3(256), 0, (const uint4*)v->T, v->KG,      \
             (int64_t)0, v->KG, dgp, t0, (const int2*)d_order, (int)nun, bestS, d_slabs)
      if (td == 4) GRAM_LAUNCH(4);
      else if (td == 3) GRAM_LAUNCH(3);
      else if (td == 2) GRAM_LAUNCH(2);
      else GRAM_LAUNCH(1);
#undef GRAM_LAUNCH
      pass_base += v->KG * 4 * 2 * td * 8;
    }
    GHIP(hipGetLastError());
  }
  if (e == hipSuccess && own_center) {
    if (!by_classes)
      TPG_LAUNCH(ctx, "pca_assemble", tpg_pca_assemble_kernel, dim3(2048), dim3(256), 0, d_slabs, (const int32_t*)d_lut, nsbf,
                 (int)n, F, (const double*)nullptr, 0.0, d_K);
    TPG_LAUNCH(ctx, "pca_colmean", tpg_colmean_kernel, dim3((unsigned)n), dim3(256), 0, (const double*)d_K, (int)n, d_r);
    TPG_LAUNCH(ctx, "pca_colmean", tpg_mean_kernel, dim3(1), dim3(256), 0, (const double*)d_r, (int)n, d_part);
    TPG_LAUNCH(ctx, "pca_double_center", tpg_double_center_kernel, dim3(2048), dim3(256), 0, d_K, (int)n,
               (const double*)d_r, (const double*)d_part);
    GHIP(hipGetLastError());  // (no wait: the scratch blocks go back to the pool in stream order)
  }
  if (e == hipSuccess && !own_center) {
    // r_i = sum_j what_j c_j g_ij  (RAW sweep with a one-column table), C = sum_j what_j c_j^2
    rc = tpg_view_need_T(ctx, v);
    if (rc == TPG_OK) rc = run_sweep(ctx, SW_RAW, v->T, v->Q * 4, v->KG, n, m, nullptr, nullptr, d_wc, m, 1, d_r, nullptr, nullptr);
  }
  double Cc = 0;
  if (e == hipSuccess && rc == TPG_OK && !own_center) {
    TPG_LAUNCH(ctx, "pca_dot", tpg_dot_kernel, dim3(512), dim3(256), 0, d_wc, d_center, m, d_part);
    std::vector<double> hp(512);
    GHIP(tpg_fetch_small(ctx, hp.data(), d_part, sizeof(double) * 512));
    long double s = 0;
    for (int b = 0; b < 512; b++) s += hp[(size_t)b];
    Cc = (double)s;
  }
  if (e == hipSuccess && rc == TPG_OK && !own_center) {
    if (!by_classes) {
      TPG_LAUNCH(ctx, "pca_assemble", tpg_pca_assemble_kernel, dim3(2048), dim3(256), 0, d_slabs, (const int32_t*)d_lut, nsbf,
                 (int)n, F, (const double*)d_r, Cc, d_K);
    } else {  // K = S' - r 1' - 1 r' + C on the matrix the class path left in d_K
      GHIP(tpg_push_small(ctx, d_part, &Cc, sizeof(double)));
      TPG_LAUNCH(ctx, "pca_double_center", tpg_double_center_kernel, dim3(2048), dim3(256), 0, d_K, (int)n,
                 (const double*)d_r, (const double*)d_part);
    }
    GHIP(hipGetLastError());
  }
#undef GHIP
  tpg_pfree(d_DG); tpg_pfree(d_what); tpg_pfree(d_wc); tpg_pfree(d_r); tpg_pfree(d_part);
  tpg_pfree(d_slabs);
  tpg_pfree(d_order);
  tpg_pfree(d_lut);
  if (e != hipSuccess) return TPG_EHIP;
  return rc;
}

// K = H S' H for S' in d_K (the double centering pca_gram_device applies when `center` is the column mean): linear in
// S', so every rank applies it to the S' of the classes it owns before the partial matrices are summed
static int pca_double_center_inplace(tpg_ctx* ctx, double* d_K, int64_t n) {
  double *d_r = nullptr, *d_part = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_r, sizeof(double) * (size_t)n));
  hipError_t e = tpg_pmalloc((void**)&d_part, sizeof(double) * 512);
  if (e == hipSuccess) {
    TPG_LAUNCH(ctx, "pca_colmean", tpg_colmean_kernel, dim3((unsigned)n), dim3(256), 0, (const double*)d_K, (int)n, d_r);
    TPG_LAUNCH(ctx, "pca_colmean", tpg_mean_kernel, dim3(1), dim3(256), 0, (const double*)d_r, (int)n, d_part);
    TPG_LAUNCH(ctx, "pca_double_center", tpg_double_center_kernel, dim3(2048), dim3(256), 0, d_K, (int)n, (const double*)d_r,
               (const double*)d_part);
    e = hipGetLastError();
  }
  tpg_pfree(d_r);
  tpg_pfree(d_part);
  TPG_HIP(e);
  return TPG_OK;
}

extern "C" int tpg_pca_gram(tpg_ctx* ctx, const tpg_view* v, const double* center, const double* scale, double* K) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && center && scale && K, TPG_EINVAL, "null argument");
  InBuf ic, is;
  TPG_TRY(ic.init(ctx, center, sizeof(double) * (size_t)v->m));
  TPG_TRY(is.init(ctx, scale, sizeof(double) * (size_t)v->m));
  OutBuf ok;
  TPG_TRY(ok.init(K, sizeof(double) * (size_t)v->n * (size_t)v->n));
  // is `center` the column mean of these individuals (what tpg_pca_center_scale returns)?
  int32_t* d_counts = nullptr;
  int* d_flag = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_counts, sizeof(int32_t) * 4 * (size_t)v->m));
  hipError_t e = tpg_pmalloc((void**)&d_flag, 2 * sizeof(int));
  int flag[2] = {1, 0};
  int rc = e == hipSuccess ? tpg_launch_loci_counts(ctx, v, d_counts) : TPG_EHIP;
  if (rc == TPG_OK) {
    e = hipMemsetAsync(d_flag, 0, 2 * sizeof(int), ctx->stream);
    TPG_LAUNCH(ctx, "pca_center_check", tpg_center_is_mean_kernel, dim3(1024), dim3(256), 0, (const int4*)d_counts,
               ic.dev<double>(), v->m, v->n, d_flag);
    if (e == hipSuccess) e = tpg_fetch_small(ctx, flag, d_flag, 2 * sizeof(int));
    if (e != hipSuccess) { tpg_set_error("pca_gram: %s", hipGetErrorString(e)); rc = TPG_EHIP; }
  }
  tpg_pfree(d_counts);
  tpg_pfree(d_flag);
  TPG_TRY(rc);
  TPG_REQUIRE(!flag[1], TPG_ENUMERIC, "You can't have missing values in 'X'.");
  TPG_TRY(pca_gram_device(ctx, v, ic.dev<double>(), is.dev<double>(), ok.dev<double>(), flag[0] == 0));
  return ok.commit(ctx);
}

// K += Gram matrix of this view's loci (K in device memory): the Gram matrix is additive over loci, so a caller that
// receives the genotypes block of loci by block of loci (tpg_fbm_upload_cols) accumulates it block by block -- each
// block with its own per-locus center and scale -- and runs tpg_sym_eig_topk once at the end.
__global__ void tpg_add_inplace_kernel(double* __restrict__ y, const double* __restrict__ x, int64_t count) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) y[i] += x[i];
}

extern "C" int tpg_pca_gram_add(tpg_ctx* ctx, const tpg_view* v, const double* center, const double* scale, double* K) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && center && scale && K, TPG_EINVAL, "null argument");
  TPG_REQUIRE(tpg_is_device_ptr(K), TPG_EINVAL, "K must be device memory");
  double* d_tmp = nullptr;
  const int64_t nn = v->n * v->n;
  TPG_HIP(tpg_pmalloc((void**)&d_tmp, sizeof(double) * (size_t)nn));
  int rc = tpg_pca_gram(ctx, v, center, scale, d_tmp);
  if (rc == TPG_OK) {
    TPG_LAUNCH(ctx, "pca_gram_add", tpg_add_inplace_kernel, dim3(2048), dim3(256), 0, K, (const double*)d_tmp, nn);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { tpg_set_error("gram add: %s", hipGetErrorString(e)); rc = TPG_EHIP; }
  }
  tpg_pfree(d_tmp);
  return rc;
}

// ---------------------------------------------------------------------------
// dense FP64 helpers for the eigen solver (all matrices column-major)

// part[split][row + c n] = sum_{k in split} K[row + k n] * Q[k + c n]   (c < b <= 64)
// FP64 MFMA (v_mfma_f64_16x16x4_f64): one wave owns 16 rows x 64 columns (4 accumulator tiles), so every
// element of K is read exactly once per product; the workgroup's 4 waves share the 64-deep slice of Q
// staged in LDS.  A = K[row0 + (l & 15)][k], B = Q[k][col + (l & 15)] with k = 16 (l >> 4) + u in MFMA step u of a chunk
// (any assignment of the chunk's 64 k to (step, lane quarter) is as good as another: the sum runs over all of them), so a
// lane's sixteen K values of a chunk are CONTIGUOUS in a row of K -- read as K[k + row n] (K is symmetric): one 128-byte
// line per lane in eight 16-byte loads, no predicates (indices past n are clamped: such a row is never stored, such a k
// meets a zero row of the Q slice), where sixteen 8-byte loads 4 n doubles apart, each under its own bounds test, made
// this kernel 126 address computations and 80 exec-mask branches per 64 MFMAs.
// C/D: col = l & 15, row = (l >> 4) + 4 reg.
__global__ __launch_bounds__(256) void tpg_symm_apply_kernel(const double* __restrict__ K, int n,
                                                             const double* __restrict__ Q, int b, int S,
                                                             double* __restrict__ part) {
  // two buffers of the 64-deep slice of Q: the slice and the K values of chunk c + 1 are fetched (into registers) before
  // the MFMAs of chunk c and written to the other buffer after them, one barrier per chunk
  __shared__ double qs[2][64][65];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int row0 = (blockIdx.x * 4 + wv) * 16;
  const int row = row0 + r16;
  const int split = blockIdx.y;
  // split boundaries on multiples of 64 so that chunks never straddle
  const int nchunk = (n + 63) / 64;
  const int cbeg = (int)((int64_t)nchunk * split / S), cend = (int)((int64_t)nchunk * (split + 1) / S);
  v4d acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ct++) acc[ct] = (v4d){0, 0, 0, 0};
  double av[16], qv[16];
  typedef double sa_v2d __attribute__((ext_vector_type(2)));
  typedef sa_v2d sa_v2d_a8 __attribute__((aligned(8)));  // (a row of K starts on an 8-byte boundary only when n is odd)
  const double* Krow = K + (int64_t)(row < n ? row : n - 1) * n;
  auto fetch = [&](int ch) {  // K values of this lane and this thread's 16 entries of the Q slice, chunk ch
    const int k0 = ch * 64;
    const int kb = k0 + 16 * kq;
    if (kb + 16 <= n) {  // (false only in the last chunk)
#pragma unroll
      for (int u = 0; u < 16; u += 2) {
        const sa_v2d t = *(const sa_v2d_a8*)(Krow + kb + u);
        av[u] = t[0]; av[u + 1] = t[1];
      }
    } else {
#pragma unroll
      for (int u = 0; u < 16; u++) av[u] = Krow[kb + u < n ? kb + u : n - 1];
    }
    // the Q slice: column wv + 4 u (wave-uniform), row k0 + lane: zero past n and past b, both tests scalar but for the last chunk
    const int wvu = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool inside = k0 + 64 <= n || k0 + lane < n;
    const double* Qk = Q + (k0 + lane < n ? k0 + lane : n - 1);
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int c = wvu + 4 * u;
      double t = 0.0;
      if (c < b) t = Qk[(int64_t)c * n];
      qv[u] = inside ? t : 0.0;
    }
  };
  if (cbeg < cend) {
    fetch(cbeg);
#pragma unroll
    for (int u = 0; u < 16; u++) { const int idx = threadIdx.x + 256 * u; qs[0][idx & 63][idx >> 6] = qv[u]; }
  }
  for (int ch = cbeg; ch < cend; ch++) {
    const int cur = (ch - cbeg) & 1;
    __syncthreads();  // qs[cur] is complete, qs[cur ^ 1] has been read by everybody
    double a0[16];
#pragma unroll
    for (int u = 0; u < 16; u++) a0[u] = av[u];
    if (ch + 1 < cend) fetch(ch + 1);
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int ct = 0; ct < 4; ct++)
        acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], qs[cur][16 * kq + u][ct * 16 + r16], acc[ct], 0, 0, 0);
    }
    if (ch + 1 < cend) {
#pragma unroll
      for (int u = 0; u < 16; u++) { const int idx = threadIdx.x + 256 * u; qs[cur ^ 1][idx & 63][idx >> 6] = qv[u]; }
    }
  }
#pragma unroll
  for (int ct = 0; ct < 4; ct++) {
    const int col = ct * 16 + r16;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      const int orow = row0 + kq + 4 * reg;
      if (orow < n && col < b) part[((int64_t)split * b + col) * n + orow] = acc[ct][reg];
    }
  }
}

// Y = alpha * (sum_split part - Dm) + beta * Y1 + gamma * Y0   (Dm / Y1 / Y0 may be null)
__global__ void tpg_combine_kernel(const double* __restrict__ part, int S, int64_t nb, double alpha,
                                   const double* __restrict__ Y1, double beta, const double* __restrict__ Y0,
                                   double gamma, double* __restrict__ Y, const double* __restrict__ Dm = nullptr) {
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < nb; idx += (int64_t)gridDim.x * blockDim.x) {
    double s = 0;
    for (int sp = 0; sp < S; sp++) s += part[(int64_t)sp * nb + idx];
    if (Dm) s -= Dm[idx];
    double y = alpha * s;
    if (Y1) y += beta * Y1[idx];
    if (Y0) y += gamma * Y0[idx];
    Y[idx] = y;
  }
}

// partial[chunk][i + j p] = sum_{rows in chunk} A[row + i n] * B[row + j n]   (p, b <= 64)
__global__ __launch_bounds__(256) void tpg_gram_small_kernel(const double* __restrict__ A, int p,
                                                             const double* __restrict__ B, int b, int n,
                                                             int rows_per_chunk, double* __restrict__ part) {
  __shared__ double as[64][33], bs[64][33];
  const int chunk = blockIdx.x;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = r0 + rows_per_chunk < n ? r0 + rows_per_chunk : n;
  // each thread owns outputs (i, j) with i = tid & 63 (< p), j = (tid >> 6) + 4 jj
  const int i = threadIdx.x & 63, jq = threadIdx.x >> 6;
  double acc[16];
#pragma unroll
  for (int c = 0; c < 16; c++) acc[c] = 0;
  for (int rr = r0; rr < r1; rr += 32) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {
      const int col = idx >> 5, dr = idx & 31;
      as[col][dr] = (col < p && rr + dr < r1) ? A[(rr + dr) + (int64_t)col * n] : 0.0;
      bs[col][dr] = (col < b && rr + dr < r1) ? B[(rr + dr) + (int64_t)col * n] : 0.0;
    }
    __syncthreads();
    for (int dr = 0; dr < 32; dr++) {
      const double av = as[i][dr];
#pragma unroll
      for (int c = 0; c < 16; c++) acc[c] += av * bs[jq + 4 * c][dr];
    }
  }
  if (i < p)
#pragma unroll
    for (int c = 0; c < 16; c++) {
      const int j = jq + 4 * c;
      if (j < b) part[(int64_t)chunk * p * b + i + (int64_t)j * p] = acc[c];
    }
}

// C[i + j p] = rowscale[i] * sum_chunks part[chunk][i + j p]   (device-side finish of gram_small).  64 outputs per
// workgroup, its four waves take every fourth chunk each (a single thread per output walked all ~80 - 160 chunks with
// dependent adds: 26 us per call); the four partial sums are added in wave order, so every run gives the same sums.
__global__ __launch_bounds__(256) void tpg_gram_reduce_kernel(const double* __restrict__ part, int nchunks, int p, int b,
                                                              const double* __restrict__ rowscale, double* __restrict__ Cm) {
  __shared__ double sh[4][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;
  double s = 0;
  if (idx < p * b)
    for (int c = sl; c < nchunks; c += 4) s += part[(int64_t)c * p * b + idx];
  sh[sl][o] = s;
  __syncthreads();
  if (sl == 0 && idx < p * b) {
    s = ((sh[0][o] + sh[1][o]) + sh[2][o]) + sh[3][o];
    if (rowscale) s *= rowscale[idx % p];
    Cm[idx] = s;
  }
}

// Y[row + j n] = sum_i A[row + i n] * X[i + j p]   (X small, p <= 64, b2 <= 64); optional residual
// form: Y = A X - Bm * diag(theta) is done by the caller through two calls / combine.
__global__ __launch_bounds__(256) void tpg_right_mult_kernel(const double* __restrict__ A, int n, int p,
                                                             const double* __restrict__ X, int b2,
                                                             double* __restrict__ Y) {
  __shared__ double xs[64 * 64];
  for (int idx = threadIdx.x; idx < p * b2; idx += 256) xs[idx] = X[idx];
  __syncthreads();
  const int row = blockIdx.x * 32 + (threadIdx.x & 31);
  const int cg = threadIdx.x >> 5;  // 8 groups of 8 columns
  if (row >= n) return;
  double acc[8];
#pragma unroll
  for (int c = 0; c < 8; c++) acc[c] = 0;
  for (int i = 0; i < p; i++) {
    const double a = A[row + (int64_t)i * n];
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const int j = cg * 8 + c;
      acc[c] += a * (j < b2 ? xs[i + j * p] : 0.0);
    }
  }
#pragma unroll
  for (int c = 0; c < 8; c++) {
    const int j = cg * 8 + c;
    if (j < b2) Y[row + (int64_t)j * n] = acc[c];
  }
}

// Rayleigh-Ritz rotation and residuals in one pass over the block: Y0 = A X (Ritz vectors), Y1 = Yk X (K times them,
// Yk = K A) and, per workgroup of 32 rows, the column sums of (Y1 - Y0 diag(theta))^2 -- only the residual NORMS are
// wanted, not the Gram matrix of the residuals; tpg_gram_reduce_kernel adds the partial sums in block order.
__global__ __launch_bounds__(256) void tpg_ritz_kernel(const double* __restrict__ A, const double* __restrict__ Yk, int n,
                                                       int p, const double* __restrict__ X,
                                                       const double* __restrict__ theta, double* __restrict__ Y0,
                                                       double* __restrict__ Y1, double* __restrict__ rpart) {
  __shared__ double xs[64 * 64];
  for (int idx = threadIdx.x; idx < p * p; idx += 256) xs[idx] = X[idx];
  __syncthreads();
  const int row = blockIdx.x * 32 + (threadIdx.x & 31);
  const int cg = threadIdx.x >> 5;  // 8 groups of 8 columns
  const bool live = row < n;
  double a0[8], a1[8];
#pragma unroll
  for (int c = 0; c < 8; c++) a0[c] = 0, a1[c] = 0;
  if (live)
    for (int i = 0; i < p; i++) {
      const double a = A[row + (int64_t)i * n], y = Yk[row + (int64_t)i * n];
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const int j = cg * 8 + c;
        const double x = j < p ? xs[i + j * p] : 0.0;
        a0[c] += a * x;
        a1[c] += y * x;
      }
    }
#pragma unroll
  for (int c = 0; c < 8; c++) {
    const int j = cg * 8 + c;
    double r = 0;
    if (live && j < p) {
      Y0[row + (int64_t)j * n] = a0[c];
      Y1[row + (int64_t)j * n] = a1[c];
      r = a1[c] - theta[j] * a0[c];
      r *= r;
    }
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) r += __shfl_xor(r, off, 32);
    if ((threadIdx.x & 31) == 0 && j < p) rpart[(int64_t)blockIdx.x * p + j] = r;
  }
}

__global__ void tpg_fill_random_kernel(double* __restrict__ Q, int64_t total, uint64_t seed) {
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = (uint64_t)idx * 0x9E3779B97F4A7C15ull + seed;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    Q[idx] = (double)(int64_t)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}

#include "host/host_eig.h"  // host_cholesky_upper, host_upper_inverse, host_sym_eig (b <= 64)

struct EigWork {
  tpg_ctx* ctx;
  const double* K;
  int n, b, S;
  double *part = nullptr, *gpart = nullptr;
  int nchunks, rows_per_chunk;
  // deflation: K' = K - L diag(lam) L' for the nl locked eigenpairs (L = first nl columns of the block)
  const double* L = nullptr;
  int nl = 0;
  double *lam_dev = nullptr, *cdev = nullptr, *dtmp = nullptr;
  // small host matrices go to the device, and the small products come back, through the context's mailbox (runtime.hip:
  // tpg_push_small / tpg_fetch_small): no copy engine and no stream synchronisation anywhere in the iteration
  double* xdev = nullptr;
  int init() {
    // K splits: ONE round of workgroups (two fit a CU: 66 KiB of LDS each) -- at n = 5 000, 79 row blocks x 6 splits = 474
    // of 512 places, 89.5 us per product; 16 splits (2.5 rounds, the last half empty) 94.6, 13 splits (two rounds and three
    // workgroups) 108 (tools/eig_s_scan.py)
    const int row_blocks = (n + 63) / 64;
    S = 1;
    while (row_blocks * S < 4 * ctx->num_cu && S < 32 && n / (S * 2) >= 64) S *= 2;  // (short ranges: 1 000 rows take 4)
    S = std::max(1, std::min((int)S, 2 * ctx->num_cu / row_blocks));
    if (getenv("TPG_EIG_S")) S = std::max(1, atoi(getenv("TPG_EIG_S")));  // (experiments)
    // A'B over chunks of rows: a workgroup per 32 (64) rows, so that a product on a few thousand rows is one short round
    // of many workgroups instead of a long loop in a few
    rows_per_chunk = n <= 4096 ? 32 : 64;
    nchunks = (n + rows_per_chunk - 1) / rows_per_chunk;
    TPG_HIP(tpg_pmalloc((void**)&part, sizeof(double) * (size_t)S * (size_t)b * (size_t)n));
    TPG_HIP(tpg_pmalloc((void**)&gpart, sizeof(double) * (size_t)nchunks * 64 * 64));  // nchunks = n/32
    TPG_HIP(tpg_pmalloc((void**)&lam_dev, sizeof(double) * 64));
    TPG_HIP(tpg_pmalloc((void**)&cdev, sizeof(double) * 64 * 64));
    TPG_HIP(tpg_pmalloc((void**)&dtmp, sizeof(double) * (size_t)b * (size_t)n));
    TPG_HIP(tpg_pmalloc((void**)&xdev, sizeof(double) * (64 * 64 + 64)));
    return TPG_OK;
  }
  ~EigWork() {
    if (part) tpg_pfree(part);
    if (gpart) tpg_pfree(gpart);
    if (lam_dev) tpg_pfree(lam_dev);
    if (cdev) tpg_pfree(cdev);
    if (dtmp) tpg_pfree(dtmp);
    if (xdev) tpg_pfree(xdev);
  }
  int set_locked(const double* Lptr, int count, const double* lam_host) {
    L = Lptr;
    nl = count;
    if (count > 0) TPG_HIP(tpg_push_small(ctx, lam_dev, lam_host, sizeof(double) * (size_t)count));
    return TPG_OK;
  }
  // Y = alpha K' Q + beta Y1 + gamma Y0   with K' = K - L diag(lam) L'; no host synchronisation
  int apply(const double* Q, double alpha, const double* Y1, double beta, const double* Y0, double gamma, double* Y) {
    dim3 grid((unsigned)((n + 63) / 64), (unsigned)S);  // 4 waves x 16 rows per workgroup
    TPG_LAUNCH(ctx, "eig_symm_apply", tpg_symm_apply_kernel, grid, dim3(256), 0, K, n, Q, b, S, part);
    const double* Dm = nullptr;
    if (nl > 0) {
      TPG_LAUNCH(ctx, "eig_gram_small", tpg_gram_small_kernel, dim3((unsigned)nchunks), dim3(256), 0, L, nl, Q, b, n,
                 rows_per_chunk, gpart);
      TPG_LAUNCH(ctx, "eig_gram_reduce", tpg_gram_reduce_kernel, dim3((unsigned)((nl * b + 63) / 64)), dim3(256), 0,
                 (const double*)gpart, nchunks, nl, b, (const double*)lam_dev, cdev);
      TPG_LAUNCH(ctx, "eig_right_mult", tpg_right_mult_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, L, n, nl,
                 (const double*)cdev, b, dtmp);
      Dm = dtmp;
    }
    TPG_LAUNCH(ctx, "eig_combine", tpg_combine_kernel, dim3(1024), dim3(256), 0, (const double*)part, S,
               (int64_t)n * b, alpha, Y1, beta, Y0, gamma, Y, Dm);
    TPG_CHECK_LAUNCH();
    return TPG_OK;
  }
  // C (p x bb, host, column-major) = A' B   (partials reduced on the device, p*bb doubles copied back)
  int gram(const double* A, int p, const double* B, int bb, std::vector<double>& C) {
    TPG_LAUNCH(ctx, "eig_gram_small", tpg_gram_small_kernel, dim3((unsigned)nchunks), dim3(256), 0, A, p, B, bb, n,
               rows_per_chunk, gpart);
    TPG_LAUNCH(ctx, "eig_gram_reduce", tpg_gram_reduce_kernel, dim3((unsigned)((p * bb + 63) / 64)), dim3(256), 0,
               (const double*)gpart, nchunks, p, bb, (const double*)nullptr, cdev);
    C.assign((size_t)p * bb, 0.0);
    TPG_HIP(tpg_fetch_small(ctx, C.data(), cdev, sizeof(double) * (size_t)p * bb));
    return TPG_OK;
  }
  // Ritz step: Y0 = A X, Y1 = Yk X, res2[j] = |Y1_j - theta_j Y0_j|^2 (host, after a synchronisation)
  int ritz(const double* A, const double* Yk, int p, const std::vector<double>& X, const std::vector<double>& theta,
           double* Y0, double* Y1, std::vector<double>& res2) {
    double* dx = xdev;
    TPG_HIP(tpg_push_small(ctx, dx, X.data(), sizeof(double) * (size_t)p * p));
    TPG_HIP(tpg_push_small(ctx, dx + 64 * 64, theta.data(), sizeof(double) * (size_t)p));
    const int nblk = (n + 31) / 32;
    TPG_LAUNCH(ctx, "eig_ritz", tpg_ritz_kernel, dim3((unsigned)nblk), dim3(256), 0, A, Yk, n, p, (const double*)dx,
               (const double*)(dx + 64 * 64), Y0, Y1, gpart);
    TPG_LAUNCH(ctx, "eig_gram_reduce", tpg_gram_reduce_kernel, dim3((unsigned)((p + 63) / 64)), dim3(256), 0,
               (const double*)gpart, nblk, p, 1, (const double*)nullptr, cdev);
    res2.assign((size_t)p, 0.0);
    TPG_HIP(tpg_fetch_small(ctx, res2.data(), cdev, sizeof(double) * (size_t)p));
    return TPG_OK;
  }
  // Y = A X  (X host p x b2)
  int rmult(const double* A, int p, const std::vector<double>& X, int b2, double* Y) {
    double* dx = xdev;
    TPG_HIP(tpg_push_small(ctx, dx, X.data(), sizeof(double) * (size_t)p * b2));
    TPG_LAUNCH(ctx, "eig_right_mult", tpg_right_mult_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, A, n, p,
               (const double*)dx, b2, Y);
    TPG_CHECK_LAUNCH();
    return TPG_OK;
  }
};

// Top-k eigenpairs of the symmetric PSD matrix d_K (n x n, device).  lambda[k] (descending), d_U n x k.
//
// Chebyshev-filtered subspace iteration with locking: a block of b = k + 12 vectors; every outer
// iteration does Rayleigh-Ritz on the active (not yet converged) columns, locks the leading Ritz pairs
// whose residual is below TOL * lambda_1, then applies a Chebyshev polynomial of K that damps
// [0, smallest active Ritz value].  The degree is capped so that the amplification spread inside the
// block stays below 1e7 (otherwise the trailing columns drown in rounding noise of the leading
// directions and the block loses rank); locking shrinks that spread as the large eigenvalues converge.
// TPG_DEBUG=1: wall-clock per stage (the stream is drained at every mark, so the marks perturb the run)
struct StageTimer {
  tpg_ctx* ctx; const char* tag; bool on; double t0;
  static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
  StageTimer(tpg_ctx* c, const char* t) : ctx(c), tag(t), on(getenv("TPG_DEBUG") != nullptr), t0(0) { if (on) { (void)hipStreamSynchronize(ctx->stream); t0 = now(); } }
  void mark(const char* what) {
    if (!on) return;
    (void)hipStreamSynchronize(ctx->stream);
    const double t1 = now();
    fprintf(stderr, "[%s] %-28s %8.3f ms\n", tag, what, t1 - t0);
    t0 = t1;
  }
};

static int eig_topk(tpg_ctx* ctx, const double* d_K, int n, int k, double* lambda_host, double* d_U, double tol = 1e-12) {
  // block = 2k + 12 (RSpectra, which the reference calls, keeps ncv = 2k + 1 Lanczos vectors): K Q is bound by
  // reading K, so extra columns are nearly free, and a block that reaches past the k wanted values into the
  // bulk of the spectrum converges in far fewer filter / Rayleigh-Ritz rounds than k + 12 columns do
  int b = 2 * k + 12;
  if (getenv("TPG_EIG_BLOCK")) b = atoi(getenv("TPG_EIG_BLOCK"));
  if (b > 64) b = 64;
  if (b > n) b = n;
  TPG_REQUIRE(k <= b, TPG_EINVAL, "k = %d too large (at most %d components)", k, b);
  EigWork w{ctx, d_K, n, b, 1};
  TPG_TRY(w.init());
  double *Q = nullptr, *Y = nullptr, *Y0 = nullptr, *Y1 = nullptr;
  const size_t nbytes = sizeof(double) * (size_t)n * (size_t)b;
  TPG_HIP(tpg_pmalloc((void**)&Q, nbytes));
  TPG_HIP(tpg_pmalloc((void**)&Y, nbytes));
  TPG_HIP(tpg_pmalloc((void**)&Y0, nbytes));
  TPG_HIP(tpg_pmalloc((void**)&Y1, nbytes));
  struct Free { double *a, *b, *c, *d; ~Free() { tpg_pfree(a); tpg_pfree(b); tpg_pfree(c); tpg_pfree(d); } } fr{Q, Y, Y0, Y1};
  auto colbytes = [&](int cols) { return sizeof(double) * (size_t)n * (size_t)cols; };
  auto axpby = [&](const double* x, double alpha, const double* y, double beta, double* out, int cols) -> int {
    TPG_LAUNCH(ctx, "eig_combine", tpg_combine_kernel, dim3(1024), dim3(256), 0, x, 1, (int64_t)n * cols, alpha, y,
               beta, (const double*)nullptr, 0.0, out, (const double*)nullptr);
    TPG_CHECK_LAUNCH();
    return TPG_OK;
  };

  // CholQR (twice) of the `act` columns at A, after projecting out the `nl` locked columns at L
  auto orthonormalize = [&](double* A, int act, const double* L, int nl, double* tmp, int passes = 2) -> int {
    int extra_passes = 0;
    for (int pass = 0; pass < passes; pass++) {
      if (nl > 0) {
        std::vector<double> Cm;
        TPG_TRY(w.gram(L, nl, A, act, Cm));      // nl x act
        TPG_TRY(w.rmult(L, nl, Cm, act, tmp));   // L (L'A)
        TPG_TRY(axpby(tmp, -1.0, A, 1.0, A, act));
      }
      std::vector<double> G, Ri;
      TPG_TRY(w.gram(A, act, A, act, G));
      std::vector<double> D((size_t)act);
      for (int j = 0; j < act; j++) D[(size_t)j] = G[j + (size_t)j * act] > 0 ? 1.0 / sqrt(G[j + (size_t)j * act]) : 1.0;
      for (int j = 0; j < act; j++)
        for (int i = 0; i < act; i++) G[i + (size_t)j * act] *= D[(size_t)i] * D[(size_t)j];
      std::vector<double> Gsave = G;
      if (host_cholesky_upper(G, act)) {
        host_upper_inverse(G, act, Ri);
      } else {
        // ill-conditioned block: whiten with the eigen-decomposition of the Gram matrix instead
        // (G = V diag(lam) V', A <- A V diag(lam^-1/2), tiny lam clamped); the next pass restores
        // orthogonality to rounding.
        std::vector<double> lamg, Vg;
        host_sym_eig(Gsave, act, lamg, Vg);  // tridiagonal QL: ~0.1 ms at 52 x 52 where cyclic Jacobi took over a millisecond
        if (getenv("TPG_DEBUG")) fprintf(stderr, "[eig] CholQR fell back to the eigen-decomposition of the block Gram matrix\n");
        const double floor_ = std::max(lamg[0], 1e-300) * 1e-14;
        Ri.assign((size_t)act * act, 0.0);
        for (int j = 0; j < act; j++) {
          const double sc = 1.0 / sqrt(std::max(lamg[(size_t)j], floor_));
          for (int i = 0; i < act; i++) Ri[i + (size_t)j * act] = Vg[i + (size_t)j * act] * sc;
        }
        if (pass == passes - 1) pass = passes - 2, extra_passes++;  // one more clean-up pass
        if (extra_passes > 3) { tpg_set_error("eigen solver: block lost rank"); return TPG_ENUMERIC; }
      }
      for (int j = 0; j < act; j++)
        for (int i = 0; i < act; i++) Ri[i + (size_t)j * act] *= D[(size_t)i];
      TPG_TRY(w.rmult(A, act, Ri, act, tmp));
      TPG_HIP(tpg_copy_dev(ctx, A, tmp, colbytes(act)));
    }
    return TPG_OK;
  };

  StageTimer st(ctx, "eig");
  TPG_LAUNCH(ctx, "eig_random", tpg_fill_random_kernel, dim3(512), dim3(256), 0, Q, (int64_t)n * b, (uint64_t)0x5EED);
  // uniform random columns are well conditioned: one CholQR pass leaves them orthonormal to ~1e-13, which is all the
  // first Rayleigh-Ritz step (spectral bounds for the filter) asks for
  TPG_TRY(orthonormalize(Q, b, nullptr, 0, Y, 1));
  st.mark("init + orthonormalize");
  std::vector<double> lam((size_t)b, 0.0), theta, X, H;
  int nl = 0;
  const int MAXIT = 200;
  const double TOL = tol, AMP = 1e6;  // a pair is accepted when its residual |K u - lambda u| <= TOL * lambda_1
  double lam1 = 0;
  for (int it = 0; it < MAXIT && nl < k; it++) {
    const int act = b - nl;
    double* A = Q + (size_t)n * nl;
    w.b = act;
    TPG_TRY(w.set_locked(Q, nl, lam.data()));
    // Rayleigh-Ritz on the active columns (deflated operator)
    TPG_TRY(w.apply(A, 1.0, nullptr, 0, nullptr, 0, Y));  // Y = K' A
    TPG_TRY(w.gram(A, act, Y, act, H));
    st.mark("apply + gram(H)");
    for (int i = 0; i < act; i++)
      for (int j = i + 1; j < act; j++) {
        const double sy = 0.5 * (H[i + (size_t)j * act] + H[j + (size_t)i * act]);
        H[i + (size_t)j * act] = sy;
        H[j + (size_t)i * act] = sy;
      }
    // The random start block has no converged pair and the filter works on any basis of the subspace: its first
    // Rayleigh-Ritz step only supplies the Ritz VALUES (the filter's interval), so the rotation to Ritz vectors and the
    // residuals (three products, a Gram matrix and a host round trip) are left out -- and so are the eigenvectors of H
    const bool first = it == 0 && b < n;  // (a block that spans the whole space is exact at once)
    host_sym_eig(H, act, theta, X, !first);
    st.mark("host_sym_eig");
    std::vector<double> RR;
    if (first) {
      lam1 = fabs(theta[0]) > 0 ? fabs(theta[0]) : 1.0;
    } else {
      TPG_TRY(w.ritz(A, Y, act, X, theta, Y0, Y1, RR));  // Ritz vectors, K * Ritz vectors, |K a - theta a|^2
      TPG_HIP(tpg_copy_dev(ctx, A, Y0, colbytes(act)));
      if (nl == 0) lam1 = fabs(theta[0]) > 0 ? fabs(theta[0]) : 1.0;
      st.mark("ritz vectors + residuals");
    }
    int newly = 0;
    while (!first && newly < act && nl + newly < k && sqrt(std::max(0.0, RR[(size_t)newly])) < TOL * lam1) newly++;
    for (int j = 0; j < newly; j++) lam[(size_t)(nl + j)] = theta[(size_t)j];
    nl += newly;
    if (nl >= k) break;
    const int act2 = b - nl;
    TPG_REQUIRE(act2 >= 2, TPG_ENUMERIC, "eigen solver ran out of active vectors");
    double* A2 = Q + (size_t)n * nl;
    const double* KA2 = first ? Y : Y1 + (size_t)n * newly;  // K * (active Ritz vectors), still valid
    // Chebyshev filter damping [0, smallest active Ritz value], scaled at the largest active one
    const double up = std::max(theta[(size_t)act - 1], 1e-300 * lam1), lo = 0.0;
    const double a0 = std::max(theta[(size_t)newly], up * (1 + 1e-8));
    const double ec = (up - lo) / 2, cc = (up + lo) / 2;
    const double x0 = (a0 - cc) / ec;
    int deg = 20;
    if (x0 > 1.0 + 1e-12) {
      const double dmax = log(2 * AMP) / acosh(x0);
      deg = dmax < 2 ? 2 : (dmax > 20 ? 20 : (int)dmax);
    }
    if (getenv("TPG_DEBUG"))
      fprintf(stderr, "[eig] it %d act %d locked %d (+%d) deg %d theta_first %.4g theta_last %.4g lam1 %.4g res0 %.3g\n", it,
              act, nl, newly, deg, theta[(size_t)newly], theta[(size_t)act - 1], lam1,
              first ? -1.0 : sqrt(std::max(0.0, RR[(size_t)newly])) / lam1);
    w.b = act2;
    TPG_TRY(w.set_locked(Q, nl, lam.data()));
    // KA2 was formed with the previous deflation; the newly locked directions are (numerically)
    // orthogonal to the remaining Ritz vectors, so K' A2 = K'_old A2 up to rounding.
    double sigma = ec / (a0 - cc);
    const double sigma1 = sigma;
    // first step: cur = (sigma1/e)(K A2 - c A2), prev = A2
    double* prev = Y0;
    double* cur = first ? Y1 : Y;  // never the buffer KA2 lives in
    double* nxt = first ? Y : Y1;
    TPG_HIP(tpg_copy_dev(ctx, prev, A2, colbytes(act2)));
    TPG_LAUNCH(ctx, "eig_combine", tpg_combine_kernel, dim3(1024), dim3(256), 0, KA2, 1, (int64_t)n * act2,
               sigma1 / ec, (const double*)prev, -cc * sigma1 / ec, (const double*)nullptr, 0.0, cur,
               (const double*)nullptr);
    for (int dgr = 2; dgr <= deg; dgr++) {
      const double sigma2 = 1.0 / (2.0 / sigma1 - sigma);
      TPG_TRY(w.apply(cur, 2 * sigma2 / ec, cur, -2 * sigma2 * cc / ec, prev, -sigma * sigma2, nxt));
      double* t = prev; prev = cur; cur = nxt; nxt = t;
      sigma = sigma2;
    }
    TPG_HIP(tpg_copy_dev(ctx, A2, cur, colbytes(act2)));
    st.mark("chebyshev filter");
    TPG_TRY(orthonormalize(A2, act2, Q, nl, prev));
    st.mark("orthonormalize");
  }
  TPG_REQUIRE(nl >= k, TPG_ENUMERIC, "eigen solver did not converge (%d of %d pairs)", nl, k);
  for (int j = 0; j < k; j++) lambda_host[j] = lam[(size_t)j];
  TPG_HIP(tpg_copy_dev(ctx, d_U, Q, colbytes(k)));  // (stream order: the scratch blocks go back to the pool after it)
  return TPG_OK;
}

static int pca_loadings_device(tpg_ctx* ctx, const tpg_view* v, const double* d_center, const double* d_scale,
                               const double* d_U, const double* d_dk, int k, double* d_V);

// ---------------------------------------------------------------------------
// More than 52 components (the reference's k is free, R/gt_pca_partialSVD.R:70-89): the block of the subspace iteration
// is at most 64 columns wide (LDS tiles of its kernels), so the spectrum is taken in batches of 26 pairs (block 2 * 26 +
// 12 = 64) with EXPLICIT deflation in between: K <- K - U_b diag(lambda_b) U_b' on a scratch copy of the matrix moves the
// converged eigenvalues to 0 +- eps * lambda_1 and leaves the other pairs alone, so the next batch converges to the next
// 26.  Later batches meet a stricter absolute tolerance (it is relative to THEIR largest eigenvalue).
__global__ __launch_bounds__(256) void tpg_deflate_kernel(double* __restrict__ K, int n, const double* __restrict__ U,
                                                          const double* __restrict__ lam, int kb) {
  __shared__ double ui[16][33], uj[16][33];  // [row in tile][component], kb <= 32
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
  for (int q = threadIdx.x; q < 16 * kb; q += 256) {
    const int r = q & 15, c = q >> 4;
    ui[r][c] = i0 + r < n ? U[(int64_t)(i0 + r) + (int64_t)c * n] * lam[c] : 0.0;
    uj[r][c] = j0 + r < n ? U[(int64_t)(j0 + r) + (int64_t)c * n] : 0.0;
  }
  __syncthreads();
  const int i = i0 + tx, j = j0 + ty;
  if (i >= n || j >= n) return;
  double s = 0.0;
  for (int c = 0; c < kb; c++) s = fma(ui[tx][c], uj[ty][c], s);
  K[(int64_t)i + (int64_t)j * n] -= s;
}

static int eig_topk_any(tpg_ctx* ctx, const double* d_K, int n, int k, double* lambda_host, double* d_U, double tol) {
  if (k <= 52) return eig_topk(ctx, d_K, n, k, lambda_host, d_U, tol);
  double *Kw = nullptr, *d_lam = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&Kw, sizeof(double) * (size_t)n * (size_t)n));
  struct Free { void *a, *b; ~Free() { tpg_pfree(a); tpg_pfree(b); } } fr{Kw, nullptr};
  TPG_HIP(tpg_pmalloc((void**)&d_lam, sizeof(double) * 32));
  fr.b = d_lam;
  TPG_HIP(hipMemcpyAsync(Kw, d_K, sizeof(double) * (size_t)n * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
  const int KB = 26;
  for (int done = 0; done < k;) {
    const int kb = std::min(KB, k - done);
    double* Ub = d_U + (size_t)done * (size_t)n;
    TPG_TRY(eig_topk(ctx, Kw, n, kb, lambda_host + done, Ub, tol));
    done += kb;
    if (done < k) {
      TPG_HIP(tpg_h2d_async(ctx, d_lam, lambda_host + done - kb, sizeof(double) * (size_t)kb));
      TPG_LAUNCH(ctx, "eig_deflate", tpg_deflate_kernel, dim3((unsigned)((n + 15) / 16), (unsigned)((n + 15) / 16)), dim3(256), 0, Kw, n,
                 (const double*)Ub, (const double*)d_lam, kb);
      TPG_CHECK_LAUNCH();
    }
  }
  return TPG_OK;
}

// ---------------------------------------------------------------------------
static int pca_svd_impl(tpg_ctx* ctx, tpg_comm* comm, const tpg_view* v, int k, double tol, double* d, double* u,
                        double* vload, double* center, double* scale, double* square_frobenius);

extern "C" int tpg_pca_partial_svd(tpg_ctx* ctx, const tpg_view* v, int k, double* d, double* u, double* vload,
                                   double* center, double* scale, double* square_frobenius) {
  TpgEnter _enter(ctx);
  return pca_svd_impl(ctx, nullptr, v, k, 1e-12, d, u, vload, center, scale, square_frobenius);
}

// The same with the loci sharded over the ranks of a communicator: `v` holds this rank's loci; center, scale and the
// loadings are per locus (this rank's rows of v come back); the Gram matrix and the squared Frobenius norm are
// additive over loci and summed over the ranks inside (one N x N all-reduce of doubles); the eigen step is replicated.
// One rank: identical to tpg_pca_partial_svd.
extern "C" int tpg_pca_partial_svd_sharded(tpg_ctx* ctx, tpg_comm* comm, const tpg_view* v, int k, double* d, double* u,
                                           double* vload, double* center, double* scale, double* square_frobenius) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(comm && comm->ctx == ctx, TPG_EINVAL, "bad communicator");
  return pca_svd_impl(ctx, comm, v, k, 1e-12, d, u, vload, center, scale, square_frobenius);
}

// gt_pca_randomSVD (R/gt_pca_randomSVD.R:77-135): the reference reaches the same truncated SVD through
// bigstatsr::big_randomSVD -> RSpectra::svds on the implicit operator (third-party, recalled), whose `tol` is the
// relative residual at which a singular triplet is accepted (default 1e-4).  Same Gram + subspace iteration here,
// stopped at that tolerance: residual |K u - d^2 u| <= tol * d_1^2 for every returned pair (the eigenvalue error is
// then of order tol^2, the vector error of order tol / gap) -- fewer iterations than the partialSVD path's 1e-12.
extern "C" int tpg_pca_random_svd(tpg_ctx* ctx, const tpg_view* v, int k, double tol, double* d, double* u,
                                  double* vload, double* center, double* scale, double* square_frobenius) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(tol > 0 && tol < 1, TPG_EINVAL, "tol = %g out of (0, 1)", tol);
  return pca_svd_impl(ctx, nullptr, v, k, tol < 1e-12 ? 1e-12 : tol, d, u, vload, center, scale, square_frobenius);
}

// upper triangle (column k holds rows 0 .. k at k (k + 1) / 2) <-> full symmetric matrix
__global__ void tpg_tri_pack_kernel(const double* __restrict__ K, int n, double* __restrict__ tri) {
  const int64_t total = (int64_t)n * n;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx % n, k = idx / n;
    if (i <= k) tri[k * (k + 1) / 2 + i] = K[idx];
  }
}
__global__ void tpg_tri_unpack_kernel(const double* __restrict__ tri, int n, double* __restrict__ K) {
  const int64_t total = (int64_t)n * n;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx % n, k = idx / n;
    const int64_t lo = i < k ? i : k, hi = i < k ? k : i;
    K[idx] = tri[hi * (hi + 1) / 2 + lo];
  }
}

// K (n x n in device memory, bitwise symmetric) <- its sum over the ranks; only the upper triangle travels.  For callers that
// accumulate the Gram matrix themselves (stream.hip: block by block); the allocation is agreed on before the collective.
int tpg_pca_gram_allreduce(tpg_ctx* ctx, tpg_comm* comm, double* d_K, int64_t n) {
  if (!comm || !(comm->nranks > 1 || comm->nccl)) return TPG_OK;
  const int64_t ntri = n * (n + 1) / 2;
  double* d_tri = nullptr;
  hipError_t e = tpg_pmalloc((void**)&d_tri, sizeof(double) * (size_t)ntri);
  int lrc = TPG_OK;
  if (e != hipSuccess) { tpg_set_error("gram all-reduce: %s", hipGetErrorString(e)); (void)hipGetLastError(); lrc = TPG_EHIP; }
  lrc = tpg_comm_agree(comm, lrc);
  if (lrc != TPG_OK) { tpg_pfree(d_tri); return lrc; }
  TPG_LAUNCH(ctx, "pca_gram_tri", tpg_tri_pack_kernel, dim3(2048), dim3(256), 0, (const double*)d_K, (int)n, d_tri);
  int rc;
  {
    ProfScope ps(ctx, "pca_gram_allreduce");
    rc = tpg_comm_allreduce(comm, d_tri, ntri, 1);
  }
  if (rc == TPG_OK) TPG_LAUNCH(ctx, "pca_gram_tri", tpg_tri_unpack_kernel, dim3(2048), dim3(256), 0, (const double*)d_tri, (int)n, d_K);
  tpg_pfree(d_tri);
  return rc;
}

static int pca_svd_impl(tpg_ctx* ctx, tpg_comm* comm, const tpg_view* v, int k, double tol, double* d, double* u,
                        double* vload, double* center, double* scale, double* square_frobenius) {
  TPG_REQUIRE(ctx && v && d && u && vload && center && scale, TPG_EINVAL, "null argument");
  TPG_REQUIRE(k >= 1 && k <= v->n, TPG_EINVAL, "k = %d out of range", k);
  const int64_t n = v->n, m = v->m;
  OutBuf oc, os, ou, ov;
  int32_t* d_counts = nullptr;
  double *d_K = nullptr, *d_dk = nullptr, *d_tri = nullptr;
  struct Free { void *a, *b, *c, *d; ~Free() { tpg_pfree(a); tpg_pfree(b); tpg_pfree(c); tpg_pfree(d); } } fr{nullptr, nullptr, nullptr, nullptr};
  StageTimer st(ctx, "svd");
  const bool exchange = comm && (comm->nranks > 1 || comm->nccl);
  const int64_t ntri = n * (n + 1) / 2;
  // Everything that can fail on THIS rank alone (allocations; a missing value or a zero scale among this shard's loci)
  // comes first, then the ranks agree on a status: a rank that gave up here would otherwise leave the others waiting in
  // the all-reduces below for ever.
  auto local_steps = [&]() -> int {
    TPG_REQUIRE(k <= m, TPG_EINVAL, "k = %d but this view has %lld loci", k, (long long)m);
    TPG_TRY(oc.init(center, sizeof(double) * (size_t)m));
    TPG_TRY(os.init(scale, sizeof(double) * (size_t)m));
    TPG_TRY(ou.init(u, sizeof(double) * (size_t)n * (size_t)k));
    TPG_TRY(ov.init(vload, sizeof(double) * (size_t)m * (size_t)k));
    TPG_HIP(tpg_pmalloc((void**)&d_counts, sizeof(int32_t) * 4 * (size_t)m));
    fr.a = d_counts;
    TPG_HIP(tpg_pmalloc((void**)&d_K, sizeof(double) * (size_t)n * (size_t)n));
    fr.b = d_K;
    TPG_HIP(tpg_pmalloc((void**)&d_dk, sizeof(double) * (size_t)k));
    fr.d = d_dk;
    if (exchange) {
      TPG_HIP(tpg_pmalloc((void**)&d_tri, sizeof(double) * (size_t)ntri));
      fr.c = d_tri;
    }
    TPG_TRY(pca_counts_center_scale(ctx, v, d_counts, oc.dev<double>(), os.dev<double>()));
    st.mark("counts, center, scale");
    if (square_frobenius) TPG_TRY(frobenius_from_counts(ctx, v, d_counts, oc.dev<double>(), os.dev<double>(), square_frobenius));
    return TPG_OK;
  };
  int lrc = local_steps();
  if (exchange) lrc = tpg_comm_agree(comm, lrc);
  TPG_TRY(lrc);
  if (square_frobenius && exchange) TPG_TRY(tpg_comm_allreduce_f64(ctx, comm, square_frobenius, 1));
  st.mark("frobenius + alloc K");
  // Several ranks: every rank takes whole weight classes (the packed columns are exchanged once, gramcls.hip) when that
  // pays -- decided alike on every rank --, otherwise the Gram matrix of its own loci.
  // The Gram branch CAN fail on one rank alone (after the exchange a rank holds the loci of its classes, a different
  // number on every rank: its allocations, the 2^31 limit of the class path, the pool) and what follows is an all-reduce
  // RCCL never times out of: one status for the whole branch, agreed on before the triangle travels.  (Inside
  // tpg_gram_classes_exchanged the ranks agree before each of ITS collectives; a failure there comes back alike everywhere.)
  auto gram_branch = [&]() -> int {
    bool exchanged = false;
    if (exchange) TPG_TRY(tpg_gram_classes_exchanged(ctx, comm, v, d_counts, os.dev<double>(), d_K, &exchanged));
    if (exchanged) return pca_double_center_inplace(ctx, d_K, n);
    return pca_gram_device(ctx, v, oc.dev<double>(), os.dev<double>(), d_K, true);
  };
  int grc = gram_branch();
  if (exchange) grc = tpg_comm_agree(comm, grc);
  TPG_TRY(grc);
  st.mark("gram");
  if (exchange) {  // K = sum over the ranks' loci of z_j z_j'
    // only the upper triangle travels: n (n + 1) / 2 doubles instead of n^2 (K is symmetric bit for bit on every rank)
    TPG_LAUNCH(ctx, "pca_gram_tri", tpg_tri_pack_kernel, dim3(2048), dim3(256), 0, (const double*)d_K, (int)n, d_tri);
    {
      ProfScope ps(ctx, "pca_gram_allreduce");
      TPG_TRY(tpg_comm_allreduce(comm, d_tri, ntri, 1));
    }
    TPG_LAUNCH(ctx, "pca_gram_tri", tpg_tri_unpack_kernel, dim3(2048), dim3(256), 0, (const double*)d_tri, (int)n, d_K);
  }
  std::vector<double> lam((size_t)k);
  TPG_TRY(eig_topk_any(ctx, d_K, (int)n, k, lam.data(), ou.dev<double>(), tol));
  st.mark("eig_topk");
  std::vector<double> dh((size_t)k);
  for (int j = 0; j < k; j++) dh[(size_t)j] = sqrt(lam[(size_t)j] > 0 ? lam[(size_t)j] : 0.0);
  TPG_HIP(tpg_push_small(ctx, d_dk, dh.data(), sizeof(double) * (size_t)k));
  // v = Z'u / d  (no missing values: checked by pca_counts_center_scale above)
  TPG_TRY(pca_loadings_device(ctx, v, oc.dev<double>(), os.dev<double>(), ou.dev<double>(), d_dk, k, ov.dev<double>()));
  if (tpg_is_device_ptr(d)) TPG_HIP(tpg_push_small(ctx, d, dh.data(), sizeof(double) * (size_t)k));
  else memcpy(d, dh.data(), sizeof(double) * (size_t)k);
  TPG_HIP(hipStreamSynchronize(ctx->stream));
  st.mark("loadings");
  TPG_TRY(oc.commit(ctx));
  TPG_TRY(os.commit(ctx));
  TPG_TRY(ou.commit(ctx));
  return ov.commit(ctx);
}

// ---------------------------------------------------------------------------
// Loadings v = Z'u/d on int8 MFMA.  With no missing values,
//     v_jk = ( sum_i g_ij u_ik  -  c_j sum_i u_ik ) / (s_j d_k),
// and sum_i g_ij u_ik is a contraction over individuals of the dosage plane with u.  u is rounded to
// fixed point (FU fractional bits, relative 2^-40 of max|u|) and split into TU = 6 balanced base-128 digits;
// column kk*TU + t of the B operand holds digit t of u[:, kk], so k = 20 components need 120 int8 columns
// = 4 MFMA column tiles.  Exact int32 accumulation; digits are recombined in int64 and scaled once.
#define LD_TU 6

__global__ void tpg_u_digits_kernel(const double* __restrict__ U, int64_t n, int k, int FU, int64_t Q, int CT,
                                    uint4* __restrict__ UD) {
  const int64_t total = Q * 4 * CT * 64;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    const int64_t tq = idx >> 6;
    const int ct = (int)(tq % CT);
    const int64_t qs = tq / CT;
    const int s = (int)(qs & 3);
    const int64_t q = qs >> 2;
    const int col = 32 * ct + (lane & 31), h = lane >> 5;
    const int kk = col / LD_TU, t = col % LD_TU;
    uint32_t w[4] = {0, 0, 0, 0};
    if (kk < k)
      for (int e = 0; e < 16; e++) {
        const int64_t i = 128 * q + 32 * s + 16 * h + e;
        if (i >= n) continue;
        long long W = llrint(ldexp(U[i + (int64_t)kk * n], FU));
        long long dig = 0;
        for (int tt = 0; tt <= t; tt++) {
          dig = W & 127;
          if (dig >= 64) dig -= 128;
          W = (W - dig) >> 7;
        }
        w[e >> 2] |= (uint32_t)((int)dig & 0xFF) << (8 * (e & 3));
      }
    UD[idx] = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// usum[kk] = sum_i round(u_ik 2^FU) 2^-FU  (the same rounded values the MFMA contraction sees)
__global__ __launch_bounds__(256) void tpg_uq_colsum_kernel(const double* __restrict__ U, int64_t n, int FU,
                                                            double* __restrict__ usum) {
  __shared__ double sh[256];
  const int kk = blockIdx.x;
  double acc = 0;
  for (int64_t i = threadIdx.x; i < n; i += 256) acc += ldexp((double)llrint(ldexp(U[i + (int64_t)kk * n], FU)), -FU);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) usum[kk] = sh[0];
}

// out[locus][column] = sum_i g_i,locus * UD[i][column].  One wave owns LD_NLT consecutive 32-locus tiles and CTP
// column tiles (LD_NLT * CTP accumulator tiles, up to 128 AGPRs, two workgroups per CU: with four locus tiles per wave
// and one resident workgroup its prologue, barriers and flush were exposed, 0.70 against 0.59 ms), and the four waves of a
// workgroup share the digit fragments of a 128-individual group through double-buffered LDS (each wave fetches a
// quarter, one barrier per group): a 1-KiB fragment of UD fetched from L2 feeds 4 * LD_NLT MFMAs.  With one locus
// tile per wave and a fragment per MFMA straight from L2 the kernel moved 20 GB per launch at C5 through the L1s
// and was bound by that (1.6 ms).  The A side is the code bytes themselves (no missing values here, see the Gram
// kernel).
#ifndef LD_NLT
#define LD_NLT 2
#endif
#ifndef LD_WGS
#define LD_WGS 2  // workgroups per CU the register budget is cut for
#endif
__device__ __forceinline__ int64_t tpg_uniform64_pca(int64_t x) {  // a wave-uniform value the compiler keeps in SGPRs
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)x >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <int C, class F>
__device__ __forceinline__ void tpg_pca_static_for(F&& f) {
  if constexpr (C > 0) {
    tpg_pca_static_for<C - 1>(f);
    f(std::integral_constant<int, C - 1>{});
  }
}

template <int CTP>
__global__ __launch_bounds__(256, LD_WGS) void tpg_loadings_mfma_kernel(const uint4* __restrict__ L,
                                                                   const uint4* __restrict__ UD, int64_t n_lt,
                                                                   int64_t Q, int ct0, int CT,
                                                                   int32_t* __restrict__ out, int Cpad) {
  __shared__ __attribute__((aligned(16))) uint4 ubuf[2][4 * CTP][64];  // [buffer][K step * CTP + column tile][lane]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t lt0 = ((int64_t)blockIdx.x * 4 + wv) * LD_NLT;  // may lie past n_lt: the wave still serves the LDS
  v16i acc[LD_NLT][CTP];
#pragma unroll
  for (int t = 0; t < LD_NLT; t++)
#pragma unroll
    for (int c = 0; c < CTP; c++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[t][c][r] = 0;
  // (wave-uniform bases in SGPRs + a 32-bit lane offset per group, 32-bit group indices: no 64-bit per-lane pointers)
  const int Qi = (int)Q;
  const int64_t lt0u = tpg_uniform64_pca(lt0);
  const char* pa[LD_NLT];
#pragma unroll
  for (int t = 0; t < LD_NLT; t++) pa[t] = (const char*)(L + ((lt0u + t < n_lt ? lt0u + t : 0) * Q) * 64);  // past the end: a copy
  auto LDG = [&](const char* p, int blk) {
    uint32_t off = (uint32_t)lane * 16u + (uint32_t)blk * 1024u;
    asm("" : "+v"(off));
    return *(const uint4*)(p + off);
  };
  // The genotype stream goes through LD_D rotating register slots, fetched LD_D - 1 groups ahead, and the digit fragments of
  // group q' wait in slot q' & 1, fetched TWO groups ahead of the barrier behind which they are used; the loop is unrolled by
  // LD_D so that every slot is a compile-time index (copying a register that a load has just been issued into -- the
  // a = a1, a1 = a2 of the first form of this loop -- makes the wave wait for that load: SQ_WAIT_ANY 52 %).
  constexpr int LD_D = 4;
  static_assert(LD_D % 2 == 0, "the slot of a group's digit fragments is its parity");
  uint4 AR[LD_D][LD_NLT];
#pragma unroll
  for (int d = 0; d < LD_D - 1; d++)
#pragma unroll
    for (int t = 0; t < LD_NLT; t++) AR[d][t] = LDG(pa[t], d < Qi ? d : Qi - 1);
  // this wave's share of a group's fragments: items wv, wv + 4, ... of the 4 * CTP (K step, column tile) pairs
  const char* pu = (const char*)(UD + ct0 * 64);  // fragment (ks, c) = block ks * CT + c
  uint4 un[2][CTP];
  auto ufrag = [&](int q, int j) {
    const int it = wv + 4 * j;
    return LDG(pu, (q * 4 + it / CTP) * CT + it % CTP);
  };
#pragma unroll
  for (int j = 0; j < CTP; j++) {
    ubuf[0][wv + 4 * j][lane] = ufrag(0, j);
    un[1][j] = ufrag(Qi > 1 ? 1 : 0, j);
  }
  tpg_lds_barrier();
  auto group = [&](auto Cc, auto Mm, int q) {
    constexpr int C = decltype(Cc)::value, M = decltype(Mm)::value;
    const int qa = q + LD_D - 1 < Qi ? q + LD_D - 1 : Qi - 1, qu = q + 2 < Qi ? q + 2 : Qi - 1;
    const int cur = q & 1;
#pragma unroll
    for (int t = 0; t < LD_NLT; t++) AR[M][t] = LDG(pa[t], qa);
#pragma unroll
    for (int j = 0; j < CTP; j++) un[C & 1][j] = ufrag(qu, j);
#pragma unroll
    for (int s = 0; s < 4; s++) {
      v4i fg[LD_NLT];
#pragma unroll
      for (int t = 0; t < LD_NLT; t++) {
        const uint32_t w = s == 0 ? AR[C][t].x : s == 1 ? AR[C][t].y : s == 2 ? AR[C][t].z : AR[C][t].w;
#pragma unroll
        for (int k = 0; k < 4; k++) fg[t][k] = (int)tpg_codes(w, k);
      }
#pragma unroll
      for (int c = 0; c < CTP; c++) {
        const uint4 bq = ubuf[cur][s * CTP + c][lane];
        v4i fb = {(int)bq.x, (int)bq.y, (int)bq.z, (int)bq.w};
#pragma unroll
        for (int t = 0; t < LD_NLT; t++) acc[t][c] = MFMA_I8(fg[t], fb, acc[t][c]);
      }
    }
    // the other buffer was last read in group q - 1, which every wave left through the barrier below
#pragma unroll
    for (int j = 0; j < CTP; j++) ubuf[cur ^ 1][wv + 4 * j][lane] = un[(C & 1) ^ 1][j];
    tpg_lds_barrier();
  };
  for (int q = 0; q < Qi; q += LD_D)  // Q is the same for every wave: all of them meet every barrier
    tpg_pca_static_for<LD_D>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      if (q + k < Qi) group(std::integral_constant<int, k>{}, std::integral_constant<int, (k + LD_D - 1) % LD_D>{}, q + k);
    });
#pragma unroll
  for (int t = 0; t < LD_NLT; t++) {
    if (lt0 + t >= n_lt) break;
#pragma unroll
    for (int c = 0; c < CTP; c++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int64_t row = (lt0 + t) * 32 + tpg_cd_row(r, lane);
        out[row * Cpad + 32 * (ct0 + c) + (lane & 31)] = acc[t][c][r];
      }
  }
}

__global__ void tpg_loadings_finalize_kernel(const int32_t* __restrict__ acc, int Cpad, int64_t m, int k, int FU,
                                             const double* __restrict__ center, const double* __restrict__ scale,
                                             const double* __restrict__ usum, const double* __restrict__ d,
                                             double* __restrict__ vload) {
  const int64_t total = m * k;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(idx % k);  // threads of a wave read one locus row of `acc` contiguously
    const int64_t j = idx / k;
    long long S = 0;
#pragma unroll
    for (int t = LD_TU - 1; t >= 0; t--) S = S * 128 + (long long)acc[j * Cpad + kk * LD_TU + t];
    const double gu = ldexp((double)S, -FU);
    vload[j + (int64_t)kk * m] = (gu - center[j] * usum[kk]) / (scale[j] * d[kk]);
  }
}

// out[0] = max |x| (NaN if any element is NaN); one workgroup
__global__ __launch_bounds__(1024) void tpg_absmax_kernel(const double* __restrict__ x, int64_t total, double* __restrict__ out) {
  __shared__ double sh[1024];
  double mx = 0;
  bool bad = false;
  for (int64_t i = threadIdx.x; i < total; i += 1024) {
    const double a = fabs(x[i]);
    if (a != a) bad = true;
    mx = a > mx ? a : mx;
  }
  sh[threadIdx.x] = bad ? __longlong_as_double(0x7FF8000000000000ll) : mx;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      const double a = sh[threadIdx.x], b = sh[threadIdx.x + w];
      sh[threadIdx.x] = (a != a || b != b) ? __longlong_as_double(0x7FF8000000000000ll) : (a > b ? a : b);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

// device pointers throughout; requires a view without missing values (checked by the callers' counts)
static int pca_loadings_device(tpg_ctx* ctx, const tpg_view* v, const double* d_center, const double* d_scale,
                               const double* d_U, const double* d_dk, int k, double* d_V) {
  const int64_t n = v->n, m = v->m;
  // the scale of the digits: max |u|, found on the device (8 bytes come back, not the n x k matrix)
  double amax = 0;
  {
    double* d_amax = nullptr;
    TPG_HIP(tpg_pmalloc((void**)&d_amax, sizeof(double)));
    TPG_LAUNCH(ctx, "loadings_u_digits", tpg_absmax_kernel, dim3(1), dim3(1024), 0, d_U, n * (int64_t)k, d_amax);
    hipError_t ea = tpg_fetch_small(ctx, &amax, d_amax, sizeof(double));
    tpg_pfree(d_amax);
    TPG_HIP(ea);
  }
  TPG_REQUIRE(amax > 0 && amax == amax && amax < 1e300, TPG_ENUMERIC, "degenerate eigenvectors");
  int ex = 0;
  frexp(amax, &ex);                      // amax < 2^ex
  const int FU = 7 * LD_TU - 2 - ex;     // |u| 2^FU < 2^(7 TU - 2), inside the balanced digit range
  const int CT = (int)ceil_div((int64_t)k * LD_TU, 32);
  const int Cpad = CT * 32;
  const int64_t n_lt = v->KG * 4;
  uint4* d_UD = nullptr;
  int32_t* d_acc = nullptr;
  double* d_usum = nullptr;
  hipError_t e = tpg_pmalloc((void**)&d_UD, (size_t)v->Q * 4 * CT * 1024);
  if (e == hipSuccess) e = tpg_pmalloc((void**)&d_acc, sizeof(int32_t) * (size_t)n_lt * 32 * (size_t)Cpad);
  if (e == hipSuccess) e = tpg_pmalloc((void**)&d_usum, sizeof(double) * (size_t)k);
  if (e == hipSuccess) {
    TPG_LAUNCH(ctx, "loadings_u_digits", tpg_u_digits_kernel, dim3(1024), dim3(256), 0, d_U, n, k, FU, v->Q, CT, d_UD);
    TPG_LAUNCH(ctx, "loadings_u_digits", tpg_uq_colsum_kernel, dim3((unsigned)k), dim3(256), 0, d_U, n, FU, d_usum);
    const unsigned grid = (unsigned)ceil_div(n_lt, 4 * LD_NLT);
    for (int ct0 = 0; ct0 < CT;) {
      const int left = CT - ct0;
#define LD_LAUNCH(C)                                                                                             \
  TPG_LAUNCH(ctx, "loadings_mfma", tpg_loadings_mfma_kernel<C>, dim3(grid), dim3(256), 0, (const uint4*)v->L,     \
             (const uint4*)d_UD, n_lt, v->Q, ct0, CT, d_acc, Cpad)
      if (left >= 4) { LD_LAUNCH(4); ct0 += 4; }
      else if (left == 3) { LD_LAUNCH(3); ct0 += 3; }
      else if (left == 2) { LD_LAUNCH(2); ct0 += 2; }
      else { LD_LAUNCH(1); ct0 += 1; }
#undef LD_LAUNCH
    }
    TPG_LAUNCH(ctx, "loadings_finalize", tpg_loadings_finalize_kernel, dim3(2048), dim3(256), 0, (const int32_t*)d_acc,
               Cpad, m, k, FU, d_center, d_scale, (const double*)d_usum, d_dk, d_V);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  tpg_pfree(d_UD); tpg_pfree(d_acc); tpg_pfree(d_usum);
  if (e != hipSuccess) { tpg_set_error("loadings: %s", hipGetErrorString(e)); return TPG_EHIP; }
  return TPG_OK;
}

int tpg_sym_eig_topk_tol(tpg_ctx* ctx, const double* d_K, int64_t n, int k, double tol, double* lambda_host, double* d_U) {
  TPG_REQUIRE(ctx && d_K && lambda_host && d_U, TPG_EINVAL, "null argument");
  TPG_REQUIRE(n > 0 && k >= 1 && k <= n, TPG_EINVAL, "bad n = %lld / k = %d", (long long)n, k);
  return eig_topk_any(ctx, d_K, (int)n, k, lambda_host, d_U, tol);
}

extern "C" int tpg_sym_eig_topk(tpg_ctx* ctx, const double* K, int64_t n, int k, double* lambda, double* U) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && K && lambda && U, TPG_EINVAL, "null argument");
  TPG_REQUIRE(n > 0 && k >= 1 && k <= n, TPG_EINVAL, "bad n = %lld / k = %d", (long long)n, k);
  InBuf ik;
  TPG_TRY(ik.init(ctx, K, sizeof(double) * (size_t)n * (size_t)n));
  OutBuf ou;
  TPG_TRY(ou.init(U, sizeof(double) * (size_t)n * (size_t)k));
  std::vector<double> lam((size_t)k);
  TPG_TRY(eig_topk_any(ctx, ik.dev<double>(), (int)n, k, lam.data(), ou.dev<double>(), 1e-12));
  if (tpg_is_device_ptr(lambda)) TPG_HIP(hipMemcpyAsync(lambda, lam.data(), sizeof(double) * (size_t)k, hipMemcpyHostToDevice, ctx->stream));
  else memcpy(lambda, lam.data(), sizeof(double) * (size_t)k);
  TPG_HIP(hipStreamSynchronize(ctx->stream));
  return ou.commit(ctx);
}

extern "C" int tpg_pca_loadings(tpg_ctx* ctx, const tpg_view* v, const double* center, const double* scale,
                                const double* U, const double* d, int k, double* vload) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && center && scale && U && d && vload, TPG_EINVAL, "null argument");
  TPG_REQUIRE(k >= 1, TPG_EINVAL, "k must be positive");
  const int64_t n = v->n, m = v->m;
  InBuf ic, is, iu, id;
  TPG_TRY(ic.init(ctx, center, sizeof(double) * (size_t)m));
  TPG_TRY(is.init(ctx, scale, sizeof(double) * (size_t)m));
  TPG_TRY(iu.init(ctx, U, sizeof(double) * (size_t)n * (size_t)k));
  TPG_TRY(id.init(ctx, d, sizeof(double) * (size_t)k));
  OutBuf ov;
  TPG_TRY(ov.init(vload, sizeof(double) * (size_t)m * (size_t)k));
  // missing values? (then z = 0 there and the FP64 sweep handles it; big_SVD itself never gets here)
  int32_t* d_counts = nullptr;
  int* d_flag = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_counts, sizeof(int32_t) * 4 * (size_t)m));
  hipError_t e = tpg_pmalloc((void**)&d_flag, 2 * sizeof(int));
  int flag[2] = {0, 1};
  int rc = e == hipSuccess ? tpg_launch_loci_counts(ctx, v, d_counts) : TPG_EHIP;
  if (rc == TPG_OK) {
    e = hipMemsetAsync(d_flag, 0, 2 * sizeof(int), ctx->stream);
    TPG_LAUNCH(ctx, "pca_center_check", tpg_center_is_mean_kernel, dim3(1024), dim3(256), 0, (const int4*)d_counts,
               ic.dev<double>(), m, n, d_flag);
    if (e == hipSuccess) e = tpg_fetch_small(ctx, flag, d_flag, 2 * sizeof(int));
    if (e != hipSuccess) { tpg_set_error("pca_loadings: %s", hipGetErrorString(e)); rc = TPG_EHIP; }
  }
  tpg_pfree(d_counts);
  tpg_pfree(d_flag);
  TPG_TRY(rc);
  if (!flag[1]) {
    rc = pca_loadings_device(ctx, v, ic.dev<double>(), is.dev<double>(), iu.dev<double>(), id.dev<double>(), k,
                             ov.dev<double>());
  } else {
    double* d_inv = nullptr;
    TPG_HIP(tpg_pmalloc((void**)&d_inv, sizeof(double) * (size_t)m));
    TPG_LAUNCH(ctx, "inv_scale", tpg_inv_kernel, dim3(1024), dim3(256), 0, is.dev<double>(), m, d_inv);
    rc = run_sweep(ctx, SW_ROWSCALE, v->L, v->KG * 4, v->Q, m, n, ic.dev<double>(), d_inv, iu.dev<double>(), n, k,
                   ov.dev<double>(), id.dev<double>(), nullptr);
    tpg_pfree(d_inv);
  }
  TPG_TRY(rc);
  return ov.commit(ctx);
}
