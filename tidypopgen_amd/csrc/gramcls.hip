// gramcls.hip -- the weighted cross-product S' = G W G' of the PCA Gram matrix (pca.hip) by WEIGHT CLASSES.
//
// Replaces the FP64 crossprod of bigstatsr::big_SVD (third-party, recalled; called from R/gt_pca_partialSVD.R:76-84)
// on the scaled genotype matrix: K = Z Z', z_ij = (g_ij - c_j) / s_j, which pca.hip writes as
// K = S' - r 1' - 1 r' + C with S'_ik = sum_j w_j g_ij g_kj, w_j = 1 / s_j^2.
//
// The weights of a genotype panel take few distinct values: under the binomial scaling s_j^2 = 2 p_j (1 - p_j) with
// p_j = (alternate allele count) / 2N there are at most 2N + 1 of them however many loci there are.  So
//     S' = sum_c w_c G_c,        G_c = sum over the loci j of class c of g_i g_k'   (an INTEGER matrix),
// and G_c is a plain unweighted contraction: one FP4 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, dosages 0 / 1 / 2 are
// FP4 values, products and their FP32 sums exact below 2^24) per 32 x 32 pairs and 64 loci, where the digit-split
// int8 kernel of pca.hip needs four int8 MFMAs per 32 loci to carry 28 bits of fixed-point weight.  The price is a
// FOLD of the accumulators, out64 += w_c * (double)acc, whenever the class changes: 16 v_cvt_f64_f32 + 16 v_fma_f64
// per 32 x 32 tile and class (tools/ubench_fold.hip: ~130 ns against ~17 ns per MFMA), which pays as long as classes
// are not much shorter than a few dozen loci; tpg_gram_classes says "not done" otherwise and the caller runs the
// digit kernel.  The result is MORE accurate than the digit kernel's: class Gram matrices are exact, the weights are
// doubles (identified up to 2^-47 relative, so that c and 2N - c share a class) and the sums are FP64.
//
// Steps, all on the device:
//   1. key_j = bit pattern of w_j with the last five mantissa bits rounded away; radix sort of (key, locus);
//      run-length encoding -> classes; every class is padded to whole 64-locus blocks (zero dosage contributes
//      nothing); per block its weight and an "accumulators must be folded after this block" flag (end of class, or
//      2^16 blocks = 2^22 loci since the last fold: 4 x 2^22 is the largest FP32 sum that is still exact).
//   2. tpg_gcls_gather_kernel: the class-sorted operand layout T4g -- block (rt, b) = 32 individuals x the 64 loci
//      of sorted block b, one FP4 nibble per dosage (0, 1.0, 2.0) -- gathered from the view's L layout.
//   3. tpg_gcls_gram_kernel: one wave = a 64 x 96 tile of pairs (2 x 3 accumulator tiles) over a range of blocks.
//      Operands go from the loads (six blocks ahead) straight into the MFMAs: no decode instructions at all.
//      K split S: each (unit, split) writes its own FP64 slab with plain stores.
//   4. tpg_gcls_assemble_kernel: the S slabs of a unit are added in a fixed order (run-to-run identical results)
//      and written to both triangles of the n x n matrix.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "common.h"
#include "devfrag.h"

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

constexpr int GA = 2, GB = 3, GP = GA * GB;  // row tiles of the A side, of the B side, accumulator tiles per wave
#define GCLS_SLAB (GP * 16 * 64)             // doubles per (unit, split)
#define GCLS_MAX_RUN 65536                   // blocks between two folds: 4 * 64 * 65536 = 2^24

// ---------------------------------------------------------------------------
// 1. classes
__global__ void tpg_gcls_keys_kernel(const double* __restrict__ w, int64_t m, unsigned long long* __restrict__ key,
                                     uint32_t* __restrict__ idx) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(w[j]);
    key[j] = (b + 16ull) & ~31ull;  // positive doubles order like their bit patterns; 2^-47 relative
    idx[j] = (uint32_t)j;
  }
}

// blocks per run; entries past the number of runs are zero so that the scans can run over m entries
__global__ void tpg_gcls_run_blocks_kernel(const uint32_t* __restrict__ counts, const int* __restrict__ nruns, int64_t m,
                                           uint32_t* __restrict__ cnt_out, uint32_t* __restrict__ nblk) {
  const int nr = nruns[0];
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < m; r += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t c = r < nr ? counts[r] : 0u;
    cnt_out[r] = c;
    nblk[r] = (c + 63u) / 64u;
  }
}

__global__ void tpg_gcls_totals_kernel(const int* __restrict__ nruns, const uint32_t* __restrict__ blk_start,
                                       const uint32_t* __restrict__ nblk, long long* __restrict__ totals) {
  const int nr = nruns[0];
  totals[0] = nr;
  totals[1] = nr > 0 ? (long long)blk_start[nr - 1] + nblk[nr - 1] : 0;
}

__device__ __forceinline__ int tpg_gcls_find_run(const uint32_t* __restrict__ start, int nr, uint32_t x) {
  int lo = 0, hi = nr - 1;  // last r with start[r] <= x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (start[mid] <= x) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// sorted element i -> its slot in the padded layout; the class weight goes back to the locus (what)
__global__ void tpg_gcls_place_kernel(const uint32_t* __restrict__ sorted_idx, const unsigned long long* __restrict__ ukeys,
                                      const uint32_t* __restrict__ elem_start, const uint32_t* __restrict__ blk_start,
                                      int nr, int64_t m, int32_t* __restrict__ src, double* __restrict__ what) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = tpg_gcls_find_run(elem_start, nr, (uint32_t)i);
    const uint32_t j = sorted_idx[i];
    src[(int64_t)blk_start[r] * 64 + (i - elem_start[r])] = (int32_t)j;
    if (what) what[j] = __longlong_as_double((long long)ukeys[r]);
  }
}

__global__ void tpg_gcls_block_table_kernel(const unsigned long long* __restrict__ ukeys, const uint32_t* __restrict__ blk_start,
                                            const uint32_t* __restrict__ nblk, int nr, int64_t nblocks,
                                            unsigned long long* __restrict__ wblk) {
  for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < nblocks; b += (int64_t)gridDim.x * blockDim.x) {
    const int r = tpg_gcls_find_run(blk_start, nr, (uint32_t)b);
    const uint32_t o = (uint32_t)b - blk_start[r];
    // the keys end in five zero bits: the last one carries the "fold after this block" flag
    const unsigned long long fl = (o + 1 == nblk[r] || (o % GCLS_MAX_RUN) == GCLS_MAX_RUN - 1) ? 1ull : 0ull;
    wblk[b] = ukeys[r] | fl;
  }
}

// ---------------------------------------------------------------------------
// 2. gather.  One wave = sorted block b x the four row tiles 4q .. 4q+3 (one 16-byte column of an L block per locus and
// lane half): lane l fetches the two 16-byte pieces of locus src[64 b + l] (individuals 128 q + 32 s + 16 h + e), the
// wave transposes them through LDS, and lane (r, ho) of output block s ends up with the 32 dosages of individual
// 32 (4q + s) + r at the loci 32 ho .. 32 ho + 31 of the block, one FP4 nibble each: 0 -> 0, 1 -> 0x2 (1.0),
// 2 -> 0x4 (2.0), missing or padding -> 0.  (Which locus sits on which nibble is immaterial: both MFMA operands come
// from this layout.)
__global__ __launch_bounds__(256) void tpg_gcls_gather_kernel(const uint4* __restrict__ L, int64_t Q,
                                                              const int32_t* __restrict__ src, int64_t nblocks,
                                                              uint4* __restrict__ T4g) {
  __shared__ __attribute__((aligned(16))) uint32_t sh[4][2][4][64];  // [wave][source half][s][locus]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 31, ho = lane >> 5, hs = r >> 4, shf = tpg_elem_shift(r & 15);
  for (int64_t task = (int64_t)blockIdx.x * 4 + wv; task < Q * nblocks; task += (int64_t)gridDim.x * 4) {
    const int64_t q = task % Q, b = task / Q;
    const int32_t j = src[b * 64 + lane];
    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0;
    if (j >= 0) {
      const uint4* p = L + (((int64_t)(j >> 5)) * Q + q) * 64 + (j & 31);
      w0 = p[0];
      w1 = p[32];
    }
    sh[wv][0][0][lane] = w0.x; sh[wv][0][1][lane] = w0.y; sh[wv][0][2][lane] = w0.z; sh[wv][0][3][lane] = w0.w;
    sh[wv][1][0][lane] = w1.x; sh[wv][1][1][lane] = w1.y; sh[wv][1][2][lane] = w1.z; sh[wv][1][3][lane] = w1.w;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const uint4* rd = (const uint4*)&sh[wv][hs][s][32 * ho];
      uint32_t out[4];
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const uint4 a = rd[2 * d], c = rd[2 * d + 1];
        const uint32_t ws[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        uint32_t acc = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) acc = (acc << 4) | ((ws[e] >> shf) & 3u);
        const uint32_t m3 = acc & (acc >> 1) & 0x11111111u;  // code 3
        out[d] = ((acc << 1) & 0x66666666u) & ~((m3 << 1) | (m3 << 2));
      }
      T4g[((4 * q + s) * nblocks + b) * 64 + lane] = make_uint4(out[0], out[1], out[2], out[3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------
// 3. class Gram.  Unit (I, J): A row tiles 2I, 2I+1 against B row tiles 3J .. 3J+2 (a tile past the data reads the
// last tile instead; the assemble kernel never looks at its products).
#define MFMA_G4(a, b, c) \
  __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(tpg_g8(a), tpg_g8(b), (c), 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f)
__device__ __forceinline__ v8i tpg_g8(v4u a) { return v8i{(int)a[0], (int)a[1], (int)a[2], (int)a[3], 0, 0, 0, 0}; }

// Operand blocks go from the loads straight into the MFMAs (no decode, no LDS): GCLS_D rotating register slots of
// GA + GB fragments, each filled GCLS_D - 1 blocks (~0.6 us of MFMAs, more with the folds) ahead of its use; the K
// loop is unrolled by GCLS_D so that no slot is ever copied.
// (Tried and dropped: a wave-private LDS ring filled by LDS-DMA, 20.3 ms at 5 000 x 1 000 000; a ring shared by the four
// waves of a workgroup -- half the L2 traffic, one s_barrier per two blocks -- 32 ms.)
#define GCLS_D 6
// the accumulators live in AGPRs; a volatile read keeps hipcc from hoisting the 96 v_accvgpr_read of a fold out of
// the "class ends here" branch into every block of the K loop (it does, behind an s_nop for the MFMA results)
__device__ __forceinline__ float tpg_acc_read(float a) {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}

template <int C, class F>
__device__ __forceinline__ void tpg_static_for(F&& f) {
  if constexpr (C > 0) {
    tpg_static_for<C - 1>(f);
    f(std::integral_constant<int, C - 1>{});
  }
}

__global__ __launch_bounds__(256, 1) void tpg_gcls_gram_kernel(const uint4* __restrict__ T4g, int64_t nblocks, int nrtv,
                                                                  const unsigned long long* __restrict__ wblk,
                                                                  const int2* __restrict__ order, int64_t nun, int S,
                                                                  double* __restrict__ slabs) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, cidx = blockIdx.x >> 3, cpx = gridDim.x >> 3;
  for (int64_t round = 0;; round++) {
    const int64_t un = ((round * 8 + xcd) * cpx + cidx) * 4 + wv;
    if (un >= nun * S) break;
    const int ks = (int)(un / nun);
    const int64_t u = un % nun;
    const int2 ij = order[u];
    const int64_t b0 = (nblocks * ks) / S, b1 = (nblocks * (ks + 1)) / S;
    const uint4* pt[GA + GB];
#pragma unroll
    for (int t = 0; t < GA; t++) pt[t] = T4g + ((int64_t)min(GA * ij.x + t, nrtv - 1) * nblocks) * 64 + lane;
#pragma unroll
    for (int t = 0; t < GB; t++) pt[GA + t] = T4g + ((int64_t)min(GB * ij.y + t, nrtv - 1) * nblocks) * 64 + lane;

    double o[GP][16];
#pragma unroll
    for (int p = 0; p < GP; p++)
#pragma unroll
      for (int i = 0; i < 16; i++) o[p][i] = 0.0;
    v16f acc[GP];
#pragma unroll
    for (int p = 0; p < GP; p++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[p][i] = 0.f;

    if (b0 < b1) {
      const int64_t bl = b1 - 1;
      v4u R[GCLS_D][GA + GB];
      auto LD = [&](const uint4* p) { return *(const v4u*)p; };
      // blocks past the range re-fetch the last one
      tpg_static_for<GCLS_D - 1>([&](auto dd) {
        constexpr int d = decltype(dd)::value;
        const int64_t bc = b0 + d < b1 ? b0 + d : bl;
#pragma unroll
        for (int t = 0; t < GA + GB; t++) R[d][t] = LD(pt[t] + bc * 64);
      });
      bool first = true;
      // weight of the block with the fold flag in its last bit, fetched one block ahead (a scalar load on the
      // path of every block would cost its latency every 6 MFMAs)
      unsigned long long wf_next = wblk[b0];
      for (int64_t bb = b0; bb < b1; bb += GCLS_D) {
        tpg_static_for<GCLS_D>([&](auto cc) {
          constexpr int C = decltype(cc)::value, M = (C + GCLS_D - 1) % GCLS_D;
          const int64_t b = bb + C;
          // the loads are issued on every path (a block past the range re-fetches the last one): hipcc's s_waitcnt
          // bookkeeping merges control-flow paths pessimistically, and a path without them turns the counted waits
          // of the whole loop into vmcnt(0)
          const int64_t bn = b + GCLS_D - 1;
          const int64_t bc = bn < b1 ? bn : bl;
#pragma unroll
          for (int t = 0; t < GA + GB; t++) R[M][t] = LD(pt[t] + bc * 64);
          if (b < b1) {
            const unsigned long long wf = wf_next;
            wf_next = wblk[b < bl ? b + 1 : bl];
            if (first) {  // a new class: the accumulators start from zero (an inline constant, no register writes)
              const v16f z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
              for (int p = 0; p < GP; p++) acc[p] = MFMA_G4(R[C][p / GB], R[C][GA + p % GB], z);
            } else {
#pragma unroll
              for (int p = 0; p < GP; p++) acc[p] = MFMA_G4(R[C][p / GB], R[C][GA + p % GB], acc[p]);
            }
            first = false;
            if ((wf & 1ull) || b == bl) {  // end of the class (or of the range): fold
              const double w = __longlong_as_double((long long)(wf & ~1ull));
#pragma unroll
              for (int p = 0; p < GP; p++)
#pragma unroll
                for (int i = 0; i < 16; i++) o[p][i] = __builtin_fma((double)tpg_acc_read(acc[p][i]), w, o[p][i]);
              first = true;
            }
          }
        });
      }
    }
    double* slab = slabs + ((int64_t)ks * nun + u) * GCLS_SLAB + lane;
#pragma unroll
    for (int p = 0; p < GP; p++)
#pragma unroll
      for (int i = 0; i < 16; i++) slab[(p * 16 + i) * 64] = o[p][i];
  }
}

// 4. the S slabs of every unit, added in split order, into both triangles of K (n x n, column-major)
__global__ __launch_bounds__(256) void tpg_gcls_assemble_kernel(const double* __restrict__ slabs, const int2* __restrict__ order,
                                                                int64_t nun, int S, int n, double* __restrict__ K) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t u = blockIdx.x; u < nun; u += gridDim.x) {
    const int2 ij = order[u];
    for (int q = wv; q < GP * 16; q += 4) {
      const int p = q >> 4, reg = q & 15;
      const int ti = GA * ij.x + p / GB, tk = GB * ij.y + p % GB;
      if (tk < ti) continue;
      const int i = 32 * ti + tpg_cd_row(reg, lane), k = 32 * tk + (lane & 31);
      if (i >= n || k >= n) continue;
      double v = 0;
      for (int s = 0; s < S; s++) v += slabs[((int64_t)s * nun + u) * GCLS_SLAB + (p * 16 + reg) * 64 + lane];
      K[i + (int64_t)k * n] = v;
      K[k + (int64_t)i * n] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// host
struct GclsBufs {
  std::vector<void*> ptrs;
  ~GclsBufs() { for (void* p : ptrs) tpg_pfree(p); }
  template <class T> hipError_t get(T** out, size_t count) {
    void* p = nullptr;
    hipError_t e = tpg_pmalloc(&p, sizeof(T) * std::max<size_t>(count, 1));
    if (e == hipSuccess) { ptrs.push_back(p); *out = (T*)p; }
    return e;
  }
};

// cost models (microseconds per wave), fitted on the two kernels at 5 000 x 1 000 000: per tile 0.095 us per class (fold)
// + 0.043 us per block (operand streaming, not the 0.0175 us of the MFMA itself); digit kernel 1.0 us per 128 loci
static double gcls_cost_classes(int64_t nunits, int64_t nruns, int64_t nblocks, int nwaves, int* bestS) {
  double best = -1;
  for (int S = 1; S <= 32; S++) {
    if (nblocks / S < 4 && S > 1) break;
    const int64_t rounds = ceil_div(nunits * S, (int64_t)nwaves);
    const double per = ((double)nruns / S + 1.0) * GP * 0.095 + (double)ceil_div(nblocks, (int64_t)S) * GP * 0.043 + 6.0;
    const double cost = (double)rounds * per;
    if (best < 0 || cost < best * 0.995) { best = cost; *bestS = S; }
  }
  return best;
}

int tpg_gram_classes(tpg_ctx* ctx, const tpg_view* v, const double* d_w, double* d_what, double* d_K, bool* done) {
  *done = false;
  const int64_t n = v->n, m = v->m;
  if (m >= (1ll << 31) - 64 || getenv("TPG_GRAM_DIGITS")) return TPG_OK;
  GclsBufs B;
  unsigned long long *d_key = nullptr, *d_key2 = nullptr, *d_ukeys = nullptr;
  uint32_t *d_idx = nullptr, *d_idx2 = nullptr, *d_counts = nullptr, *d_cnt = nullptr, *d_nblk = nullptr, *d_estart = nullptr,
           *d_bstart = nullptr;
  int* d_nruns = nullptr;
  long long* d_totals = nullptr;
  void* d_tmp = nullptr;
  TPG_HIP(B.get(&d_key, (size_t)m)); TPG_HIP(B.get(&d_key2, (size_t)m)); TPG_HIP(B.get(&d_ukeys, (size_t)m));
  TPG_HIP(B.get(&d_idx, (size_t)m)); TPG_HIP(B.get(&d_idx2, (size_t)m)); TPG_HIP(B.get(&d_counts, (size_t)m));
  TPG_HIP(B.get(&d_cnt, (size_t)m)); TPG_HIP(B.get(&d_nblk, (size_t)m)); TPG_HIP(B.get(&d_estart, (size_t)m));
  TPG_HIP(B.get(&d_bstart, (size_t)m)); TPG_HIP(B.get(&d_nruns, 1)); TPG_HIP(B.get(&d_totals, 2));
  size_t t_sort = 0, t_rle = 0, t_scan = 0;
  TPG_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, t_sort, d_key, d_key2, d_idx, d_idx2, (int)m, 0, 64, ctx->stream));
  TPG_HIP(hipcub::DeviceRunLengthEncode::Encode(nullptr, t_rle, d_key2, d_ukeys, d_counts, d_nruns, (int)m, ctx->stream));
  TPG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, t_scan, d_cnt, d_estart, (int)m, ctx->stream));
  const size_t t_bytes = std::max(t_sort, std::max(t_rle, t_scan));
  TPG_HIP(B.get((uint8_t**)&d_tmp, t_bytes));
  long long totals[2] = {0, 0};
  {
    ProfScope ps(ctx, "gcls_classes");
    hipLaunchKernelGGL(tpg_gcls_keys_kernel, dim3(1024), dim3(256), 0, ctx->stream, d_w, m, d_key, d_idx);
    size_t t = t_bytes;
    TPG_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp, t, d_key, d_key2, d_idx, d_idx2, (int)m, 0, 64, ctx->stream));
    t = t_bytes;
    TPG_HIP(hipcub::DeviceRunLengthEncode::Encode(d_tmp, t, d_key2, d_ukeys, d_counts, d_nruns, (int)m, ctx->stream));
    hipLaunchKernelGGL(tpg_gcls_run_blocks_kernel, dim3(1024), dim3(256), 0, ctx->stream, (const uint32_t*)d_counts,
                       (const int*)d_nruns, m, d_cnt, d_nblk);
    t = t_bytes;
    TPG_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, t, d_cnt, d_estart, (int)m, ctx->stream));
    t = t_bytes;
    TPG_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, t, d_nblk, d_bstart, (int)m, ctx->stream));
    hipLaunchKernelGGL(tpg_gcls_totals_kernel, dim3(1), dim3(1), 0, ctx->stream, (const int*)d_nruns,
                       (const uint32_t*)d_bstart, (const uint32_t*)d_nblk, d_totals);
    TPG_HIP(hipMemcpyAsync(totals, d_totals, sizeof(totals), hipMemcpyDeviceToHost, ctx->stream));
  }
  TPG_HIP(hipStreamSynchronize(ctx->stream));
  const int64_t nruns = totals[0], nblocks = totals[1];
  TPG_REQUIRE(nruns > 0 && nblocks > 0, TPG_EHIP, "class table is empty");

  // units (I, J) that hold at least one wanted pair of row tiles (B tile >= A tile), in patch order
  const int nrtv = (int)ceil_div(n, 32);
  const int nI = (int)ceil_div(nrtv, GA), nJ = (int)ceil_div(nrtv, GB);
  std::vector<int2> order;
  for (int pj = 0; pj * 8 < nJ; pj++) {
    const int j1 = std::min(nJ, pj * 8 + 8);
    for (int I = 0; I < nI; I++)
      for (int J = pj * 8; J < j1; J++)
        if (GB * J + GB - 1 >= GA * I) order.push_back(make_int2(I, J));
  }
  const int64_t nun = (int64_t)order.size();
  int nblk_grid = ctx->num_cu / 8 * 8;
  if (nblk_grid < 8) nblk_grid = 8;
  const int nwaves = 4 * nblk_grid;
  int S = 1;
  // + the sort, the gather (2.3 us per 1000 loci at n = 5 000: it scales with n m) and the assemble pass
  const double cost_cls = gcls_cost_classes(nun, nruns, nblocks, nwaves, &S) + 650.0 + 2.3e-3 * (double)m * ((double)n / 5000.0);
  // the digit kernel: 32 x 128 wave tiles, 64 int8 MFMAs (~1.0 us) per 128 loci, 4 row tiles x super-tiles of 4
  const int64_t nun_dig = (int64_t)nrtv * ceil_div((int64_t)nrtv, 4) / 2 + nrtv;
  const double cost_dig = (double)ceil_div(nun_dig, (int64_t)nwaves) * ((double)ceil_div(m, 128) * 1.0) + 65.0;
  if (getenv("TPG_DEBUG"))
    fprintf(stderr, "[tpg] gram classes: %lld classes, %lld blocks for %lld loci, S = %d, model %.0f us (digits %.0f us)\n",
            (long long)nruns, (long long)nblocks, (long long)m, S, cost_cls, cost_dig);
  if (cost_cls > cost_dig && !getenv("TPG_GRAM_CLASSES")) return TPG_OK;

  int32_t* d_src = nullptr;
  double* d_slabs = nullptr;
  unsigned long long* d_wblk = nullptr;
  uint4* d_T4g = nullptr;
  int2* d_order = nullptr;
  TPG_HIP(B.get(&d_src, (size_t)nblocks * 64));
  TPG_HIP(B.get(&d_wblk, (size_t)nblocks));
  TPG_HIP(B.get(&d_T4g, (size_t)(4 * v->Q) * (size_t)nblocks * 64));
  TPG_HIP(B.get(&d_order, (size_t)nun));
  TPG_HIP(B.get(&d_slabs, (size_t)S * (size_t)nun * GCLS_SLAB));
  TPG_HIP(tpg_h2d_async(ctx, d_order, order.data(), sizeof(int2) * (size_t)nun));
  TPG_HIP(hipMemsetAsync(d_src, 0xFF, sizeof(int32_t) * (size_t)nblocks * 64, ctx->stream));
  {
    ProfScope ps(ctx, "gcls_layout");
    hipLaunchKernelGGL(tpg_gcls_place_kernel, dim3(1024), dim3(256), 0, ctx->stream, (const uint32_t*)d_idx2,
                       (const unsigned long long*)d_ukeys, (const uint32_t*)d_estart, (const uint32_t*)d_bstart, (int)nruns, m,
                       d_src, d_what);
    hipLaunchKernelGGL(tpg_gcls_block_table_kernel, dim3(256), dim3(256), 0, ctx->stream, (const unsigned long long*)d_ukeys,
                       (const uint32_t*)d_bstart, (const uint32_t*)d_nblk, (int)nruns, nblocks, d_wblk);
  }
  {
    const int64_t tasks = v->Q * nblocks;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(tasks, 4), (int64_t)ctx->num_cu * 16);
    TPG_LAUNCH(ctx, "gcls_gather", tpg_gcls_gather_kernel, dim3(grid), dim3(256), 0, (const uint4*)v->L, v->Q,
               (const int32_t*)d_src, nblocks, d_T4g);
  }
  TPG_LAUNCH(ctx, "pca_gram_classes", tpg_gcls_gram_kernel, dim3((unsigned)nblk_grid), dim3(256), 0, (const uint4*)d_T4g,
             nblocks, nrtv, (const unsigned long long*)d_wblk, (const int2*)d_order, nun, S, d_slabs);
  TPG_LAUNCH(ctx, "gcls_assemble", tpg_gcls_assemble_kernel, dim3((unsigned)std::min<int64_t>(nun, 4096)), dim3(256), 0,
             (const double*)d_slabs, (const int2*)d_order, nun, S, (int)n, d_K);
  TPG_CHECK_LAUNCH();
  *done = true;
  return TPG_OK;  // the scratch blocks go back to the pool in stream order (GclsBufs)
}
