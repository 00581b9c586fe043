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
// doubles (identified up to 2^-47 relative, so that c and 2N - c share a class) and the sums are FP64.  Where classes are
// short the fold is taken apart (tpg_gcls_gram2_kernel): neighbouring classes form groups, the small weight differences
// inside a group are added in packed FP32 (a quarter of the instructions), FP64 comes in at the group ends: 3e-10 of the
// all-FP64 result.
//
// Steps, all on the device:
//   1. key_j = bit pattern of w_j with the last five mantissa bits rounded away; radix sort of (key, locus);
//      run-length encoding -> classes; every class is padded to whole 64-locus blocks (zero dosage contributes
//      nothing); per block its weight and an "accumulators must be folded after this block" flag (end of class, or
//      2^16 blocks = 2^22 loci since the last fold: 4 x 2^22 is the largest FP32 sum that is still exact).
//   2. tpg_gcls_gather_kernel: the class-sorted operand layout T2g -- block (rt, b) = 32 individuals x the 64 loci
//      of sorted block b, the 2-bit dosage codes (missing -> 0) -- gathered from the view's L layout.
//   3. tpg_gcls_gram_kernel / tpg_gcls_gram2_kernel: one wave = a 64 x 64 tile of pairs (2 x 2 accumulator tiles) over a
//      range of blocks, two waves per SIMD.  A 16-byte load per lane carries the operand of two blocks; one v_and_b32 (+ one shift) per
//      operand word makes the FP4 nibbles in registers.  K split S: each (unit, split) writes its own FP64 slab.
//   4. tpg_gcls_assemble_kernel: the S slabs of a unit are added in a fixed order (run-to-run identical results)
//      and written to both triangles of the n x n matrix.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "common.h"
#include "devfrag.h"
#include "host/host_bits2.h"  // tpg_transpose16_2bit

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// This file is compiled with -mllvm -amdgpu-mfma-vgpr-form (csrc/Makefile): the MFMAs of the Gram kernel write their
// sums to VGPRs and the kernel uses no AGPR at all, so that hipcc gives a two-waves-per-SIMD kernel its whole budget as
// VGPRs (256 instead of 128 + 128) and a fold is v_cvt_f64_f32 + v_fma_f64 per element, without a v_accvgpr_read.
constexpr int GA = 2, GB = 2, GP = GA * GB;  // wave tile = 2 x 2 accumulator tiles of 32 x 32 pairs
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

__device__ __forceinline__ int tpg_gcls_find_run(const uint32_t* __restrict__ start, int nr, uint32_t x) {
  int lo = 0, hi = nr - 1;  // last r with start[r] <= x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (start[mid] <= x) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Groups of neighbouring classes (the mixed-precision fold, see tpg_gcls_block_table2_kernel) for the one-wave-per-SIMD
// kernel: a group must END on a multiple of `body` blocks, so that its FP64 fold only ever falls between two bodies of the
// kernel's unrolled loop.  Group ends are a property of the classes alone (key prefix, position in the run of classes), so
// every group starts on a multiple of `body` if all before it do, and the padding of a group is its own size rounded up:
// the last class of every group gets the extra (empty) blocks.  sblk[r] = blocks reserved for class r, ge[r] = last of its group.
__global__ void tpg_gcls_group_pad_kernel(const unsigned long long* __restrict__ ukeys, const int* __restrict__ nruns, int64_t m,
                                          const uint32_t* __restrict__ nblk, int gmax, int gq, int body,
                                          uint32_t* __restrict__ sblk, uint8_t* __restrict__ ge) {
  const int nr = nruns[0];
  auto group_end = [&](int r) {
    return r == nr - 1 || (ukeys[r] >> (52 - gq)) != (ukeys[r + 1] >> (52 - gq)) || (r % gmax) == gmax - 1;
  };
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < m; r += (int64_t)gridDim.x * blockDim.x) {
    if (r >= nr) { sblk[r] = 0; ge[r] = 0; continue; }
    const bool e = group_end((int)r);
    uint32_t tot = nblk[r];
    if (e && body > 1) {
      for (int q = (int)r - 1; q >= 0 && !group_end(q); q--) tot += nblk[q];  // at most gmax classes
      sblk[r] = nblk[r] + (uint32_t)((body - (int)(tot % (uint32_t)body)) % body);
    } else {
      sblk[r] = nblk[r];
    }
    ge[r] = e ? 1 : 0;
  }
}

// Block table of tpg_gcls_gram1w_kernel over the padded layout: entry as in tpg_gcls_block_table2_kernel (x = weight | flags,
// y = bits of (float)(w_c - w_{c+1})); flag 1 = last block of a class that is not the last of its group; flag 2 = FP64 fold
// after this block -- the last (possibly empty) block of a group, or a block whose index is GCLS3_GRUN - 1 modulo GCLS3_GRUN
// (integer sums stay exact: 4 * 64 * 16 320 < 2^23): always the last block of a body, GCLS3_GRUN being a multiple of 8, 10, 12.
#define GCLS3_GRUN 16320
__global__ void tpg_gcls_block_table3_kernel(const unsigned long long* __restrict__ ukeys, const uint32_t* __restrict__ blk_start,
                                             const uint32_t* __restrict__ nblk, const uint32_t* __restrict__ sblk,
                                             const uint8_t* __restrict__ ge, int nr, int64_t nblocks, ulonglong2* __restrict__ wblk) {
  for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < nblocks; b += (int64_t)gridDim.x * blockDim.x) {
    const int r = tpg_gcls_find_run(blk_start, nr, (uint32_t)b);
    const uint32_t o = (uint32_t)b - blk_start[r];
    const bool full = (ge[r] && o + 1 == sblk[r]) || (b % GCLS3_GRUN) == GCLS3_GRUN - 1;
    const bool class_end = !ge[r] && o + 1 == nblk[r];
    ulonglong2 e;
    e.x = ukeys[r] | (full ? 2ull : class_end ? 1ull : 0ull);
    e.y = 0;
    if (class_end && !full)
      e.y = (unsigned long long)__float_as_uint(
          (float)(__longlong_as_double((long long)ukeys[r]) - __longlong_as_double((long long)ukeys[r + 1])));
    wblk[b] = e;
  }
}

__global__ void tpg_gcls_totals_kernel(const int* __restrict__ nruns, const uint32_t* __restrict__ blk_start,
                                       const uint32_t* __restrict__ nblk, long long* __restrict__ totals) {
  const int nr = nruns[0];
  totals[0] = nr;
  totals[1] = nr > 0 ? (long long)blk_start[nr - 1] + nblk[nr - 1] : 0;
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

// Block table of the mixed-precision fold (tpg_gcls_gram2_kernel).  Classes are taken in GROUPS of neighbouring weights
// (same exponent and leading GCLS_GQ mantissa bits: a relative span below 2^-GCLS_GQ; at most GCLS_GMAX classes; fewer than
// 2^15 blocks, so that the running integer sums of a group stay below 2^23).  Inside a group the pair counts are never
// reset: with P_c the running sums at the end of class c,
//   sum_c w_c G_c = w_last P_last + sum_{c < last} (w_c - w_{c+1}) P_c        (summation by parts, exact),
// and the second term is small (|w_c - w_{c+1}| <= 2^-GCLS_GQ w), so it is accumulated in FP32 by one packed FMA per two
// elements and class (v_pk_fma_f32); only the group end pays the FP64 fold (cvt + fma for P_last, cvt + add for the FP32
// sum).  Entry of block b: x = weight of its class (the keys end in five zero bits) | flags, y = bits of
// (float)(w_c - w_{c+1}); flag 1 = last block of a class inside a group, flag 2 = full fold after this block (group end,
// or 2^14 blocks of one class).
#define GCLS_GQ 6
#define GCLS_GMAX 64
#define GCLS_GRUN 16384
__global__ void tpg_gcls_block_table2_kernel(const unsigned long long* __restrict__ ukeys, const uint32_t* __restrict__ blk_start,
                                             const uint32_t* __restrict__ nblk, int nr, int64_t nblocks, int gmax, int gq,
                                             ulonglong2* __restrict__ wblk) {
  // (entries after the last block: empty blocks of the last class without a flag -- tpg_gcls_gram2_kernel works in whole
  // block pairs and reads the entry of the block that pads an odd list)
  if (blockIdx.x == 0 && threadIdx.x < 4) wblk[nblocks + threadIdx.x] = make_ulonglong2(ukeys[nr - 1], 0ull);
  for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < nblocks; b += (int64_t)gridDim.x * blockDim.x) {
    const int r = tpg_gcls_find_run(blk_start, nr, (uint32_t)b);
    const uint32_t o = (uint32_t)b - blk_start[r];
    const bool class_end = o + 1 == nblk[r];
    bool full = (o % GCLS_GRUN) == GCLS_GRUN - 1;
    if (class_end)
      full = full || r == nr - 1 || (ukeys[r] >> (52 - gq)) != (ukeys[r + 1] >> (52 - gq)) ||
             (r % gmax) == gmax - 1 || blk_start[r] / GCLS_GRUN != (blk_start[r] + nblk[r]) / GCLS_GRUN;
    ulonglong2 e;
    e.x = ukeys[r] | (full ? 2ull : class_end ? 1ull : 0ull);
    e.y = 0;
    if (class_end && !full)
      e.y = (unsigned long long)__float_as_uint(
          (float)(__longlong_as_double((long long)ukeys[r]) - __longlong_as_double((long long)ukeys[r + 1])));
    wblk[b] = e;
  }
}

// ---------------------------------------------------------------------------
// 2. gather into the class-sorted 2-BIT operand layout T2g of the Gram kernel.  One wave = a PAIR of sorted blocks
// (2 bp, 2 bp + 1) x the four row tiles 4q .. 4q+3 (one 16-byte column of an L block per locus and lane half): lane l
// fetches the two 16-byte pieces of locus src[64 b + l] (individuals 128 q + 32 s + 16 h + e), the wave transposes them
// through LDS, and lane (r, ho) of output row tile s ends up with the 32 dosages of individual 32 (4q + s) + r at the
// loci 32 ho .. 32 ho + 31 of each block: one uint4 = {P0, P1 of block 2 bp, P0, P1 of block 2 bp + 1}, a source dword
// P packing 16 codes as nibbles [c_odd | c_even], so that P & 0x33333333 and (P >> 2) & 0x33333333 are FP4 operand
// words of value dosage / 2.  Padding -> 0; the view holds no missing value (checked by both callers of the class path).  (Which locus sits on which nibble is immaterial: both MFMA
// operands come from this layout.)
// Where the 16-byte pieces of a locus live (uint4 units): piece (j, q, h) at base + (j >> sh) * sa + (j & msk) * sb +
// q * sq + h * sh1.  A view's L layout: 32 loci per 1-KiB block, {5, Q * 64, 31, 1, 64, 32}; records received from other
// ranks (tpg_gram_classes_exchanged): one locus after the other, {0, record size / 16, 0, 0, 2, 1}.
struct GclsSrc {
  const uint4* base;
  int sh;
  int64_t sa;
  int msk;
  int64_t sb, sq, sh1;
};

// CEN (tpg_gcls_gram2_kernel<VT, true> only): a source dword packs the codes of BOTH blocks of the pair -- low two bits of
// a nibble: the dosage code of a locus of block 2 bp, as before (P & 0x33333333 = FP4 dosage / 2); high two bits: a locus
// of block 2 bp + 1 as its CENTRED dosage g - 1, coded 3 / 0 / 1 so that P & 0xCCCCCCCC is the FP4 value -2 / 0 / +2 -- and
// the kernel makes an operand word with ONE v_and_b32 instead of a shift and an AND (4 instead of 6 VALU per MFMA).  A
// padding locus of the odd block is 0 = "dosage 1".  What the products of the odd blocks lack, w (g_i + g_k - 1) summed over
// their loci, is r_i + r_k + const: the double centring of the PCA removes exactly such terms, so this layout is only used
// where that centring follows (tpg_gram_classes' centred_ok).
//
// One wave = one task (QW chunks q of 128 individuals, block pair); per chunk 128 loci x 128 individuals, 4 KiB in and out.
// Lane l fetches the 32 QW bytes of locus l of either block (the next task's are in flight while this one is turned) and, chunk
// after chunk, puts eight dwords (16 individuals each) into LDS rows of 64 loci; then every lane owns ONE 16 x 16
// tile of 2-bit codes (block hb, row tile s, half hs, 16 loci g): four ds_read_b128, the register transposition of
// host/host_bits2.h (96 instructions for 256 genotypes; picking two bits at a time took 512), sixteen ds_write_b32 that hand
// the words to the lanes of the operand layout.  A transposed word IS an operand dword -- 16 loci of one individual, the
// kernels take P & 0x33333333 and (P >> 2) & 0x33333333 -- and the order of the loci inside a block is free as long as every
// individual uses the same one: dword 2 hb + t of lane (r, ho) holds loci 16 (2 ho + t) + k of block hb at field k.
#define GSH_ROW 68                 // source rows: 64 loci + 4 (a tile's four 16-byte reads meet no other tile's bank)
#define GSH_HS 140                 // result rows of 8 dwords per individual: 16 individuals + 12 dwords
#define GSH_WORDS (4 * 2 * GSH_HS) // per wave; the 16 source rows (1088 dwords) live in the same place
template <bool CEN, int QW>
__global__ __launch_bounds__(256) void tpg_gcls_gather_kernel(GclsSrc S, int64_t Q,
                                                               const int32_t* __restrict__ src, int64_t nblocks,
                                                               int64_t rs2, uint4* __restrict__ T2g) {  // rs2 pairs per row tile, all written
  __shared__ __attribute__((aligned(16))) uint32_t shm[4][GSH_WORDS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t* sh = shm[wv];
  // this lane's tile in the transposition: row x = (hb * 2 + hs) * 4 + s of the source, loci 16 g .. 16 g + 15
  const int g = lane & 3, x = lane >> 2, t_s = x & 3, t_hs = (x >> 2) & 1, t_hb = x >> 3;
  const uint4* rd_src = (const uint4*)(sh + x * GSH_ROW + 16 * g);
  uint32_t* wr_dst = sh + t_s * (2 * GSH_HS) + t_hs * GSH_HS + (g >> 1) * 4 + 2 * t_hb + (g & 1);  // + 8 * individual
  // this lane in the operand layout: individual r of a row tile, loci 32 ho .. 32 ho + 31
  const int r = lane & 31, ho = lane >> 5;
  const uint4* rd_dst = (const uint4*)(sh + (r >> 4) * GSH_HS + (r & 15) * 8 + ho * 4);  // + s * 2 * GSH_HS
  const uint32_t uQ = (uint32_t)((Q + QW - 1) / QW), ntask = uQ * (uint32_t)rs2, step = gridDim.x * 4u;  // (the host has checked Q rs2 < 2^31)
  auto load_j = [&](uint32_t task, int32_t (&j)[2]) {
    j[0] = j[1] = -1;
    if (task < ntask) {
      const int64_t bp = task / uQ;
#pragma unroll
      for (int hb = 0; hb < 2; hb++)
        if (2 * bp + hb < nblocks) j[hb] = src[(2 * bp + hb) * 64 + lane];
    }
  };
  auto load_d = [&](uint32_t task, const int32_t (&j)[2], uint4 (&w)[QW][2][2]) {
    const int64_t q0 = (int64_t)(task % uQ) * QW;
#pragma unroll
    for (int hb = 0; hb < 2; hb++) {
#pragma unroll
      for (int qi = 0; qi < QW; qi++) w[qi][hb][0] = w[qi][hb][1] = make_uint4(0, 0, 0, 0);
      if (j[hb] >= 0) {
        const uint4* p = S.base + (int64_t)(j[hb] >> S.sh) * S.sa + (int64_t)(j[hb] & S.msk) * S.sb + q0 * S.sq;
#pragma unroll
        for (int qi = 0; qi < QW; qi++)
          if (qi == 0 || q0 + qi < Q) {
            w[qi][hb][0] = p[qi * S.sq];
            w[qi][hb][1] = p[qi * S.sq + S.sh1];
          }
      }
    }
  };
  uint32_t task = blockIdx.x * 4u + wv;
  int32_t j0[2], j1[2];
  uint4 w[QW][2][2];
  load_j(task, j0);
  load_d(task, j0, w);
  load_j(task + step, j1);
  for (; task < ntask; task += step) {
    uint4 wn[QW][2][2];  // the next task's bytes and the indices of the one after it are on their way while this one is turned
    int32_t j2[2];
    load_d(task + step, j1, wn);
    load_j(task + 2 * step, j2);
    const int64_t q0 = (int64_t)(task % uQ) * QW, bp = task / uQ;
    const unsigned long long valid = __ballot(j0[1] >= 0);  // bit l: locus l of the odd block exists
    (void)valid;
    uint32_t v55[2] = {0, 0};
    if constexpr (CEN) {  // bit 2 k of v55[t]: locus 16 (2 ho + t) + k of the odd block exists
#pragma unroll
      for (int t = 0; t < 2; t++) {
        uint32_t vb = (uint32_t)(valid >> (32 * ho + 16 * t)) & 0xFFFFu;
        vb = (vb | (vb << 8)) & 0x00FF00FFu;
        vb = (vb | (vb << 4)) & 0x0F0F0F0Fu;
        vb = (vb | (vb << 2)) & 0x33333333u;
        v55[t] = (vb | (vb << 1)) & 0x55555555u;
      }
    }
#pragma unroll
    for (int qi = 0; qi < QW; qi++) {
    const int64_t q = q0 + qi;
    if (qi && q >= Q) break;
#pragma unroll
    for (int hb = 0; hb < 2; hb++)
#pragma unroll
      for (int hs = 0; hs < 2; hs++) {
        uint32_t* row = sh + ((hb * 2 + hs) * 4) * GSH_ROW + lane;
        row[0] = w[qi][hb][hs].x; row[GSH_ROW] = w[qi][hb][hs].y; row[2 * GSH_ROW] = w[qi][hb][hs].z; row[3 * GSH_ROW] = w[qi][hb][hs].w;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint32_t W[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint4 a = rd_src[i];
      W[4 * i] = a.x; W[4 * i + 1] = a.y; W[4 * i + 2] = a.z; W[4 * i + 3] = a.w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (one wave, one LDS queue: the stores below cannot pass these loads)
    __builtin_amdgcn_wave_barrier();
    tpg_transpose16_2bit(W);
#pragma unroll
    for (int p = 0; p < 16; p++) wr_dst[8 * ((p >> 2) + 4 * (p & 3))] = W[p];  // field p of a source dword = individual p / 4 + 4 (p % 4) (tpg_elem_shift)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int s = 0; s < 4; s++) {
      uint4 o = rd_dst[s * (2 * GSH_HS / 4)];  // {even block t = 0, 1; odd block t = 0, 1}
      if constexpr (CEN) {
        const uint32_t A[2] = {o.x, o.y}, B[2] = {o.z, o.w};
        uint32_t out[4];
#pragma unroll
        for (int t = 0; t < 2; t++) {
          // g - 1 on the loci that exist, as the code (c + 3) & 3: low bit = ~c0, high bit = ~c0 & ~c1
          const uint32_t nb = ~B[t], lo = nb & v55[t], hi = (nb >> 1) & lo, bc = lo | (hi << 1);
          out[2 * t] = (A[t] & 0x33333333u) | ((bc & 0x33333333u) << 2);
          out[2 * t + 1] = ((A[t] >> 2) & 0x33333333u) | (bc & 0xCCCCCCCCu);
        }
        o = make_uint4(out[0], out[1], out[2], out[3]);
      }
      T2g[((4 * q + s) * rs2 + bp) * 64 + lane] = o;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int hb = 0; hb < 2; hb++) {
#pragma unroll
      for (int qi = 0; qi < QW; qi++) { w[qi][hb][0] = wn[qi][hb][0]; w[qi][hb][1] = wn[qi][hb][1]; }
      j0[hb] = j1[hb]; j1[hb] = j2[hb];
    }
  }
}

// ---------------------------------------------------------------------------
__device__ __forceinline__ v8i tpg_g8(v4u a) { return v8i{(int)a[0], (int)a[1], (int)a[2], (int)a[3], 0, 0, 0, 0}; }

__device__ __forceinline__ int64_t tpg_uniform64(int64_t x) {  // a wave-uniform value the compiler keeps in SGPRs
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)x >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <int C, class F>
__device__ __forceinline__ void tpg_static_for(F&& f) {
  if constexpr (C > 0) {
    tpg_static_for<C - 1>(f);
    f(std::integral_constant<int, C - 1>{});
  }
}

// ---------------------------------------------------------------------------
// 3. class Gram.  Unit (I, J): row tiles 2I, 2I+1 against row tiles 2J, 2J+1 (a tile past the data reads the last
// tile instead; the assemble kernel never looks at its products); one wave = a 64 x 64 tile of pairs over a range of
// blocks, TWO waves per SIMD (the grid is 2 x one workgroup per CU).
//
// What bounds it, measured (tools/gram_only.py, rocprofv3 --pmc): (1) the L2 -> CU path, 19 - 20 TB/s over
// the chip = 1 KiB per XCD and clock, the same with every load an L2 hit -- so a genotype crosses it as its 2-BIT code,
// not as an FP4 nibble: one 16-byte load per lane holds the operand of TWO 64-locus blocks, and two VALU instructions per
// source dword turn it into FP4 operand words in registers.  The code itself IS an FP4 value (nibble 0001 = 0.5,
// 0010 = 1.0), so X = P & 0x33333333 and X' = (P >> 2) & 0x33333333 are operand words of value dosage / 2, and the E8M0
// block scales of the MFMA (2 on both sides) give the products back as 0 / 1 / 2 / 4: exact integers in FP32 below 2^24.
// (2) VALU issue: 6 instructions per MFMA for that expansion beside the 8.5 of the folds (out64 += w_c * (double)acc at
// the end of every class: v_cvt_f64_f32 + v_fma_f64 per element).  The second wave of a SIMD is what lets MFMAs, loads
// and folds overlap at all: with one wave per SIMD (64 x 96 tiles, FP4 operands) the same work took 16.4 ms against 12.2.
// The first MFMA of a class takes the inline constant 0 as C, so nothing is ever zeroed.
// K split S: each (unit, split) writes its own FP64 slab with plain stores.
#define MFMA_G4S2(a, b, c) \
  __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(tpg_g8(a), tpg_g8(b), (c), 4, 4, 0, (int)0x80808080, 0, (int)0x80808080)

// GCLS_D rotating register slots of GA + GB source fragments (two blocks each), filled GCLS_D - 1 pairs ahead of their use;
// the K loop is unrolled by GCLS_D so that no slot is ever copied.  (Tried and dropped: three slots -- hipcc spills;
// FP4 operands in HBM -- 13.4 ms, L2 -> CU bound; the two waves of a SIMD on the same block range -- same time; 512-thread
// workgroups -- 16 ms; a wave-private LDS ring filled by LDS-DMA; a ring shared by the waves of a workgroup.)
#define GCLS_D 2
__global__ __launch_bounds__(256, 2) void tpg_gcls_gram_kernel(const uint4* __restrict__ T2g, int64_t nblocks, int64_t rs2, int nrtv,
                                                               const unsigned long long* __restrict__ wblk,
                                                               const int2* __restrict__ order, int64_t nun, int S,
                                                               double* __restrict__ slabs) {
  constexpr int WPE = 2;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Workgroups b and b + gridDim.x / 2 share a CU (dispatch is round-robin over the XCDs, then over the CUs:
  // tools/xcc_probe.hip), i.e. every SIMD holds one wave of each HALF of the grid.  The halves take different halves of the
  // K range (split indices [0, S/2) and [S/2, S)), so the two waves of a SIMD meet their class boundaries at different
  // times.  XCD x takes in every round a run of consecutive units of one K range, whose re-reads hit its own L2.
  const int hgrid = gridDim.x / WPE, half = (int)blockIdx.x / hgrid, bx = (int)blockIdx.x % hgrid;
  const int xcd = bx & 7, cidx = bx >> 3, cpx = hgrid >> 3;
  const int SH = S / WPE;  // splits per half (the host makes S even)
  const int64_t npairs = (nblocks + 1) >> 1;
  for (int64_t round = 0;; round++) {
    const int64_t un = ((round * 8 + xcd) * cpx + cidx) * 4 + wv;
    if (un >= nun * SH) break;
    const int ks = half * SH + (int)(un / nun);
    const int64_t u = un % nun;
    const int2 ijv = order[u];
    const int2 ij = make_int2(__builtin_amdgcn_readfirstlane(ijv.x), __builtin_amdgcn_readfirstlane(ijv.y));
    const int64_t p0 = tpg_uniform64((npairs * ks) / S), p1 = tpg_uniform64((npairs * (ks + 1)) / S);  // block pairs
    const int64_t bend = 2 * p1 < nblocks ? 2 * p1 : nblocks;                                          // blocks [2 p0, bend)
    const uint4* pt[GA + GB];
#pragma unroll
    for (int t = 0; t < GA; t++) pt[t] = T2g + ((int64_t)min(GA * ij.x + t, nrtv - 1) * rs2) * 64;
#pragma unroll
    for (int t = 0; t < GB; t++) pt[GA + t] = T2g + ((int64_t)min(GB * ij.y + t, nrtv - 1) * rs2) * 64;

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

    if (p0 < p1) {
      const int64_t pl = p1 - 1, bl = bend - 1;
      v4u R[GCLS_D][GA + GB];
      // uniform base + one 32-bit lane offset (global_load_dwordx4 v, v_off, s[base]); the empty asm keeps hipcc from
      // folding the lane into four loop-invariant 64-bit VGPR pointers (8 registers and a v_lshl_add_u64 per load)
      auto LD = [&](const uint4* p) {
        uint32_t off = (uint32_t)lane * 16u;
        asm("" : "+v"(off));
        return *(const v4u*)((const char*)p + off);
      };
      // pairs past the range re-fetch the last one: the loads are issued on every path, because hipcc's s_waitcnt
      // bookkeeping merges control-flow paths pessimistically and a path without them turns the counted waits of the
      // whole loop into vmcnt(0)
      tpg_static_for<GCLS_D - 1>([&](auto dd) {
        constexpr int d = decltype(dd)::value;
        const int64_t pc = p0 + d < p1 ? p0 + d : pl;
#pragma unroll
        for (int t = 0; t < GA + GB; t++) R[d][t] = LD(pt[t] + pc * 64);
      });
      bool first = true;
      unsigned long long wf_next = wblk[2 * p0];
      for (int64_t pp = p0; pp < p1; pp += GCLS_D) {
        tpg_static_for<GCLS_D>([&](auto cc) {
          constexpr int C = decltype(cc)::value, M = (C + GCLS_D - 1) % GCLS_D;
          const int64_t pr = pp + C;
          const int64_t pn = pr + GCLS_D - 1;
          const int64_t pc = pn < p1 ? pn : pl;
#pragma unroll
          for (int t = 0; t < GA + GB; t++) R[M][t] = LD(pt[t] + pc * 64);
          if (pr < p1) {
#pragma unroll
            for (int hb = 0; hb < 2; hb++) {
              const int64_t b = 2 * pr + hb;
              if (b < bend) {
                const unsigned long long wf = wf_next;
                wf_next = wblk[b < bl ? b + 1 : bl];
                v4u X[GA + GB];
#pragma unroll
                for (int t = 0; t < GA + GB; t++) {
                  const uint32_t w0 = R[C][t][2 * hb], w1 = R[C][t][2 * hb + 1];
                  X[t] = v4u{w0 & 0x33333333u, (w0 >> 2) & 0x33333333u, w1 & 0x33333333u, (w1 >> 2) & 0x33333333u};
                }
                if (first) {
                  const v16f z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                  for (int p = 0; p < GP; p++) acc[p] = MFMA_G4S2(X[p / GB], X[GA + p % GB], z);
                } else {
#pragma unroll
                  for (int p = 0; p < GP; p++) acc[p] = MFMA_G4S2(X[p / GB], X[GA + p % GB], acc[p]);
                }
                first = false;
                if ((wf & 1ull) || b == bl) {
                  const double w = __longlong_as_double((long long)(wf & ~1ull));
#pragma unroll
                  for (int p = 0; p < GP; p++)
#pragma unroll
                    for (int i = 0; i < 16; i++) o[p][i] = __builtin_fma((double)acc[p][i], w, o[p][i]);
                  first = true;
                }
              }
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

// 3b. the same contraction with the mixed-precision fold (table: tpg_gcls_block_table2_kernel), still two waves per SIMD:
// a wave keeps the FP4 sums (64 registers), the FP32 sum of the small terms (64) and HALF of its FP64 result (tiles 0, 1:
// 64 registers); the other half (tiles 2, 3: 16 KiB per wave) lives in LDS, where the group ends -- once per ~20 classes
// -- read, update and write it.  A class end costs 32 v_pk_fma_f32 instead of 64 v_cvt_f64_f32 + 64 v_fma_f64.
// (One wave per SIMD with everything in registers and AGPRs, which has room for all of it: 24 ms against 12 -- nothing
// hides the scalar bookkeeping and the load latency of a single wave.)
typedef float v2f __attribute__((ext_vector_type(2)));
// cost model of this kernel (microseconds per wave and 32 x 32 tile): per class, per block
#define GCLS2_T_CLASS 0.045  // (refitted at the end of round 4: 9.2 ms for 4 728 classes in 17 888 blocks at n = 5 000)
#define GCLS2_T_BLOCK 0.067
#define GCLS2_LDS_BYTES (4 * 2 * 16 * 64 * 8)
#ifndef GCLS2_D
#define GCLS2_D 2
#endif
// VT: the block table through one vector load per loop body + v_readlane_b32 instead of one scalar load per block (see 3c)
// The scalar stream of the loop is kept LEAN on purpose.  Two experiments (operand loads removed: 10.4 -> 8.75 ms; class-end
// and fold flags ignored: 10.3 -> 7.8 ms; neither leaves less than 2.4 x the MFMA time) said that what the waves wait for is
// mostly their OWN instruction stream: a wave issues one instruction per four cycles, whatever its kind, and per block (4 MFMAs
// = 128 cycles of the pipe) the first form of this loop spent 16 - 24 VALU, 22 SALU, 4 branches and half a dozen s_waitcnt /
// s_nop.  So: 32-bit block indices (as 64-bit values `b < bend` was a VALU compare of two uniform values + a branch on VCC,
// every increment an s_add_u32 + s_addc_u32); K ranges in whole block pairs over a list whose padding block is empty and
// flagless (no range test per block); fixed tile bases in SGPRs + ONE lane offset per pair (a v_lshl_or_b32 where a pointer
// per tile cost an s_add_u32 + s_addc_u32 each); ONE test of the two flag bits per block with everything behind it out of
// line (__builtin_expect: the FP64 fold no longer sits in the loop body); the fold of the range end after the loop instead of
// a comparison in every block.  10.4 -> 9.2 ms with the same results (a rewrite from scratch on these lines spilled 358
// registers and ran at 43 ms: at 256 registers the allocator decides, so the existing loop was trimmed instead).
// CEN: operands in the centred layout of tpg_gcls_gather_kernel<true> (one v_and_b32 per operand word; the odd block of a
// pair takes the high halves of the nibbles, values +-2, with the block scales 2^-1 where the even block has 2^+1)
#define MFMA_G4S2_ODD(a, b, c) \
  __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(tpg_g8(a), tpg_g8(b), (c), 4, 4, 0, (int)0x7E7E7E7E, 0, (int)0x7E7E7E7E)
template <bool VT, bool CEN = false>
__global__ __launch_bounds__(256, 2) void tpg_gcls_gram2_kernel(const uint4* __restrict__ T2g, int64_t nblocks, int64_t rs2, int nrtv,
                                                                const ulonglong2* __restrict__ wblk,
                                                                const int2* __restrict__ order, int64_t nun, int S,
                                                                double* __restrict__ slabs) {
  extern __shared__ double olds_raw[];  // [wave][tile 2, 3][register][lane]
  constexpr int WPE = 2;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* olds = olds_raw + (size_t)wv * (2 * 16 * 64) + lane;
  const int hgrid = gridDim.x / WPE, half = (int)blockIdx.x / hgrid, bx = (int)blockIdx.x % hgrid;
  const int xcd = bx & 7, cidx = bx >> 3, cpx = hgrid >> 3;
  const int SH = S / WPE;  // splits per half (the host makes S even)
  const int64_t npairs = (nblocks + 1) >> 1;
  for (int64_t round = 0;; round++) {
    const int64_t un = ((round * 8 + xcd) * cpx + cidx) * 4 + wv;
    if (un >= nun * SH) break;
    const int ks = half * SH + (int)(un / nun);
    const int64_t u = un % nun;
    const int2 ijv = order[u];
    const int2 ij = make_int2(__builtin_amdgcn_readfirstlane(ijv.x), __builtin_amdgcn_readfirstlane(ijv.y));
    // block pairs [p0, p1), blocks [2 p0, bend): 32-bit (the class path stops below 2^31 loci), so that the range tests of
    // the loop are scalar compares -- as 64-bit values `b < bend` was a VALU compare of two uniform values and a branch on VCC
    const int p0 = __builtin_amdgcn_readfirstlane((int)((npairs * ks) / S)), p1 = __builtin_amdgcn_readfirstlane((int)((npairs * (ks + 1)) / S));
    // (whole pairs: the block after an odd number of blocks is empty in T2g and has a flagless entry in the table, so
    // no block of the loop needs a range test)
    const int bend = 2 * p1;
    // the tiles' rows of T2g from pair p0 on: fixed bases in SGPRs; a pair is one 32-bit lane offset (a v_lshl_add_u32 per
    // pair where a pointer per tile cost an s_add_u32 + s_addc_u32 each; the host keeps a K range below 2^21 pairs)
    const char* pt[GA + GB];
#pragma unroll
    for (int t = 0; t < GA; t++) pt[t] = (const char*)(T2g + ((int64_t)min(GA * ij.x + t, nrtv - 1) * rs2 + p0) * 64);
#pragma unroll
    for (int t = 0; t < GB; t++) pt[GA + t] = (const char*)(T2g + ((int64_t)min(GB * ij.y + t, nrtv - 1) * rs2 + p0) * 64);

    double o[2][16];
    v2f dev[GP][8];
    v16f acc[GP];
#pragma unroll
    for (int p = 0; p < GP; p++) {
#pragma unroll
      for (int i = 0; i < 16; i++) acc[p][i] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; i++) dev[p][i] = v2f{0.f, 0.f};
    }
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int i = 0; i < 16; i++) { o[p][i] = 0.0; olds[(p * 16 + i) * 64] = 0.0; }

    if (p0 < p1) {
      const int pl = p1 - 1, bl = bend - 1;
      v4u R[GCLS2_D][GA + GB];
      auto LDP = [&](v4u* slot, int pc) {  // pair pc of the four tiles
        uint32_t off = (uint32_t)lane * 16u + (uint32_t)(pc - p0) * 1024u;
        asm("" : "+v"(off));
#pragma unroll
        for (int t = 0; t < GA + GB; t++) slot[t] = *(const v4u*)(pt[t] + off);
      };
      tpg_static_for<GCLS2_D - 1>([&](auto dd) {
        constexpr int d = decltype(dd)::value;
        LDP(R[d], p0 + d < p1 ? p0 + d : pl);
      });
      int st = 0;  // 2: the class end before this block is still to be folded
      float pdelta = 0.f;
      ulonglong2 wf_next = wblk[2 * p0];
      const unsigned long long wlast = wblk[bl].x;  // for the fold after the loop
      typedef uint32_t v3u __attribute__((ext_vector_type(3)));
      auto LDT = [&](int first_block) {  // lane l: the table entry of block first_block + l (three dwords: x, low half of y)
        const int bb = first_block + lane;
        return *(const v3u*)(wblk + (bb < bl ? bb : bl));
      };
      v3u TB = {0, 0, 0};
      if constexpr (VT) TB = LDT(2 * p0);
      uint32_t tx0[2 * GCLS2_D], tx1[2 * GCLS2_D], ty0[2 * GCLS2_D];
      for (int pp = p0; pp < p1; pp += GCLS2_D) {
        if constexpr (VT) {
#pragma unroll
          for (int k = 0; k < 2 * GCLS2_D; k++) {
            tx0[k] = (uint32_t)__builtin_amdgcn_readlane((int)TB[0], k);
            tx1[k] = (uint32_t)__builtin_amdgcn_readlane((int)TB[1], k);
            ty0[k] = (uint32_t)__builtin_amdgcn_readlane((int)TB[2], k);
          }
          __builtin_amdgcn_sched_barrier(0);
          TB = LDT(2 * (pp + GCLS2_D));
          __builtin_amdgcn_sched_barrier(0);
        }
        tpg_static_for<GCLS2_D>([&](auto cc) {
          constexpr int C = decltype(cc)::value, M = (C + GCLS2_D - 1) % GCLS2_D;
          const int pr = pp + C;
          const int pn = pr + GCLS2_D - 1;
          LDP(R[M], pn < p1 ? pn : pl);
          if (pr < p1) {
#pragma unroll
            for (int hb = 0; hb < 2; hb++) {
              const int b = 2 * pr + hb;
              {
                ulonglong2 wf;
                if constexpr (VT) {
                  wf.x = ((unsigned long long)tx1[2 * C + hb] << 32) | tx0[2 * C + hb];
                  wf.y = ty0[2 * C + hb];
                } else {
                  wf = wf_next;
                  wf_next = wblk[b < bl ? b + 1 : bl];
                }
                v4u X[GA + GB];
#pragma unroll
                for (int t = 0; t < GA + GB; t++) {
                  if constexpr (CEN) {
                    X[t] = R[C][t] & (hb == 0 ? 0x33333333u : 0xCCCCCCCCu);
                  } else {
                    const uint32_t w0 = R[C][t][2 * hb], w1 = R[C][t][2 * hb + 1];
                    X[t] = v4u{w0 & 0x33333333u, (w0 >> 2) & 0x33333333u, w1 & 0x33333333u, (w1 >> 2) & 0x33333333u};
                  }
                }
                if (st & 2) {  // the class that ended with the previous block: dev += (w_c - w_{c+1}) * P_c
                  const v2f dl = v2f{pdelta, pdelta};
#pragma unroll
                  for (int p = 0; p < GP; p++) {
                    // (register pairs of the accumulator tuple as they are: a shuffle, not v2f{acc[2 i], acc[2 i + 1]}, which
                    // hipcc builds with two v_mov_b32 per pair)
                    tpg_static_for<8>([&](auto ii) {
                      constexpr int i = decltype(ii)::value;
                      dev[p][i] = __builtin_elementwise_fma(dl, __builtin_shufflevector(acc[p], acc[p], 2 * i, 2 * i + 1), dev[p][i]);
                    });
                  }
                }
                if (CEN && hb == 1) {
#pragma unroll
                  for (int p = 0; p < GP; p++) acc[p] = MFMA_G4S2_ODD(X[p / GB], X[GA + p % GB], acc[p]);
                } else {
#pragma unroll
                  for (int p = 0; p < GP; p++) acc[p] = MFMA_G4S2(X[p / GB], X[GA + p % GB], acc[p]);
                }
                st = 0;
                const uint32_t fl = (uint32_t)wf.x & 3u;
                // ONE scalar test on the common path (three blocks of four have no flag); the end of the K range is folded
                // after the loop, not found by a comparison in every block
                if (__builtin_expect(fl != 0u, 0)) {
                if (fl & 2u) {  // group end: the FP64 fold, then everything starts from zero
                  const double w = __longlong_as_double((long long)(wf.x & ~3ull));
                  // four rounds of 8 elements of a register tile and 8 of an LDS tile: the LDS reads of a round are in
                  // flight while its register elements are folded, and a round needs 16 temporaries, not more
                  tpg_static_for<4>([&](auto hh) {
                    constexpr int h = decltype(hh)::value, pr_ = h >> 1, pl_ = 2 + (h >> 1), e0 = (h & 1) * 8;
                    double t[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) t[e] = olds[((pl_ - 2) * 16 + e0 + e) * 64];
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                      const int i = e0 + e;
                      o[pr_][i] = __builtin_fma((double)acc[pr_][i], w, o[pr_][i]) + (double)dev[pr_][i >> 1][i & 1];
                      acc[pr_][i] = 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                      const int i = e0 + e;
                      olds[((pl_ - 2) * 16 + i) * 64] = __builtin_fma((double)acc[pl_][i], w, t[e]) + (double)dev[pl_][i >> 1][i & 1];
                      acc[pl_][i] = 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) dev[pr_][(e0 >> 1) + e] = dev[pl_][(e0 >> 1) + e] = v2f{0.f, 0.f};
                    __builtin_amdgcn_sched_barrier(0);
                  });
                } else {
                  pdelta = __uint_as_float((uint32_t)wf.y);
                  st = 2;
                }
                }
              }
            }
          }
        });
      }
      // the end of the range: whatever the sums hold belongs to the class of the last block (a class end still pending there is
      // part of it: folding with w_c itself needs no difference; after a group end the sums are zero and this adds nothing)
      {
        const double w = __longlong_as_double((long long)(wlast & ~3ull));
        tpg_static_for<4>([&](auto hh) {
          constexpr int h = decltype(hh)::value, pr_ = h >> 1, pl_ = 2 + (h >> 1), e0 = (h & 1) * 8;
          double t[8];
#pragma unroll
          for (int e = 0; e < 8; e++) t[e] = olds[((pl_ - 2) * 16 + e0 + e) * 64];
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const int i = e0 + e;
            o[pr_][i] = __builtin_fma((double)acc[pr_][i], w, o[pr_][i]) + (double)dev[pr_][i >> 1][i & 1];
          }
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const int i = e0 + e;
            olds[((pl_ - 2) * 16 + i) * 64] = __builtin_fma((double)acc[pl_][i], w, t[e]) + (double)dev[pl_][i >> 1][i & 1];
          }
        });
      }
    }
    double* slab = slabs + ((int64_t)ks * nun + u) * GCLS_SLAB + lane;
#pragma unroll
    for (int p = 0; p < GP; p++)
#pragma unroll
      for (int i = 0; i < 16; i++) slab[(p * 16 + i) * 64] = p < 2 ? o[p][i] : olds[((p - 2) * 16 + i) * 64];
  }
}

// 3c. the mixed-precision fold at ONE wave per SIMD (TPG_GRAM_KERNEL=14 / 34; the A/B of DESIGN.md 3.2, not a path anything
// takes): what two waves per SIMD buy the kernel above is that one wave's loads and folds hide behind the other's MFMAs.
// Here a wave has the whole register file of its SIMD and nobody to hide behind, so everything is explicit:
//   * operands go through NS rotating register slots (one 16-byte load per lane = a PAIR of blocks), fetched NS - 1 pairs
//     ahead of their use; the prologue issues them slot by slot (sched_barrier) so that hipcc's wait-count pass, which merges
//     the loop entry with the back edge, does not put a vmcnt(0) at the top of every loop body (pairwise.hip);
//   * group ends fall between two bodies of the unrolled loop by construction (tpg_gcls_group_pad_kernel pads every group to
//     a multiple of 2 NS blocks), so the FP64 fold exists once, not in every step;
//   * the FP64 result lives in LDS (32 KiB per wave, 128 KiB per workgroup, one workgroup per CU): touched at group
//     ends only, and an LDS read-modify-write is cheaper than moving 128 registers through AGPRs;
//   * the block table comes through ONE vector load per body, issued a body ahead and taken apart into SGPRs
//     (v_readlane_b32) at the top of the body.  (Scalar loads return out of order, so each use is an s_waitcnt lgkmcnt(0):
//     with one wave per SIMD a full scalar-load latency per block -- 1 750 cycles per block, 44 ms, measured; a load under a
//     condition, or a loaded register copied into another, makes hipcc drain every load in flight.)
//   * the lean scalar stream of the two-waves kernel: 32-bit indices, fixed tile bases in SGPRs + one lane offset that
//     advances by a pair, no range tests (the padded layout ends on a whole body, T2g has a slack of NS pairs behind it),
//     flags and class differences in SGPRs; the centred operand layout (CEN);
//   * IL: a step written tile by tile -- MFMA of tile p, then the operand words of fragment p of the NEXT block (4 v_and_b32)
//     -- and pinned in that order (sched_barrier + an empty asm on the words: left alone hipcc makes them in the next block,
//     in front of its four MFMAs, and the pipe idles while the wave issues VALU); the class end of the block before
//     (32 v_pk_fma_f32) sits in front of the step behind ONE scalar branch, and only `dev` changes behind it -- with the
//     MFMAs inside two variants of the step hipcc gave each variant its own accumulators, ran BOTH sets of MFMAs and merged
//     them with 32 v_mov_b64 per block (20.7 ms, 1.65 x the MFMAs by SQ_INSTS_MFMA).
// Measured at 5 000 x 1 000 000 in one job with the two-waves kernel (9.13 ms): 9.36 ms interleaved, 9.93 ms not, whatever
// the slots (3 / 4 / 5) and the K split (4 ... 24); round 3's first form of it (100 instructions per block) 12.8 - 13.4.
// PMC: 7.9 VALU-class instructions per MFMA, the wave issues VALU 55 % of its cycles, waits (vmcnt) 26 %, everything else
// 19 %; MFMA pipe 42 % busy -- the figure of the two-waves kernel -- L2 hit rate 77 % against 85 % (22.8 GB HBM-side).
// (The same pipelined kernel at TWO waves per SIMD, its FP64 result half in LDS and half read-modify-written in the unit's
// slab: 13.9 ms, 60 GB of slab traffic; DESIGN.md 3.2.  Two lessons from it live on as comments in the kernels: an address
// that went through an integer is FLAT to hipcc unless cast to address_space(1), and ONE flat access in a loop turns every
// vmcnt wait of the loop into vmcnt(0); 64-bit per-lane addresses get hoisted out of loops and spill.)
// Slabs and the assemble pass are those of the kernel above; K ranges start on multiples of NS pairs.
#define GCLS3_LDS_BYTES (4 * GP * 16 * 64 * 8)
template <int NS, bool CEN, bool IL>
__global__ __launch_bounds__(256, 1) void tpg_gcls_gram1w_kernel(const uint4* __restrict__ T2g, int64_t nblocks, int64_t rs2, int nrtv,
                                                                  const ulonglong2* __restrict__ wblk,
                                                                  const int2* __restrict__ order, int64_t nun, int S,
                                                                  double* __restrict__ slabs) {
  extern __shared__ double olds_raw[];  // [wave][tile][register][lane]
  constexpr int NT = GA + GB, BODY = 2 * NS;
  static_assert(NT == GP, "a step expands fragment p beside the MFMA of tile p");
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* olds = olds_raw + (size_t)wv * (GP * 16 * 64) + lane;
  const int bx = (int)blockIdx.x, xcd = bx & 7, cidx = bx >> 3, cpx = (int)gridDim.x >> 3;
  const int nbodies = (int)(nblocks / BODY);  // (the padded layout: every group, so the whole list, ends on a body)
  typedef __attribute__((address_space(1))) const char gchar;
  typedef __attribute__((address_space(1))) const v4u gv4u;
  for (int64_t round = 0;; round++) {
    const int64_t un = ((round * 8 + xcd) * cpx + cidx) * 4 + wv;
    if (un >= nun * S) break;
    const int ks = (int)(un / nun);
    const int64_t u = un % nun;
    const int2 ijv = order[u];
    const int2 ij = make_int2(__builtin_amdgcn_readfirstlane(ijv.x), __builtin_amdgcn_readfirstlane(ijv.y));
    const int p0 = __builtin_amdgcn_readfirstlane(NS * (int)(((int64_t)nbodies * ks) / S));
    const int p1 = __builtin_amdgcn_readfirstlane(NS * (int)(((int64_t)nbodies * (ks + 1)) / S));
    gchar* pt[NT];  // the unit's four row tiles at pair p0
#pragma unroll
    for (int t = 0; t < GA; t++) pt[t] = (gchar*)tpg_uniform64((int64_t)(T2g + ((int64_t)min(GA * ij.x + t, nrtv - 1) * rs2 + p0) * 64));
#pragma unroll
    for (int t = 0; t < GB; t++) pt[GA + t] = (gchar*)tpg_uniform64((int64_t)(T2g + ((int64_t)min(GB * ij.y + t, nrtv - 1) * rs2 + p0) * 64));
    v2f dev[GP][8];
    v16f acc[GP];
#pragma unroll
    for (int p = 0; p < GP; p++) {
#pragma unroll
      for (int i = 0; i < 16; i++) { acc[p][i] = 0.f; olds[(p * 16 + i) * 64] = 0.0; }
#pragma unroll
      for (int i = 0; i < 8; i++) dev[p][i] = v2f{0.f, 0.f};
    }
    // out += w * sums + dev; sums and dev start again from zero (four elements at a time: few temporaries)
    auto fold = [&](double w) {
#pragma unroll
      for (int p = 0; p < GP; p++)
        tpg_static_for<4>([&](auto qq) {
          constexpr int q = decltype(qq)::value;
          double t[4];
#pragma unroll
          for (int e = 0; e < 4; e++) t[e] = olds[(p * 16 + 4 * q + e) * 64];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int i = 4 * q + e;
            olds[(p * 16 + i) * 64] = __builtin_fma((double)acc[p][i], w, t[e]) + (double)dev[p][i >> 1][i & 1];
            acc[p][i] = 0.f;
          }
          dev[p][2 * q] = v2f{0.f, 0.f};
          dev[p][2 * q + 1] = v2f{0.f, 0.f};
          __builtin_amdgcn_sched_barrier(0);
        });
    };
    if (p0 < p1) {
      v4u R[NS][NT];
      uint32_t voff = (uint32_t)lane * 16u;  // this lane's 16 bytes of the NEXT pair to fetch, relative to pair p0
      auto LDP = [&](v4u (&slot)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; t++) slot[t] = *(gv4u*)(pt[t] + voff);
        voff += 1024u;
      };
      // lane l: the table entry of block first + l (three dwords: with a fourth, dead, register in the tuple hipcc parks the loads' lane offset in it and waits for the table load)
      typedef uint32_t v3u __attribute__((ext_vector_type(3)));
      typedef __attribute__((address_space(1))) const v3u gv3u;
      const uint32_t blast = (uint32_t)(nblocks - 1);
      gchar* const wb = (gchar*)tpg_uniform64((int64_t)wblk);
      auto LDT = [&](int first_block) {
        uint32_t bb = (uint32_t)first_block + (uint32_t)lane;
        bb = (bb < blast ? bb : blast) * 16u;
        return *(gv3u*)(wb + bb);
      };
      v3u TB = LDT(2 * p0);
      __builtin_amdgcn_sched_barrier(0);
      tpg_static_for<NS - 1>([&](auto dd) {
        LDP(R[decltype(dd)::value]);
        __builtin_amdgcn_sched_barrier(0);  // slot by slot (see above)
      });
      // operand words of one block: half hb of a slot
      auto expand = [&](const v4u& r, auto Hh) {
        constexpr int hb = decltype(Hh)::value;
        if constexpr (CEN) {
          return r & (hb == 0 ? 0x33333333u : 0xCCCCCCCCu);
        } else {
          const uint32_t w0 = r[2 * hb], w1 = r[2 * hb + 1];
          return v4u{w0 & 0x33333333u, (w0 >> 2) & 0x33333333u, w1 & 0x33333333u, (w1 >> 2) & 0x33333333u};
        }
      };
      v4u X[2][NT];
#pragma unroll
      for (int t = 0; t < NT; t++) X[0][t] = expand(R[0][t], std::integral_constant<int, 0>{});
      bool pend = false;   // the block before ended a class inside its group: dev += pdelta * sums before this block's MFMAs
      float pdelta = 0.f;  // w_c - w_{c+1}
      uint32_t fl[BODY], dl[BODY], wlo = 0, whi = 0;  // this body's table entries, in SGPRs

      auto step = [&](auto Cc, auto Hh) {
        constexpr int C = decltype(Cc)::value, hb = decltype(Hh)::value, cur = hb, nx = hb ^ 1;
        constexpr int M = (C + NS - 1) % NS;                        // the slot the pair before this one left
        constexpr int SN = hb == 0 ? C : (C + 1) % NS, HN = hb ^ 1;  // slot and half of the NEXT block
        if constexpr (hb == 0) LDP(R[M]);
        __builtin_amdgcn_sched_barrier(0);
        if (pend) {  // (one branch per block, and only `dev` changes behind it: with the MFMAs inside two variants of the step
                     // hipcc gave each variant its own accumulators and merged them with 32 v_mov_b64 per block)
          const v2f dlt = v2f{pdelta, pdelta};
#pragma unroll
          for (int p = 0; p < GP; p++)
            tpg_static_for<8>([&](auto ii) {
              constexpr int i = decltype(ii)::value;
              dev[p][i] = __builtin_elementwise_fma(dlt, __builtin_shufflevector(acc[p], acc[p], 2 * i, 2 * i + 1), dev[p][i]);
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        tpg_static_for<GP>([&](auto pp_) {
          constexpr int p = decltype(pp_)::value;
          if constexpr (CEN && hb == 1) acc[p] = MFMA_G4S2_ODD(X[cur][p / GB], X[cur][GA + p % GB], acc[p]);
          else acc[p] = MFMA_G4S2(X[cur][p / GB], X[cur][GA + p % GB], acc[p]);
          X[nx][p] = expand(R[SN][p], std::integral_constant<int, HN>{});
          if constexpr (IL) {
            asm volatile("" : "+v"(X[nx][p]));  // HERE, in the shadow of the MFMA (left alone, the words are made in the next block, where they are used)
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        pend = (fl[2 * C + hb] & 1u) != 0;
        pdelta = __uint_as_float(dl[2 * C + hb]);
      };
      for (int pp = p0; pp < p1; pp += NS) {
        // this body's table entries -> SGPRs, then the load of the next body's into the same register
#pragma unroll
        for (int k = 0; k < BODY; k++) {
          fl[k] = (uint32_t)__builtin_amdgcn_readlane((int)TB[0], k);
          dl[k] = (uint32_t)__builtin_amdgcn_readlane((int)TB[2], k);
        }
        wlo = fl[BODY - 1];
        whi = (uint32_t)__builtin_amdgcn_readlane((int)TB[1], BODY - 1);
        __builtin_amdgcn_sched_barrier(0);
        TB = LDT(2 * (pp + NS));
        __builtin_amdgcn_sched_barrier(0);
        tpg_static_for<NS>([&](auto cc) {
          step(cc, std::integral_constant<int, 0>{});
          step(cc, std::integral_constant<int, 1>{});
        });
        // a group (or GCLS3_GRUN blocks) ends with this body's last block: the FP64 fold, here and nowhere else
        if (__builtin_expect((wlo & 2u) != 0 && pp + NS < p1, 0)) {
          fold(__longlong_as_double((long long)(((uint64_t)whi << 32) | (wlo & ~3u))));
          pend = false;
        }
      }
      // the end of the range: whatever the sums hold belongs to the class of the last block (a class end pending there is
      // part of it: with w_c itself no difference is needed)
      {
        const ulonglong2 e = wblk[2 * (int64_t)p1 - 1];
        fold(__longlong_as_double((long long)(e.x & ~3ull)));
      }
    }
    double* slab = slabs + ((int64_t)ks * nun + u) * GCLS_SLAB + lane;
#pragma unroll
    for (int p = 0; p < GP; p++)
#pragma unroll
      for (int i = 0; i < 16; i++) slab[(p * 16 + i) * 64] = olds[(p * 16 + i) * 64];
  }
}

// 4. the S slabs of every unit, added in split order, into both triangles of K (n x n, column-major).  A wave takes one
// 32 x 32 tile of the unit: it sums the slabs in MFMA register order (512 contiguous bytes per wave and slab), turns the
// tile through LDS, and writes it twice with 32 lanes on 256 contiguous bytes -- K[i, k] along i and K[k, i] along k.
// (Written straight from register order, K[i + k n] had the lanes of a wave on 64 different columns: 8-byte pieces 40 KB
// apart, SQ_WAIT_ANY 93 %.)
__global__ __launch_bounds__(256) void tpg_gcls_assemble_kernel(const double* __restrict__ slabs, const int2* __restrict__ order,
                                                                int64_t nun, int S, int n, double* __restrict__ K) {
  __shared__ double tile[GP][32][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  static_assert(GP == 4, "one wave per tile of the unit");
  for (int64_t u = blockIdx.x; u < nun; u += gridDim.x) {
    const int2 ij = order[u];
    const int p = wv;
    const int ti = GA * ij.x + p / GB, tk = GB * ij.y + p % GB;
    if (tk < ti) continue;  // (wave-uniform; no barrier below: a wave only ever reads what it wrote)
    double v[16];
#pragma unroll
    for (int reg = 0; reg < 16; reg++) v[reg] = 0.0;
    for (int s = 0; s < S; s++) {
      const double* sl = slabs + ((int64_t)s * nun + u) * GCLS_SLAB + (p * 16) * 64 + lane;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) v[reg] += sl[reg * 64];
    }
#pragma unroll
    for (int reg = 0; reg < 16; reg++) tile[p][tpg_cd_row(reg, lane)][lane & 31] = v[reg];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int a = lane & 31, h = lane >> 5;
#pragma unroll 4
    for (int it = 0; it < 16; it++) {
      const int c = 2 * it + h;
      // K[i, k] with the lanes along i (rows of the tile), then K[k, i] with the lanes along k (its columns)
      const int i1 = 32 * ti + a, k1 = 32 * tk + c;
      if (i1 < n && k1 < n) K[i1 + (int64_t)k1 * n] = tile[p][a][c];
      const int i2 = 32 * ti + c, k2 = 32 * tk + a;
      if (i2 < n && k2 < n) K[k2 + (int64_t)i2 * n] = tile[p][c][a];
    }
    __builtin_amdgcn_wave_barrier();  // the tile is rewritten in the next round
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

// cost models (microseconds per wave), fitted on the two kernels at n = 5 000 (12.2 ms at 1 000 000 loci, 6.0 ms on a
// shard of 125 000 loci that holds as many classes): per tile 0.136 us per class (fold) + 0.079 us per block (expansion,
// MFMA, operand streaming); digit kernel 1.0 us per 128 loci.  S is even: the two halves of the grid take S / 2 splits each.
// Which fold: the FP64 fold of every class (tpg_gcls_gram_kernel) when the classes are long -- below 0.15 classes per block
// the group ends of the mixed-precision kernel (three times a plain fold each) cost more than its cheap class ends save:
// 0.28 against 0.35 ms at 1 000 x 650 000 (949 classes, 10 634 blocks), 12.1 against 11.3 ms at 5 000 x 1 000 000 (4 728
// classes, 17 888 blocks) -- or when TPG_GRAM_FOLD64 asks for it (TPG_GRAM_FOLD64=0: never).
static bool gcls_fold64(int64_t nruns, int64_t nblocks) {
  const char* e = getenv("TPG_GRAM_FOLD64");
  if (e) return atoi(e) != 0;
  return (double)nruns < 0.15 * (double)nblocks;
}

static double gcls_cost_classes(int64_t nunits, int64_t nruns, int64_t nblocks, int nwaves, int* bestS) {
  double best = -1;
  const bool f64 = gcls_fold64(nruns, nblocks);
  const double t_class = f64 ? 0.136 : GCLS2_T_CLASS, t_block = f64 ? 0.079 : GCLS2_T_BLOCK;
  for (int S = 2; S <= 32; S += 2) {
    if (nblocks / S < 4 && S > 2) break;
    const int64_t rounds = ceil_div(nunits * S, (int64_t)nwaves);
    const double per = ((double)nruns / S + 1.0) * GP * t_class + (double)ceil_div(nblocks, (int64_t)S) * GP * t_block + 6.0;
    // + the assemble pass over the S slabs of every unit (0.48 ms at S = 18 and 3 160 units)
    const double cost = (double)rounds * per + (double)S * (double)nunits * 8.5e-3;
    if (best < 0 || cost < best * 0.995) { best = cost; *bestS = S; }
  }
  return best;
}

static int gram_classes_core(tpg_ctx* ctx, int64_t n, int64_t Q, int64_t m, const GclsSrc& src, const double* d_w, double* d_what,
                             double* d_K, bool force, bool* done, bool centred_ok) {
  *done = false;
  if (m >= (1ll << 31) - 64 || (getenv("TPG_GRAM_DIGITS") && !force)) return TPG_OK;
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
  // TPG_GRAM_KERNEL=14 / 34: the one-wave-per-SIMD form of the mixed fold (tpg_gcls_gram1w_kernel, four operand slots; 34 =
  // steps interleaved tile by tile; a measured A/B: 9.9 / 9.4 ms against 9.1): every group of classes is padded to whole bodies
  // of its unrolled loop.  TPG_GRAM_KERNEL=1: the two-waves kernel with its block table by scalar loads.
  const int kern3 = getenv("TPG_GRAM_KERNEL") ? atoi(getenv("TPG_GRAM_KERNEL")) : 0;
  const bool lean_il = kern3 == 34, lean1 = kern3 == 14 || lean_il;
  const int body = lean1 ? 8 : 0;
  const bool one_wave = lean1;
  const int gmax = getenv("TPG_GRAM_GMAX") ? std::max(1, atoi(getenv("TPG_GRAM_GMAX"))) : GCLS_GMAX;
  const int gq = getenv("TPG_GRAM_GQ") ? std::min(40, std::max(1, atoi(getenv("TPG_GRAM_GQ")))) : GCLS_GQ;
  uint32_t* d_sblk = nullptr;
  uint8_t* d_ge = nullptr;
  TPG_HIP(B.get(&d_sblk, (size_t)m)); TPG_HIP(B.get(&d_ge, (size_t)m));
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
    // blocks reserved per class: its own, plus (one-wave kernel) the padding that ends its group on a whole body
    hipLaunchKernelGGL(tpg_gcls_group_pad_kernel, dim3(1024), dim3(256), 0, ctx->stream, (const unsigned long long*)d_ukeys,
                       (const int*)d_nruns, m, (const uint32_t*)d_nblk, gmax, gq, body, d_sblk, d_ge);
    t = t_bytes;
    TPG_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, t, d_sblk, d_bstart, (int)m, ctx->stream));
    hipLaunchKernelGGL(tpg_gcls_totals_kernel, dim3(1), dim3(1), 0, ctx->stream, (const int*)d_nruns,
                       (const uint32_t*)d_bstart, (const uint32_t*)d_sblk, d_totals);
  }
  TPG_HIP(tpg_fetch_small(ctx, totals, d_totals, sizeof(totals)));
  const int64_t nruns = totals[0], nblocks = totals[1];
  TPG_REQUIRE(nruns > 0 && nblocks > 0, TPG_EHIP, "class table is empty");

  // units (I, J) that hold at least one wanted pair of row tiles (J tile >= I tile), in patch order: blocks of 8 J,
  // inside a block row after row, so that the run of units an XCD takes in one round shares its row tiles
  const int nrtv = (int)ceil_div(n, 32);
  const int nI = (int)ceil_div(nrtv, GA), nJ = (int)ceil_div(nrtv, GB);
  std::vector<int2> order;
  const int PJ = 8;
  for (int pj = 0; pj * PJ < nJ; pj++) {
    const int j1 = std::min(nJ, pj * PJ + PJ);
    for (int I = 0; I < nI; I++)
      for (int J = pj * PJ; J < j1; J++)
        if (GB * J + GB - 1 >= GA * I) order.push_back(make_int2(I, J));
  }
  const int64_t nun = (int64_t)order.size();
  int ncu8 = ctx->num_cu / 8 * 8;
  if (ncu8 < 8) ncu8 = 8;
  const bool f64 = body ? false : gcls_fold64(nruns, nblocks);
  // the centred operand layout (tpg_gcls_gather_kernel<true>): the default two-waves mixed-fold kernel only, and only where
  // the caller double-centres the result (the omitted terms are r_i + r_k + const); TPG_GRAM_CENTER=0: never (A/B)
  static const bool no_cen = getenv("TPG_GRAM_CENTER") && atoi(getenv("TPG_GRAM_CENTER")) == 0;
  const bool cen = centred_ok && !no_cen && !f64 && (!body || lean1) && kern3 != 1;
  const int nblk_grid = one_wave ? ncu8 : 2 * ncu8;  // two workgroups per CU = two waves per SIMD
  const int nwaves = 4 * nblk_grid;
  int S = 2;
  // + the sort, the locus-major copy and the gather (1.25 us per 1000 loci at n = 5 000: they scale with n m)
  const double cost_cls = gcls_cost_classes(nun, nruns, nblocks, nwaves, &S) + 350.0 + 1.25e-3 * (double)m * ((double)n / 5000.0);
  // the digit kernel: 32 x 128 wave tiles, 64 int8 MFMAs (~1.0 us) per 128 loci, 4 row tiles x super-tiles of 4
  const int64_t nun_dig = (int64_t)nrtv * ceil_div((int64_t)nrtv, 4) / 2 + nrtv;
  // (1.0 us is its four digits of a weight kept to 22 fractional bits; a caller that asks for more bits, pca.hip, pays per digit)
  const int t_dig = std::max(4, ((int)ceil(log2(2.0 * (double)n + 1.0)) + ctx->pca_digit_fbits + 7) / 7);
  const double cost_dig = (double)ceil_div(nun_dig, (int64_t)(4 * ncu8)) * ((double)ceil_div(m, 128) * 0.25 * t_dig) + 65.0;
  if (getenv("TPG_GRAM_S")) S = std::max(2, atoi(getenv("TPG_GRAM_S")) & ~1);  // (experiments)
  while ((nblocks / 2 + S) / S + 2 >= (1 << 21)) S += 2;  // a K range stays below 2^21 block pairs (32-bit lane offsets)
  if (getenv("TPG_DEBUG"))
    fprintf(stderr, "[tpg] gram classes: %lld classes, %lld blocks for %lld loci, S = %d, model %.0f us (digits %.0f us)\n",
            (long long)nruns, (long long)nblocks, (long long)m, S, cost_cls, cost_dig);
  if (cost_cls > cost_dig && !force && !getenv("TPG_GRAM_CLASSES")) return TPG_OK;

  int32_t* d_src = nullptr;
  double* d_slabs = nullptr;
  unsigned long long* d_wblk = nullptr;
  ulonglong2* d_wblk2 = nullptr;
  uint4* d_T2g = nullptr;
  int2* d_order = nullptr;
  TPG_HIP(B.get(&d_src, (size_t)nblocks * 64));
  if (f64) TPG_HIP(B.get(&d_wblk, (size_t)nblocks));
  else TPG_HIP(B.get(&d_wblk2, (size_t)nblocks + 4));
  const int64_t rs2 = (nblocks + 1) / 2;  // row-tile stride of T2g: a uint4 per lane and PAIR of blocks
  TPG_HIP(B.get(&d_T2g, (size_t)(4 * Q) * (size_t)rs2 * 64 + 8 * 64));  // (+ 8 pairs: tpg_gcls_gram1w_kernel fetches NS - 1 pairs past a range)
  TPG_HIP(B.get(&d_order, (size_t)nun));
  TPG_HIP(B.get(&d_slabs, (size_t)S * (size_t)nun * GCLS_SLAB));
  TPG_HIP(tpg_h2d_async(ctx, d_order, order.data(), sizeof(int2) * (size_t)nun));
  TPG_HIP(hipMemsetAsync(d_src, 0xFF, sizeof(int32_t) * (size_t)nblocks * 64, ctx->stream));
  {
    ProfScope ps(ctx, "gcls_layout");
    hipLaunchKernelGGL(tpg_gcls_place_kernel, dim3(1024), dim3(256), 0, ctx->stream, (const uint32_t*)d_idx2,
                       (const unsigned long long*)d_ukeys, (const uint32_t*)d_estart, (const uint32_t*)d_bstart, (int)nruns, m,
                       d_src, d_what);
    if (f64)
      hipLaunchKernelGGL(tpg_gcls_block_table_kernel, dim3(256), dim3(256), 0, ctx->stream, (const unsigned long long*)d_ukeys,
                         (const uint32_t*)d_bstart, (const uint32_t*)d_nblk, (int)nruns, nblocks, d_wblk);
    else if (body)
      hipLaunchKernelGGL(tpg_gcls_block_table3_kernel, dim3(256), dim3(256), 0, ctx->stream, (const unsigned long long*)d_ukeys,
                         (const uint32_t*)d_bstart, (const uint32_t*)d_nblk, (const uint32_t*)d_sblk, (const uint8_t*)d_ge,
                         (int)nruns, nblocks, d_wblk2);
    else
      hipLaunchKernelGGL(tpg_gcls_block_table2_kernel, dim3(256), dim3(256), 0, ctx->stream, (const unsigned long long*)d_ukeys,
                         (const uint32_t*)d_bstart, (const uint32_t*)d_nblk, (int)nruns, nblocks, gmax, gq, d_wblk2);
  }
  {
    const int64_t tasks = Q * rs2;
    if (tasks >= ((int64_t)1 << 31)) return TPG_OK;  // (not done: the caller's digit kernel; 2^31 tasks are 2^43 genotypes)
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(tasks, 4), (int64_t)ctx->num_cu * 16);
    // chunks of 128 individuals per task: a lane reads 32 QW contiguous bytes of its locus (0.69 / 0.61 / 0.60 ms for 1 / 2 / 4
    // at 5 000 x 1 000 000: fewer, longer random reads); TPG_GATHER_QW for the comparison
    const int qw = getenv("TPG_GATHER_QW") ? atoi(getenv("TPG_GATHER_QW")) : Q >= 4 ? 4 : Q >= 2 ? 2 : 1;
#define GATHER_GO(C, W) TPG_LAUNCH(ctx, "gcls_gather", (tpg_gcls_gather_kernel<C, W>), dim3(grid), dim3(256), 0, src, Q, (const int32_t*)d_src, nblocks, rs2, d_T2g)
    if (cen) { if (qw == 4) GATHER_GO(true, 4); else if (qw == 2) GATHER_GO(true, 2); else GATHER_GO(true, 1); }
    else { if (qw == 4) GATHER_GO(false, 4); else if (qw == 2) GATHER_GO(false, 2); else GATHER_GO(false, 1); }
#undef GATHER_GO
  }
  if (f64)
    TPG_LAUNCH(ctx, "pca_gram_classes", tpg_gcls_gram_kernel, dim3((unsigned)nblk_grid), dim3(256), 0, (const uint4*)d_T2g, nblocks,
               rs2, nrtv, (const unsigned long long*)d_wblk, (const int2*)d_order, nun, S, d_slabs);
  else if (body) {
    auto launch1 = [&](auto ns, auto cn, auto il) {
      constexpr int NS = decltype(ns)::value;
      constexpr bool CN = decltype(cn)::value, IL = decltype(il)::value;
      (void)hipFuncSetAttribute((const void*)tpg_gcls_gram1w_kernel<NS, CN, IL>, hipFuncAttributeMaxDynamicSharedMemorySize, GCLS3_LDS_BYTES);
      TPG_LAUNCH(ctx, "pca_gram_classes", (tpg_gcls_gram1w_kernel<NS, CN, IL>), dim3((unsigned)nblk_grid), dim3(256), GCLS3_LDS_BYTES,
                 (const uint4*)d_T2g, nblocks, rs2, nrtv, (const ulonglong2*)d_wblk2, (const int2*)d_order, nun, S, d_slabs);
    };
    TPG_REQUIRE(nblocks % body == 0, TPG_EHIP, "class layout does not end on a body");
    const std::integral_constant<int, 4> ns;
    if (cen) { if (lean_il) launch1(ns, std::true_type{}, std::true_type{}); else launch1(ns, std::true_type{}, std::false_type{}); }
    else { if (lean_il) launch1(ns, std::false_type{}, std::true_type{}); else launch1(ns, std::false_type{}, std::false_type{}); }
  } else {
    // TPG_GRAM_KERNEL=1: the block table by scalar loads, one per block (rounds 2 and 3; A/B)
    if (kern3 == 1) {
      (void)hipFuncSetAttribute((const void*)tpg_gcls_gram2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GCLS2_LDS_BYTES);
      TPG_LAUNCH(ctx, "pca_gram_classes", tpg_gcls_gram2_kernel<false>, dim3((unsigned)nblk_grid), dim3(256), GCLS2_LDS_BYTES,
                 (const uint4*)d_T2g, nblocks, rs2, nrtv, (const ulonglong2*)d_wblk2, (const int2*)d_order, nun, S, d_slabs);
    } else if (cen) {
      (void)hipFuncSetAttribute((const void*)tpg_gcls_gram2_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GCLS2_LDS_BYTES);
      TPG_LAUNCH(ctx, "pca_gram_classes", (tpg_gcls_gram2_kernel<true, true>), dim3((unsigned)nblk_grid), dim3(256), GCLS2_LDS_BYTES,
                 (const uint4*)d_T2g, nblocks, rs2, nrtv, (const ulonglong2*)d_wblk2, (const int2*)d_order, nun, S, d_slabs);
    } else {
      (void)hipFuncSetAttribute((const void*)tpg_gcls_gram2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GCLS2_LDS_BYTES);
      TPG_LAUNCH(ctx, "pca_gram_classes", tpg_gcls_gram2_kernel<true>, dim3((unsigned)nblk_grid), dim3(256), GCLS2_LDS_BYTES,
                 (const uint4*)d_T2g, nblocks, rs2, nrtv, (const ulonglong2*)d_wblk2, (const int2*)d_order, nun, S, d_slabs);
    }
  }
  TPG_LAUNCH(ctx, "gcls_assemble", tpg_gcls_assemble_kernel, dim3((unsigned)std::min<int64_t>(nun, 4096)), dim3(256), 0,
             (const double*)d_slabs, (const int2*)d_order, nun, S, (int)n, d_K);
  TPG_CHECK_LAUNCH();
  *done = true;
  return TPG_OK;  // the scratch blocks go back to the pool in stream order (GclsBufs)
}

// The gather wants a locus' 2 Q pieces next to each other: in the L layout they are 1 KiB apart, every 16-byte piece the
// quarter of a 64-byte sector whose other three quarters belong to other loci (5 GB fetched for 1.25 GB used at
// 5 000 x 1 000 000).  So the view is first copied LOCUS-MAJOR (LM: locus j = 2 Q consecutive pieces, q-major, then h) by a
// streaming transpose -- a workgroup takes the blocks (lt, q0 .. q0 + 3), 4 KiB, and writes 128 contiguous bytes per
// locus -- and the gather reads 32 contiguous bytes per (locus, q) whose sector neighbours are the next q of the same
// locus, wanted by the next task: 0.5 + 0.7 ms instead of 2.1.
__global__ __launch_bounds__(256) void tpg_gcls_l2lm_kernel(const uint4* __restrict__ L, int64_t Q, int64_t n_lt,
                                                            uint4* __restrict__ LM) {
  __shared__ uint4 sh[32][8];  // [locus in tile][2 (q - q0) + h]
  const int64_t QG = (Q + 3) / 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t task = blockIdx.x; task < n_lt * QG; task += gridDim.x) {
    const int64_t lt = task / QG, q0 = (task % QG) * 4;
    if (q0 + wv < Q) sh[lane & 31][2 * wv + (lane >> 5)] = L[(lt * Q + q0 + wv) * 64 + lane];
    __syncthreads();
    const int r = threadIdx.x >> 3, p = threadIdx.x & 7;
    if (q0 + (p >> 1) < Q) LM[((lt * 32 + r) * Q + q0) * 2 + p] = sh[r][p];
    __syncthreads();
  }
}

int tpg_gram_classes(tpg_ctx* ctx, const tpg_view* v, const double* d_w, double* d_what, double* d_K, bool* done,
                     bool centred_ok) {
  *done = false;
  if (getenv("TPG_GRAM_DIGITS")) return TPG_OK;
  const int64_t n_lt = 4 * v->KG;
  uint4* d_LM = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_LM, sizeof(uint4) * (size_t)n_lt * 32 * (size_t)v->Q * 2));
  TPG_LAUNCH(ctx, "gcls_l2lm", tpg_gcls_l2lm_kernel, dim3((unsigned)std::min<int64_t>(n_lt * ((v->Q + 3) / 4), (int64_t)ctx->num_cu * 32)),
             dim3(256), 0, (const uint4*)v->L, v->Q, n_lt, d_LM);
  const GclsSrc src{d_LM, 0, 2 * v->Q, 0, 0, 2, 1};
  const int rc = gram_classes_core(ctx, v->n, v->Q, v->m, src, d_w, d_what, d_K, false, done, centred_ok);
  tpg_pfree(d_LM);  // stream-ordered
  return rc;
}

// ---------------------------------------------------------------------------
// Whole weight classes per rank (SNP-block shards over several GPUs).
//
// A rank's shard of the loci holds a slice of EVERY weight class, so the fold cost of the class path (one fold per
// class and tile) does not shrink with the shard: 6.0 ms on a 125 000-locus shard of the 5 000 x 1 000 000 panel against
// 12.2 ms for the whole panel.  But S' = sum_c w_c G_c is a sum over classes as much as over loci: if rank r holds ALL
// loci of the classes it owns, it pays 1 / R of the folds AND 1 / R of the blocks.  So the packed genotype columns travel
// once (32 Q + 16 bytes per locus = 1 296 B at n = 5 000; 140 MB per rank at 8 ranks) by one all-to-all:
//   1. key_j = min(c_j, 2n - c_j) of the locus' allele count c_j (the binomial weight is a function of it); histogram of
//      the keys over all ranks (one int32 all-reduce of n + 1 bins);
//   2. the keys are cut into R contiguous ranges of equal modelled cost (the same cut on every rank);
//   3. every rank sorts its loci by destination and packs a record per locus: the 2 Q 16-byte pieces of its column of the
//      L layout, then its weight;  the counts go round (R x R int32), then the records (tpg_comm_alltoallv64);
//   4. the class Gram of the records received (gram_classes_core with the records as the source of the gather).
// The caller sums the S' of the ranks (the all-reduce it does anyway).  The decision to go this way is taken from the
// global histogram, i.e. alike on every rank.
#define GCLX_REC_WORDS(Q) (4 * (Q) + 2)  // 8-byte words per record: 32 Q bytes of column + weight + pad

__global__ void tpg_gclx_keys_kernel(const int32_t* __restrict__ counts, int64_t m, int n, int32_t* __restrict__ key,
                                     int32_t* __restrict__ hist) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
    const int c = counts[4 * j + 1] + 2 * counts[4 * j + 2];
    const int k = min(c, 2 * n - c);
    key[j] = k;
    atomicAdd(&hist[k], 1);
  }
}

__global__ void tpg_gclx_dest_kernel(const int32_t* __restrict__ key, int64_t m, const uint8_t* __restrict__ owner,
                                     uint32_t* __restrict__ dest, uint32_t* __restrict__ idx, int32_t* __restrict__ per_dest) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t d = owner[key[j]];
    dest[j] = d;
    idx[j] = (uint32_t)j;
    atomicAdd(&per_dest[d], 1);
  }
}

// record i <- locus order[i]: one wave copies the 2 Q pieces of the column (L layout) and the weight
__global__ __launch_bounds__(256) void tpg_gclx_pack_kernel(const uint4* __restrict__ L, int64_t Q, const uint32_t* __restrict__ order,
                                                            const double* __restrict__ scale, int64_t m,
                                                            uint4* __restrict__ rec) {
  const int lane = threadIdx.x & 63;
  const int64_t recq = 2 * Q + 1;  // uint4 per record
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < m; i += (int64_t)gridDim.x * 4) {
    const uint32_t j = order[i];
    const uint4* p = L + ((int64_t)(j >> 5) * Q) * 64 + (j & 31);
    for (int64_t t = lane; t < 2 * Q; t += 64) rec[i * recq + t] = p[(t >> 1) * 64 + (t & 1) * 32];
    if (lane == 0) {
      const double sc = scale[j], w = 1.0 / (sc * sc);
      rec[i * recq + 2 * Q] = make_uint4((uint32_t)__double_as_longlong(w), (uint32_t)(__double_as_longlong(w) >> 32), 0u, 0u);
    }
  }
}

__global__ void tpg_gclx_weights_kernel(const uint4* __restrict__ rec, int64_t Q, int64_t m, double* __restrict__ w) {
  const int64_t recq = 2 * Q + 1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 t = rec[i * recq + 2 * Q];
    w[i] = __longlong_as_double((long long)(((unsigned long long)t.y << 32) | t.x));
  }
}

int tpg_gram_classes_exchanged(tpg_ctx* ctx, tpg_comm* comm, const tpg_view* v, const int32_t* d_counts, const double* d_scale,
                               double* d_K, bool* done) {
  *done = false;
  const int R = comm->nranks, me = comm->rank;
  const int64_t n = v->n, m = v->m, Q = v->Q;
  if (R < 2 || R > 255 || n >= (1 << 30) || getenv("TPG_GRAM_NO_EXCHANGE")) return TPG_OK;
  GclsBufs B;
  int32_t *d_key = nullptr, *d_hist = nullptr, *d_perdest = nullptr;
  uint8_t* d_owner = nullptr;
  uint32_t *d_dest = nullptr, *d_dest2 = nullptr, *d_idx = nullptr, *d_idx2 = nullptr;
  void* d_tmp = nullptr;
  size_t t_sort = 0;
  std::vector<int32_t> hist((size_t)n + 1, 0);
  // rank-local steps first, then the ranks agree on a status before the first exchange (nobody is left waiting in it)
  auto local = [&]() -> int {
    TPG_HIP(B.get(&d_key, (size_t)std::max<int64_t>(m, 1)));
    TPG_HIP(B.get(&d_hist, (size_t)n + 1));
    TPG_HIP(B.get(&d_perdest, (size_t)R));
    TPG_HIP(B.get(&d_owner, (size_t)n + 1));
    TPG_HIP(B.get(&d_dest, (size_t)std::max<int64_t>(m, 1)));
    TPG_HIP(B.get(&d_dest2, (size_t)std::max<int64_t>(m, 1)));
    TPG_HIP(B.get(&d_idx, (size_t)std::max<int64_t>(m, 1)));
    TPG_HIP(B.get(&d_idx2, (size_t)std::max<int64_t>(m, 1)));
    TPG_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, t_sort, d_dest, d_dest2, d_idx, d_idx2, (int)m, 0, 8, ctx->stream));
    TPG_HIP(B.get((uint8_t**)&d_tmp, t_sort));
    TPG_HIP(hipMemsetAsync(d_hist, 0, sizeof(int32_t) * ((size_t)n + 1), ctx->stream));
    TPG_HIP(hipMemsetAsync(d_perdest, 0, sizeof(int32_t) * (size_t)R, ctx->stream));
    if (m > 0)
      hipLaunchKernelGGL(tpg_gclx_keys_kernel, dim3(512), dim3(256), 0, ctx->stream, d_counts, m, (int)n, d_key, d_hist);
    TPG_HIP(hipGetLastError());
    return TPG_OK;
  };
  TPG_TRY(tpg_comm_agree(comm, local()));
  if (!tpg_comm_alltoall_usable(comm)) return TPG_OK;  // the same answer on every rank
  {
    ProfScope ps(ctx, "gclx_histogram");
    TPG_TRY(tpg_comm_allreduce(comm, d_hist, n + 1, 0));
    TPG_HIP(hipMemcpyAsync(hist.data(), d_hist, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyDeviceToHost, ctx->stream));
    TPG_HIP(hipStreamSynchronize(ctx->stream));
  }
  // cost of a key = its class's blocks and fold; the same arithmetic on every rank
  int64_t m_all = 0, runs_all = 0, blocks_all = 0;
  for (int64_t k = 0; k <= n; k++)
    if (hist[(size_t)k] > 0) { m_all += hist[(size_t)k]; runs_all++; blocks_all += ((int64_t)hist[(size_t)k] + 63) / 64; }
  if (m_all == 0) return TPG_OK;
  const double cost_all = (double)runs_all * 0.136 + (double)blocks_all * 0.079;
  // does it pay?  modelled time of this rank's share of whole classes + the exchange (a rank sends (R - 1) / R of its records
  // over R - 1 xGMI links at once: ~300 GB/s per rank; a few launches) against the cheaper of the two local paths on a
  // shard (digits: 1.0 us per 128 loci and 4 row tiles)
  const int nrtv = (int)ceil_div(n, 32);
  const double units = (double)nrtv * (double)nrtv / 8.0, waves = 8.0 * (double)(ctx->num_cu > 8 ? ctx->num_cu : 8);
  const double t_exch = units / waves * 4.0 * cost_all / R + 700.0 + (double)m_all / R * (double)(32 * Q + 16) / 300e3;
  const double shard_blocks = (double)blocks_all / R + (double)runs_all;  // a shard pads every class to a block of its own
  const double t_cls_local = units / waves * 4.0 * ((double)runs_all * 0.136 + shard_blocks * 0.079) + 650.0;
  const double t_dig_local = ((double)nrtv * nrtv / 8.0 + nrtv) / (waves / 2.0) * ((double)m_all / R / 128.0) + 65.0;
  if (getenv("TPG_DEBUG") && me == 0)
    fprintf(stderr, "[tpg] gram exchange: %lld loci, %lld classes, %lld blocks over %d ranks: model %.0f us (local classes %.0f, digits %.0f)\n",
            (long long)m_all, (long long)runs_all, (long long)blocks_all, R, t_exch, t_cls_local, t_dig_local);
  if (t_exch > std::min(t_cls_local, t_dig_local) && !getenv("TPG_GRAM_EXCHANGE")) return TPG_OK;

  // contiguous key ranges of (nearly) equal cost
  std::vector<uint8_t> owner((size_t)n + 1, (uint8_t)(R - 1));
  {
    double acc = 0;
    int r = 0;
    for (int64_t k = 0; k <= n; k++) {
      owner[(size_t)k] = (uint8_t)r;
      if (hist[(size_t)k] > 0) acc += 0.136 + 0.079 * (double)(((int64_t)hist[(size_t)k] + 63) / 64);
      while (r < R - 1 && acc >= cost_all * (double)(r + 1) / R) r++;
    }
  }
  std::vector<int32_t> per_dest((size_t)R, 0);
  uint4 *d_send = nullptr, *d_recv = nullptr;
  double* d_w = nullptr;
  const int64_t recq = 2 * Q + 1;  // uint4 per record
  std::vector<size_t> scnt((size_t)R), soff((size_t)R), rcnt((size_t)R), roff((size_t)R);
  std::vector<int32_t> cm((size_t)R * R, 0);
  int64_t m_in = 0;
  auto pack = [&]() -> int {
    TPG_HIP(tpg_h2d_async(ctx, d_owner, owner.data(), (size_t)n + 1));
    if (m > 0) {
      ProfScope ps(ctx, "gclx_sort");
      hipLaunchKernelGGL(tpg_gclx_dest_kernel, dim3(512), dim3(256), 0, ctx->stream, (const int32_t*)d_key, m, (const uint8_t*)d_owner,
                         d_dest, d_idx, d_perdest);
      size_t t = t_sort;
      TPG_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp, t, d_dest, d_dest2, d_idx, d_idx2, (int)m, 0, 8, ctx->stream));
    }
    TPG_HIP(hipMemcpyAsync(per_dest.data(), d_perdest, sizeof(int32_t) * (size_t)R, hipMemcpyDeviceToHost, ctx->stream));
    TPG_HIP(hipStreamSynchronize(ctx->stream));
    TPG_HIP(B.get(&d_send, (size_t)std::max<int64_t>(m, 1) * (size_t)recq));
    if (m > 0)
      TPG_LAUNCH(ctx, "gclx_pack", tpg_gclx_pack_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(m, 4), (int64_t)ctx->num_cu * 16)),
                 dim3(256), 0, (const uint4*)v->L, Q, (const uint32_t*)d_idx2, d_scale, m, d_send);
    TPG_HIP(hipGetLastError());
    return TPG_OK;
  };
  TPG_TRY(tpg_comm_agree(comm, pack()));
  // who sends how many loci to whom: row `me` of an R x R table, summed over the ranks
  {
    int32_t* d_cm = nullptr;
    auto cm_up = [&]() -> int {  // rank-local (an allocation, a copy): agreed on before the all-reduce like the steps above
      TPG_HIP(B.get(&d_cm, (size_t)R * R));
      for (int d = 0; d < R; d++) cm[(size_t)me * R + d] = per_dest[(size_t)d];
      TPG_HIP(tpg_h2d_async(ctx, d_cm, cm.data(), sizeof(int32_t) * (size_t)R * R));
      return TPG_OK;
    };
    TPG_TRY(tpg_comm_agree(comm, cm_up()));
    TPG_TRY(tpg_comm_allreduce(comm, d_cm, (int64_t)R * R, 0));
    TPG_HIP(hipMemcpyAsync(cm.data(), d_cm, sizeof(int32_t) * (size_t)R * R, hipMemcpyDeviceToHost, ctx->stream));
    TPG_HIP(hipStreamSynchronize(ctx->stream));
  }
  {
    size_t so = 0, ro = 0;
    for (int r = 0; r < R; r++) {
      scnt[(size_t)r] = (size_t)cm[(size_t)me * R + r] * (size_t)recq * 2;  // 8-byte words
      soff[(size_t)r] = so;
      so += scnt[(size_t)r];
      rcnt[(size_t)r] = (size_t)cm[(size_t)r * R + me] * (size_t)recq * 2;
      roff[(size_t)r] = ro;
      ro += rcnt[(size_t)r];
      m_in += cm[(size_t)r * R + me];
    }
  }
  auto recv_alloc = [&]() -> int {
    TPG_HIP(B.get(&d_recv, (size_t)std::max<int64_t>(m_in, 1) * (size_t)recq));
    TPG_HIP(B.get(&d_w, (size_t)std::max<int64_t>(m_in, 1)));
    return TPG_OK;
  };
  TPG_TRY(tpg_comm_agree(comm, recv_alloc()));
  {
    ProfScope ps(ctx, "gclx_alltoall");
    TPG_TRY(tpg_comm_alltoallv64(comm, d_send, scnt.data(), soff.data(), d_recv, rcnt.data(), roff.data()));
  }
  // the class Gram of the loci this rank owns now (none: a zero matrix)
  if (m_in == 0) {
    TPG_HIP(hipMemsetAsync(d_K, 0, sizeof(double) * (size_t)n * (size_t)n, ctx->stream));
    *done = true;
    return TPG_OK;
  }
  hipLaunchKernelGGL(tpg_gclx_weights_kernel, dim3(512), dim3(256), 0, ctx->stream, (const uint4*)d_recv, Q, m_in, d_w);
  const GclsSrc src{d_recv, 0, recq, 0, 0, 2, 1};
  bool ok = false;
  TPG_TRY(gram_classes_core(ctx, n, Q, m_in, src, d_w, nullptr, d_K, true, &ok, true));  // (the caller double-centres: pca.hip)
  TPG_REQUIRE(ok, TPG_EHIP, "class Gram of the exchanged loci was not computed");
  *done = true;
  return TPG_OK;
}
