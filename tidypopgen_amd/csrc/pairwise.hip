// pairwise.hip -- N x N individual cross-products on the FP4 matrix cores (IBS / KING / allele sharing / GRM).
//
// Replaces the dense FP64 products of increment_ibs_counts (src/snp_ibs.cpp:67-72),
// increment_king_numerator (src/snp_king.cpp:70-72) and increment_as_counts (src/snp_as.cpp:64-65)
// and the R block loops around them (R/snp_ibs.R:69-82, R/snp_king.R:63-77,
// R/snp_allele_sharing.R:58-70).
//
// Per genotype three features: v (typed), d (dosage-1, 0 if missing), h (heterozygous).  With V = vv', D = dd',
// H = hh', A = hv' (A not symmetric) every count matrix of the reference is an integer combination (derivation in
// DESIGN.md):
//   IBS        = V + D + H          IBS_valid = 2 V
//   KING_num   = D - V + A + A'     N_Aa_i    = A
//   AS_num     = D                  AS_den    = V
// so the fused pass needs 3 symmetric + 1 general product = 2.5 N^2 M MACs instead of the
// reference's 12 N^2 M FP64 flops for IBS+KING+AS.
//
// The features are small enough for FP4 (E2M1: 0, 0.5, 1, 1.5, 2, 3, 4, 6 and their negatives), whose MFMA
// (v_mfma_scale_f32_32x32x64_f8f6f4) contracts 64 loci in the 32 cycles the int8 form needs for 32.  The view's T
// layout is re-coded once (tpg_t4_expand_kernel) with one NIBBLE per genotype whose bits ARE the three operand planes:
//   bit 0 = heterozygous            plane h = nibble & 0x1 -> FP4 0.5
//   bit 1 = typed                   plane v = nibble & 0x2 -> FP4 1.0
//   bit 2 = homozygous              plane d = nibble & 0xC -> FP4 +2.0 (dosage 2) / -2.0 (dosage 0: bit 3 = sign)
// so a plane of 8 loci costs ONE v_and_b32 (the int8 form of this kernel spent 19 VALU per 16 loci on v_perm lookups
// and was bound by VALU issue at 36.6 cycles per MFMA).  The E8M0 block scales of the instruction undo the 0.5 / 2.0
// (h x 2, d x 0.5), so every accumulator holds the plain integer count.  FP32 accumulation of integers is exact up to
// 2^24: a wave-unit never contracts more than 2^24 loci (tpg_pairwise_accumulate splits longer ranges), and what it
// adds to HBM is converted to int32 first.  tools/ubench_mfma_fp4.hip checks the instruction's exactness at
// accumulators next to +-2^24 and its rate (7.7 POP/s bare, 7.3 with three VALU per MFMA, at the 1.85 GHz the chip
// holds under it).
//
// Kernel (tpg_pairwise_kernel below): one wave owns a 96 x 32 tile of pairs -- super-tile I (row tiles 3I .. 3I+2)
// against column tile jt >= 3I -- and a K range; 5 products x 3 sub-tiles = 15 FP32 accumulator tiles (240 AGPRs),
// one wave per SIMD.  It streams its four operand row tiles from T4 with 16-B coalesced loads (1 KiB per wave
// instruction = 32 x 64 genotypes, prefetched two 128-locus groups ahead through rotating register slots), masks out
// the planes one K step ahead of their use and issues 15 MFMAs per 64 loci.  No LDS, no barriers.  Units are taken
// from a host-built table in XCD patch order.  Partial tiles are added to HBM with integer atomics (exact, order
// independent) into a tile-packed buffer in MFMA register order, so each atomic wave instruction is 256 contiguous
// bytes.  That buffer is what a multi-GPU run reduces.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "common.h"
#include <type_traits>
#include "devfrag.h"

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

#define TPG_NIB_V (TPG_T4_MV * 0x11111111u)
#define TPG_NIB_H (TPG_T4_MH * 0x11111111u)
#define TPG_NIB_D ((TPG_T4_MD | 8u) * 0x11111111u)
// E8M0 block scales (one byte per 32 contracted elements; all four bytes equal, so the byte select does not matter): the
// kernels' sc1 / sc2 / sch are the scales of the v / h / d planes (1, 2, 1/2 with the encoding of rounds 2 - 4)
#define TPG_SC_ONE TPG_T4_SC(TPG_T4_MV)
#define TPG_SC_TWO TPG_T4_SC(TPG_T4_MH)
#define TPG_SC_HALF TPG_T4_SC(TPG_T4_MD)

__device__ __forceinline__ v8i tpg_w8(v4i a) { return v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0}; }
// FP4 x FP4 (format code 4), 32 x 32 x 64
#define MFMA_F4(a, b, c, sa, sb) \
  __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(tpg_w8(a), tpg_w8(b), (c), 4, 4, 0, (sa), 0, (sb))

// T (2-bit codes) -> T4.  Every lane re-codes its own 16 bytes of a T block (4 dwords x 16 loci) as two 16-byte
// fragments of 32 nibbles: dwords 0, 1 -> block 2 kg, dwords 2, 3 -> block 2 kg + 1.  Which locus lands on which
// nibble of which lane half is immaterial -- the MFMA sums over all 64 -- as long as every row tile uses the same
// map, which it does.
__global__ __launch_bounds__(256) void tpg_t4_expand_kernel(const uint4* __restrict__ T, uint4* __restrict__ T4,
                                                            int64_t nblocks) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nblocks * 64; i += (int64_t)gridDim.x * 256) {
    const int64_t blk = i >> 6;
    const int lane = (int)(i & 63);
    const uint4 w = T[i];
    const uint32_t in[4] = {w.x, w.y, w.z, w.w};
    uint32_t out[8];
#pragma unroll
    for (int s = 0; s < 4; s++) tpg_t4_words(in[s], out[2 * s], out[2 * s + 1]);
    T4[(blk * 2) * 64 + lane] = make_uint4(out[0], out[1], out[2], out[3]);
    T4[(blk * 2 + 1) * 64 + lane] = make_uint4(out[4], out[5], out[6], out[7]);
  }
}

struct Frag3 {
  v4i v, d, h;
};

__device__ __forceinline__ Frag3 tpg_planes(v4u w, uint32_t mv, uint32_t md, uint32_t mh) {
  Frag3 f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    f.v[k] = (int)(w[k] & mv);
    f.d[k] = (int)(w[k] & md);
    f.h[k] = (int)(w[k] & mh);
  }
  return f;
}

// unit (I, jt): rows = super-tile I (row tiles TA I .. TA I + TA - 1), columns = row tile jt >= TA I.
// slab index u = TA (I nst - I (I-1)/2) + (jt - TA I); slab = [product][ta][reg][lane] int32.
constexpr int TA = TPG_PW_TA;
__host__ __device__ __forceinline__ int64_t tpg_pw_unit_index(int nst, int I, int jt) {
  return TA * ((int64_t)I * nst - ((int64_t)I * (I - 1)) / 2) + (jt - TA * I);
}

// The pairwise kernel.  One wave = one (I, jt) unit: a (32 TA) x 32 tile of pairs, five products, 5 TA accumulator
// tiles (240 AGPRs at TA = 3), one wave per SIMD.  Per 64 loci: TA + 1 fragments of four dwords -> 12 v_and_b32 each
// -> 5 TA MFMAs, i.e. 3.2 VALU per MFMA: at ~8 + 5.2 x (VALU per MFMA) cycles of issue per MFMA the
// wave stays under the 32 cycles of the MFMA itself, which the int8 form of this kernel (5.5 VALU per MFMA for half
// the loci) did not.  The plane masks of an A row tile past the data, or of a group past the K range, are zero
// (wave-uniform, so they sit in SGPRs and cost nothing).
// The masking is SOFTWARE-PIPELINED one K step ahead: an MFMA whose A/B operands were written by VALU
// instructions 0 / 1 / 2 MFMAs earlier takes 47.6 / 41.6 / 36.4 cycles instead of 32 (tools/ubench_mfma_dep.hip), so
// the fragments of K step s+1 are made in a second register set while the MFMAs of step s issue (one MFMA, then three
// or four VALU, enforced with sched_group_barrier).  Operand words are prefetched two 128-locus groups ahead through
// three rotating register slots.
#define SGB_MFMA 0x008
#define SGB_VALU 0x002
#define SGB_VMEM_READ 0x020
template <int NG>  // rotating load slots of one 128-locus group each: NG - 1 groups of prefetch (3 = rounds 2 and 3)
__global__ __launch_bounds__(256, 1) void tpg_pairwise_kernel(const uint4* __restrict__ T4, int64_t KG,
                                                                 int64_t kg_begin, int64_t kg_end, int nst, int nct,
                                                                 const int2* __restrict__ order, int64_t nun, int S,
                                                                 const int64_t* __restrict__ rowpad,
                                                                 int32_t* __restrict__ acc_out) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: unit, K range and bases in SGPRs
  const int64_t kgs = kg_end - kg_begin;
  // Work distribution.  `order` (host-built, pairwise_create) lists the (I, jt) units of the triangle in
  // PATCH order: blocks of 16 column tiles, inside a block row after row, so four consecutive entries nearly
  // always share the super-tile I (the four waves of a workgroup then fetch the A stream into the CU
  // once) and a run of 128 consecutive entries is ~8 rows x 16 columns.  Workgroups are dispatched round-robin
  // over the 8 XCDs (blockIdx % 8); in every round XCD x takes such a run for its 128 waves (same K range):
  // a few dozen distinct tiles per 128 loci instead of ~100 with a plain strided assignment, so the
  // re-reads hit that XCD's own L2.
  const int xcd = blockIdx.x & 7, cidx = blockIdx.x >> 3, cpx = gridDim.x >> 3;
  const int sc1 = TPG_SC_ONE, sc2 = TPG_SC_TWO, sch = TPG_SC_HALF;
  for (int64_t round = 0;; round++) {
    const int64_t un = ((round * 8 + xcd) * cpx + cidx) * 4 + wv;
    if (un >= nun * S) break;
    const int ks = (int)(un / nun);
    const int2 ijt = order[un % nun];
    const int I = __builtin_amdgcn_readfirstlane(ijt.x), jt = __builtin_amdgcn_readfirstlane(ijt.y);
    const int64_t tp0 = tpg_pw_unit_index(nst, I, jt) + rowpad[I];
    // 32-bit group indices (a view has fewer than 2^25 groups): scalar range tests, one s_add_i32 per step
    const int k0 = __builtin_amdgcn_readfirstlane((int)(kg_begin + (kgs * ks) / S));
    const int k1 = __builtin_amdgcn_readfirstlane((int)(kg_begin + (kgs * (ks + 1)) / S));

    // A row tiles past the last one with data (the last super-tile may be partial; the view holds 4 ceil(n / 128)
    // row tiles) read tile 0 instead and get zero plane masks: zero products
    // (fixed bases at group k0 in SGPRs; a block is one 32-bit lane offset: a wave-unit never spans 2^22 blocks)
    const char* pa[TA];
    bool there[TA];
#pragma unroll
    for (int t = 0; t < TA; t++) {
      there[t] = TA * I + t < nct;
      pa[t] = (const char*)(T4 + (((int64_t)(there[t] ? TA * I + t : 0) * KG + k0) * 2) * 64);
    }
    const char* pb0 = (const char*)(T4 + (((int64_t)jt * KG + k0) * 2) * 64);

    v16f cV[TA], cD[TA], cH[TA], cHV[TA], cVH[TA];
#pragma unroll
    for (int t = 0; t < TA; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) { cV[t][r] = 0.f; cD[t][r] = 0.f; cH[t][r] = 0.f; cHV[t][r] = 0.f; cVH[t][r] = 0.f; }

    if (k0 < k1) {
      const int kl = k1 - 1;
      // three rotating load slots (current group, next, two ahead): the K loop is unrolled by three so that no
      // slot is copied -- a copy at the end of a group waits for the loads issued at its start, which cuts the
      // prefetch distance to one group.  Groups past k1 run with zero A planes.
      // native vectors, not HIP's uint4 struct: with the struct the register allocator splits the loaded tuple right
      // after the load (v_mov behind an s_waitcnt vmcnt: a full memory latency at the top of every group)
      v4u RA[NG][TA][2], RB[NG][2];
      // wave-uniform base (SGPRs) + one 32-bit lane offset: global_load_dwordx4 v, v_off, s[base]; the empty asm keeps
      // hipcc from folding the lane into loop-invariant 64-bit VGPR pointers (a v_lshl_add_u64 per load)
      auto OFF = [&](int ig, int s) {  // block s of group ig, relative to group k0
        uint32_t off = (uint32_t)lane * 16u + (uint32_t)((ig - k0) * 2 + s) * 1024u;
        asm("" : "+v"(off));
        return off;
      };
#pragma unroll
      for (int g = 0; g < NG - 1; g++) {
        const int ig = k0 + g < k1 ? k0 + g : kl;
#pragma unroll
        for (int s = 0; s < 2; s++) {
          const uint32_t off = OFF(ig, s);
#pragma unroll
          for (int t = 0; t < TA; t++) RA[g][t][s] = *(const v4u*)(pa[t] + off);
          RB[g][s] = *(const v4u*)(pb0 + off);
          // block by block, in the order the loop issues them: hipcc's wait-count pass merges the loop entry with the back
          // edge, and with the two blocks of a group interleaved here the loop's second wait of every group became
          // vmcnt(4) instead of vmcnt(8) -- the prefetch one block shallower than the slots allow (see tpg_pairwise_set_kernel)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      Frag3 P[2][TA + 1];
#pragma unroll
      for (int t = 0; t < TA; t++)
        P[0][t] = tpg_planes(RA[0][t][0], there[t] ? TPG_NIB_V : 0u, there[t] ? TPG_NIB_D : 0u, there[t] ? TPG_NIB_H : 0u);
      P[0][TA] = tpg_planes(RB[0][0], TPG_NIB_V, TPG_NIB_D, TPG_NIB_H);
      // TAIL = false: a group of the main loop, whose groups kg and kg + 1 lie inside the K range: no plane mask depends on
      // the position (the masks of an absent A tile are loop-invariant SGPRs); TAIL = true: the last, partial body
      auto group = [&](auto Cc, auto Nn, auto Mm, auto Tl, int kg) {
        constexpr int C = decltype(Cc)::value, N = decltype(Nn)::value, M = decltype(Mm)::value;
        constexpr bool TAIL = decltype(Tl)::value;
        const int i2 = kg + NG - 1 < k1 ? kg + NG - 1 : kl;
        const bool live = !TAIL || kg < k1, live1 = !TAIL || kg + 1 < k1;
#pragma unroll
        for (int s = 0; s < 2; s++) {
          const int cur = s & 1, nx = cur ^ 1;
          // the loads of the group after next: the first block of every tile in step 0, the second in step 1
          const uint32_t off = OFF(i2, s);
#pragma unroll
          for (int t = 0; t < TA; t++) RA[M][t][s] = *(const v4u*)(pa[t] + off);
          RB[M][s] = *(const v4u*)(pb0 + off);
          // planes of the next K step: the second block of this group, or the first of the next group
#pragma unroll
          for (int t = 0; t < TA; t++) {
            const bool keep = there[t] && (s == 0 ? live : live1);
            P[nx][t] = tpg_planes(s == 0 ? RA[C][t][1] : RA[N][t][0], keep ? TPG_NIB_V : 0u, keep ? TPG_NIB_D : 0u,
                                  keep ? TPG_NIB_H : 0u);
          }
          P[nx][TA] = tpg_planes(s == 0 ? RB[C][1] : RB[N][0], TPG_NIB_V, TPG_NIB_D, TPG_NIB_H);
#pragma unroll
          for (int t = 0; t < TA; t++) cV[t] = MFMA_F4(P[cur][t].v, P[cur][TA].v, cV[t], sc1, sc1);
#pragma unroll
          for (int t = 0; t < TA; t++) cD[t] = MFMA_F4(P[cur][t].d, P[cur][TA].d, cD[t], sch, sch);
#pragma unroll
          for (int t = 0; t < TA; t++) cH[t] = MFMA_F4(P[cur][t].h, P[cur][TA].h, cH[t], sc2, sc2);
#pragma unroll
          for (int t = 0; t < TA; t++) cHV[t] = MFMA_F4(P[cur][t].h, P[cur][TA].v, cHV[t], sc2, sc1);
#pragma unroll
          for (int t = 0; t < TA; t++) cVH[t] = MFMA_F4(P[cur][t].v, P[cur][TA].h, cVH[t], sc1, sc2);
          // 12 (TA + 1) plane masks and TA + 1 loads spread over the 5 TA MFMAs of the step
#pragma unroll
          for (int q = 0; q < 5 * TA; q++) {
            __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SGB_VALU, 4, 0);
            if (q % 4 == 1 && q / 4 < TA + 1) __builtin_amdgcn_sched_group_barrier(SGB_VMEM_READ, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      using I3 = std::integral_constant<int, 3>;
      using TF = std::false_type;
      using TT = std::true_type;
      if constexpr (NG == 3) {
        int kg = k0;
        for (; kg + 4 <= k1; kg += 3) {  // groups kg .. kg + 3 inside the range (a step looks one group ahead)
          group(I0{}, I1{}, I2{}, TF{}, kg);
          group(I1{}, I2{}, I0{}, TF{}, kg + 1);
          group(I2{}, I0{}, I1{}, TF{}, kg + 2);
        }
        for (; kg < k1; kg += 3) {
          group(I0{}, I1{}, I2{}, TT{}, kg);
          group(I1{}, I2{}, I0{}, TT{}, kg + 1);
          group(I2{}, I0{}, I1{}, TT{}, kg + 2);
        }
      } else {
        for (int kg = k0; kg < k1; kg += 4) {
          group(I0{}, I1{}, I3{}, TT{}, kg);
          group(I1{}, I2{}, I0{}, TT{}, kg + 1);
          group(I2{}, I3{}, I1{}, TT{}, kg + 2);
          group(I3{}, I0{}, I2{}, TT{}, kg + 3);
        }
      }
    }
    // the accumulators hold integers (|sum| <= loci of the K range <= 2^24): exact in int32
    int32_t* slab = acc_out + tp0 * TPG_PW_TILE_INTS + lane;
#pragma unroll
    for (int t = 0; t < TA; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int o = (t * 16 + r) * 64;
        atomicAdd(slab + 0 * TPG_PW_PLANE_INTS + o, (int)cV[t][r]);
        atomicAdd(slab + 1 * TPG_PW_PLANE_INTS + o, (int)cD[t][r]);
        atomicAdd(slab + 2 * TPG_PW_PLANE_INTS + o, (int)cH[t][r]);
        atomicAdd(slab + 3 * TPG_PW_PLANE_INTS + o, (int)cHV[t][r]);
        atomicAdd(slab + 4 * TPG_PW_PLANE_INTS + o, (int)cVH[t][r]);
      }
  }
}

// ---------------------------------------------------------------------------
// Product subsets.  A stand-alone snp_allele_sharing / pairwise_grm needs V and D only, snp_ibs V, D, H, snp_king V, D,
// A, A' (the reference runs 2 / 6 / 4 dense products for them: src/snp_as.cpp:64-65, src/snp_ibs.cpp:67-72,
// src/snp_king.cpp:70-72).  Fewer sums per pair leave accumulator registers for MORE PAIRS per wave, i.e. more MFMAs per
// operand byte crossing the L2 -> CU path, which is what the five-product 96 x 32 tile is short of when products are
// dropped from it (6 MFMAs per 4 fragments for {V, D}).  One kernel template, instantiated per product set with a wave
// tile of (32 RA) x (32 RB) pairs:
//   {V, D}        128 x 64   16 accumulator tiles, 16 MFMAs per 6 fragments and 64 loci    8.7 - 8.9 ms at 5 000 x 1 000 000
//   {V, D, H}      64 x 64   12 accumulator tiles, 12 MFMAs per 4 fragments               13.6 ms
//   {V, D, A, A'}  64 x 64   16 accumulator tiles, 16 MFMAs per 4 fragments               16.1 - 16.2 ms
// (all five, the kernel above: 96 x 32, 15 tiles, 15 MFMAs per 4 fragments: 19.3 - 19.5 ms)
// Same T4 operands, same slab layout as the five-product kernel (a tile (rt, ct) of the wave is sub-tile rt % 3 of slab
// (rt / 3, ct)), so accumulators, reduce-scatter and epilogues do not care which kernel filled them; planes of products
// that were not asked for stay zero and tpg_pairwise.have says which sums are complete.  Only tiles on or above the
// diagonal (ct >= rt) with data are written.  Operand blocks (64 loci) go through NS rotating register slots: while the
// MFMAs of block b issue, the planes of block b + 1 are masked out of slot (b + 1) % NS and block b + NS - 1 is fetched
// into the slot block b - 1 left (prefetch distance NS - 1 steps).
template <int MASK>
struct PwSet {
  // pS: D and H into ONE accumulator (TPG_PW_DH: IBS = V + (D + H) needs nothing else) -- K-concatenation: the d x d and the
  // h x h MFMA of a tile pair add to the same sums, each with its own block scales, so the pair costs two accumulator tiles
  // instead of three and the wave tile can be the {V, D} kernel's 128 x 64
  static constexpr bool pV = (MASK & TPG_PW_V) != 0, pD = (MASK & TPG_PW_D) != 0, pH = (MASK & TPG_PW_H) != 0,
                        pA = (MASK & TPG_PW_A) != 0, pS = (MASK & TPG_PW_DH) != 0;
  static constexpr int NP = (pV ? 1 : 0) + (pD ? 1 : 0) + (pH ? 1 : 0) + (pA ? 2 : 0) + (pS ? 2 : 0);  // MFMAs per tile pair
  static constexpr bool wv = pV || pA, wd = pD || pS, wh = pH || pA || pS;  // operand planes wanted
  static constexpr int NPL = (wv ? 1 : 0) + (wd ? 1 : 0) + (wh ? 1 : 0);
};

template <int MASK>
__device__ __forceinline__ Frag3 tpg_planes_of(v4u w, uint32_t mv, uint32_t md, uint32_t mh) {
  Frag3 f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if constexpr (PwSet<MASK>::wv) f.v[k] = (int)(w[k] & mv);
    if constexpr (PwSet<MASK>::wd) f.d[k] = (int)(w[k] & md);
    if constexpr (PwSet<MASK>::wh) f.h[k] = (int)(w[k] & mh);
  }
  return f;
}

template <typename F, int... Ss>
__device__ __forceinline__ void tpg_unrolled_steps(F& step, int kb, std::integer_sequence<int, Ss...>) {
  (step(std::integral_constant<int, Ss>{}, kb + Ss), ...);
}

// DBG (timing experiments only, wrong sums; TPG_PW_VARIANT=21 / 22): 1 = the loads of the loop removed (the slots keep what the
// prologue fetched), 2 = the plane masks of the loop removed as well (the MFMAs run on the first block's planes)
template <int RA, int RB, int MASK, int NS, int DBG = 0>
__global__ __launch_bounds__(256, 1) void tpg_pairwise_set_kernel(const uint4* __restrict__ T4, int64_t KG,
                                                                     int64_t kb_begin, int64_t kb_end, int nst, int nct,
                                                                     const int2* __restrict__ order, int64_t nun, int S,
                                                                     const int64_t* __restrict__ rowpad,
                                                                     int32_t* __restrict__ acc_out) {
  using PS = PwSet<MASK>;
  constexpr int NT = RA + RB;                    // operand fragments per 64 loci
  constexpr int NM = RA * RB * PS::NP;           // MFMAs per 64 loci
  constexpr int NV = NT * 4 * PS::NPL;           // plane masks (v_and_b32) per 64 loci
  constexpr int U = (NS % 2 == 0) ? NS : 2 * NS; // steps per unrolled loop body: slots and plane sets both come round
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t kbs = kb_end - kb_begin;
  const int xcd = blockIdx.x & 7, cidx = blockIdx.x >> 3, cpx = gridDim.x >> 3;
  const int sc1 = TPG_SC_ONE, sc2 = TPG_SC_TWO, sch = TPG_SC_HALF;
  for (int64_t round = 0;; round++) {
    const int64_t un = ((round * 8 + xcd) * cpx + cidx) * 4 + wv;
    if (un >= nun * S) break;
    const int ks = (int)(un / nun);
    const int2 ij = order[un % nun];
    const int I = __builtin_amdgcn_readfirstlane(ij.x), J = __builtin_amdgcn_readfirstlane(ij.y);
    // 32-bit block indices (a view has fewer than 2^26 blocks): the range tests of the loop are scalar compares, not VALU
    // compares of two uniform 64-bit values, and an index step is one s_add_i32
    const int kb0 = __builtin_amdgcn_readfirstlane((int)(kb_begin + (kbs * ks) / S));
    const int kb1 = __builtin_amdgcn_readfirstlane((int)(kb_begin + (kbs * (ks + 1)) / S));
    // row tiles of the wave: RA A tiles (rows of the output), RB B tiles (columns).  A tile past the last one with data
    // reads tile 0 instead; what it yields is never stored
    // (fixed bases at block kb0 in SGPRs; a block is one 32-bit lane offset: a wave-unit never spans 2^22 blocks)
    const char* pt[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int tile = t < RA ? RA * I + t : RB * J + (t - RA);
      pt[t] = (const char*)(T4 + ((int64_t)(tile < nct ? tile : 0) * KG * 2 + kb0) * 64);
    }
    v16f cV[RA][RB], cD[RA][RB], cH[RA][RB], cHV[RA][RB], cVH[RA][RB];
#pragma unroll
    for (int a = 0; a < RA; a++)
#pragma unroll
      for (int b = 0; b < RB; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) { cV[a][b][r] = 0.f; cD[a][b][r] = 0.f; cH[a][b][r] = 0.f; cHV[a][b][r] = 0.f; cVH[a][b][r] = 0.f; }

    if (kb0 < kb1) {
      const int kl = kb1 - 1;
      v4u R[NS][NT];
      auto LDB = [&](v4u* slot, int b) {  // block b of the NT tiles
        uint32_t off = (uint32_t)lane * 16u + (uint32_t)(b - kb0) * 1024u;
        asm("" : "+v"(off));
#pragma unroll
        for (int t = 0; t < NT; t++) slot[t] = *(const v4u*)(pt[t] + off);
      };
#pragma unroll
      for (int s = 0; s < NS - 1; s++) {
        LDB(R[s], kb0 + s < kb1 ? kb0 + s : kl);
        // in THIS order: hipcc's wait-count pass merges the loop's entry state with its back edge, so a prologue that
        // loads the slot the first step needs last costs a vmcnt(0) at the top of every loop body
        __builtin_amdgcn_sched_barrier(0);
      }
      Frag3 P[2][NT];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        P[0][t] = tpg_planes_of<MASK>(R[0][t], TPG_NIB_V, TPG_NIB_D, TPG_NIB_H);
        if constexpr (DBG == 2) P[1][t] = tpg_planes_of<MASK>(R[1][t], TPG_NIB_V, TPG_NIB_D, TPG_NIB_H);
      }
      auto step = [&](auto Sc, int kb) {
        constexpr int s = decltype(Sc)::value, cur = s & 1, nx = cur ^ 1, sl = (s + 1) % NS, ld = (s + NS - 1) % NS;
        // the slot whose planes were taken in the previous step is free: block kb + NS - 1
        if constexpr (DBG == 0) LDB(R[ld], kb + NS - 1 < kb1 ? kb + NS - 1 : kl);
        // planes of the next block (zero A planes past the K range: the tail of the last unrolled body adds nothing)
        const bool live1 = kb + 1 < kb1;
#pragma unroll
        for (int t = 0; t < NT; t++) {
          const bool keep = t >= RA || live1;
          if constexpr (DBG < 2) P[nx][t] = tpg_planes_of<MASK>(R[sl][t], keep ? TPG_NIB_V : 0u, keep ? TPG_NIB_D : 0u, keep ? TPG_NIB_H : 0u);
        }
#pragma unroll
        for (int a = 0; a < RA; a++)
#pragma unroll
          for (int b = 0; b < RB; b++) {
            if constexpr (PS::pV) cV[a][b] = MFMA_F4(P[cur][a].v, P[cur][RA + b].v, cV[a][b], sc1, sc1);
            if constexpr (PS::pD || PS::pS) cD[a][b] = MFMA_F4(P[cur][a].d, P[cur][RA + b].d, cD[a][b], sch, sch);
            if constexpr (PS::pH) cH[a][b] = MFMA_F4(P[cur][a].h, P[cur][RA + b].h, cH[a][b], sc2, sc2);
            if constexpr (PS::pA) cHV[a][b] = MFMA_F4(P[cur][a].h, P[cur][RA + b].v, cHV[a][b], sc2, sc1);
            if constexpr (PS::pA) cVH[a][b] = MFMA_F4(P[cur][a].v, P[cur][RA + b].h, cVH[a][b], sc1, sc2);
          }
        // (the second MFMA into the D + H sums: a pass of its own, 2 RA RB MFMAs behind the first into the same registers)
        if constexpr (PS::pS) {
#pragma unroll
          for (int a = 0; a < RA; a++)
#pragma unroll
            for (int b = 0; b < RB; b++) cD[a][b] = MFMA_F4(P[cur][a].h, P[cur][RA + b].h, cD[a][b], sc2, sc2);
        }
        // NV plane masks (+ address arithmetic) and NT loads spread over the NM MFMAs of the step
#pragma unroll
        for (int q = 0; q < NM; q++) {
          __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 1, 0);
          __builtin_amdgcn_sched_group_barrier(SGB_VALU, (NV + NM - 1) / NM + 1, 0);
          if ((NT * (q + 1)) / NM != (NT * q) / NM) __builtin_amdgcn_sched_group_barrier(SGB_VMEM_READ, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      for (int kb = kb0; kb < kb1; kb += U) tpg_unrolled_steps(step, kb, std::make_integer_sequence<int, U>{});
    }
    // integer sums (<= 2^24 loci per wave-unit): exact in int32.  Tiles on or above the diagonal only.
#pragma unroll
    for (int a = 0; a < RA; a++)
#pragma unroll
      for (int b = 0; b < RB; b++) {
        const int rt = RA * I + a, ct = RB * J + b;
        if (rt < nct && ct < nct && ct >= rt) {
          const int I3 = rt / TA, a3 = rt - TA * I3;
          int32_t* slab = acc_out + (tpg_pw_unit_index(nst, I3, ct) + rowpad[I3]) * TPG_PW_TILE_INTS + a3 * 16 * 64 + lane;
#pragma unroll
          for (int r = 0; r < 16; r++) {
            if constexpr (PS::pV) atomicAdd(slab + 0 * TPG_PW_PLANE_INTS + r * 64, (int)cV[a][b][r]);
            if constexpr (PS::pD || PS::pS) atomicAdd(slab + 1 * TPG_PW_PLANE_INTS + r * 64, (int)cD[a][b][r]);  // (pS: D + H)
            if constexpr (PS::pH) atomicAdd(slab + 2 * TPG_PW_PLANE_INTS + r * 64, (int)cH[a][b][r]);
            if constexpr (PS::pA) atomicAdd(slab + 3 * TPG_PW_PLANE_INTS + r * 64, (int)cHV[a][b][r]);
            if constexpr (PS::pA) atomicAdd(slab + 4 * TPG_PW_PLANE_INTS + r * 64, (int)cVH[a][b][r]);
          }
        }
      }
  }
}


// ---------------------------------------------------------------------------
// The product-set kernels with the operands SHARED by the four waves of a workgroup (round 5).  tpg_pairwise_set_kernel is
// bound on the L2 -> CU path (every wave fetches its own RA + RB fragments per 64 loci: 146 - 195 GB per launch, 16 TB/s: what
// the 8 XCDs deliver), not on the matrix cores.  Here a workgroup of 2 x 2 waves owns a (64 RA) x (64 RB) block of pairs; the
// 2 RA + 2 RB fragments of a 64-locus block cross the L2 -> CU path ONCE (half the bytes per MFMA) by LDS-DMA
// (global_load_lds_dwordx4: one 1-KiB fragment per wave instruction, no VGPR, no VALU, no ds_write), every wave issuing a
// quarter of them, into a ring of NST stages in LDS; a wave reads its RA + RB fragments from the ring (ds_read_b128,
// lane-linear: conflict-free) one block ahead of its MFMAs.  One s_barrier per block: behind it block b + 1 is complete in LDS
// and the stage of block b is free for block b + NST.  Every stage is its OWN __shared__ array and the loop is unrolled over
// the ring, so that hipcc's wait-count pass can tell the LDS-DMA into one stage from the reads of another (alias scopes per
// LDS variable): a single array would make every ds_read wait for every DMA in flight, vmcnt(0), i.e. no prefetch at all.
// Same T4 operands, same slabs, same atomics as the other kernels.
template <typename F, int... Is>
__device__ __forceinline__ void tpg_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void tpg_static_for(F&& f) {
  tpg_static_for_impl(f, std::make_integer_sequence<int, N>{});
}
// s_waitcnt immediate of gfx9: vmcnt in bits [3:0] and [15:14], expcnt [6:4] and lgkmcnt [11:8] left at "do not wait"
constexpr int tpg_waitcnt_vm(int n) { return (n & 15) | (7 << 4) | (15 << 8) | (((n >> 4) & 3) << 14); }
#define SGB_DS_READ 0x100

// DBGW (timing experiments only, wrong sums; TPG_PW_VARIANT=24 / 25 / 26): 1 = no barrier inside the steps, 2 = neither the barrier
// nor the LDS-DMA of the steps (the ring keeps what the prologue fetched), 3 = the barrier kept, the LDS-DMA of the steps removed
template <int RA, int RB, int MASK, int NST, int DBGW = 0>
__global__ __launch_bounds__(256, 1) void tpg_pairwise_wg_kernel(const uint4* __restrict__ T4, int64_t KG, int64_t kb_begin,
                                                                    int64_t kb_end, int nst, int nct, const int2* __restrict__ order,
                                                                    int64_t nun, int S, const int64_t* __restrict__ rowpad,
                                                                    int32_t* __restrict__ acc_out) {
  using PS = PwSet<MASK>;
  constexpr int NFW = 2 * RA + 2 * RB;           // fragments of the workgroup per 64 loci
  constexpr int NLD = NFW / 4;                   // ... of which a wave fetches this many
  static_assert(NFW % 4 == 0, "the fragments of a block are dealt evenly to the four waves");
  constexpr int NT = RA + RB;                    // fragments a wave consumes per 64 loci
  constexpr int NM = RA * RB * PS::NP;           // MFMAs per wave and 64 loci
  constexpr int NV = NT * 4 * PS::NPL;           // plane masks per wave and 64 loci
  constexpr int STB = NFW * 1024;                // bytes of a stage
  typedef char __attribute__((address_space(3)))* lds_t;
  typedef const char __attribute__((address_space(1)))* glb_t;
  // one LDS variable per stage (see above); the unrolled step picks its stage at compile time
  __shared__ __attribute__((aligned(16))) char st0[STB], st1[STB], st2[STB], st3[NST > 3 ? STB : 16], st4[NST > 4 ? STB : 16],
      st5[NST > 5 ? STB : 16];
  static_assert(NST >= 4 && NST <= 6, "4 to 6 stages");
  auto stage = [&](auto Sc) -> lds_t {
    constexpr int s = decltype(Sc)::value;
    if constexpr (s == 0) return (lds_t)st0;
    else if constexpr (s == 1) return (lds_t)st1;
    else if constexpr (s == 2) return (lds_t)st2;
    else if constexpr (s == 3) return (lds_t)st3;
    else if constexpr (s == 4) return (lds_t)st4;
    else return (lds_t)st5;
  };
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wa = wv >> 1, wb = wv & 1;
  const int64_t kbs = kb_end - kb_begin;
  const int xcd = blockIdx.x & 7, cidx = blockIdx.x >> 3, cpx = gridDim.x >> 3;
  const int sc1 = TPG_SC_ONE, sc2 = TPG_SC_TWO, sch = TPG_SC_HALF;
  for (int64_t round = 0;; round++) {
    const int64_t un = (round * 8 + xcd) * cpx + cidx;  // one unit per WORKGROUP
    if (un >= nun * S) break;
    const int ks = (int)(un / nun);
    const int2 ij = order[un % nun];
    const int I = __builtin_amdgcn_readfirstlane(ij.x), J = __builtin_amdgcn_readfirstlane(ij.y);
    const int kb0 = __builtin_amdgcn_readfirstlane((int)(kb_begin + (kbs * ks) / S));
    const int kb1 = __builtin_amdgcn_readfirstlane((int)(kb_begin + (kbs * (ks + 1)) / S));
    // fragment f of the workgroup: f < 2 RA: A row tile 2 RA I + f; else B row tile 2 RB J + (f - 2 RA).  A tile past the last
    // one with data reads tile 0 instead; what it yields is never stored.  Wave wv fetches fragments NLD wv .. NLD wv + NLD - 1.
    glb_t src[NLD];
#pragma unroll
    for (int q = 0; q < NLD; q++) {
      const int f = NLD * wv + q;
      const int tile = f < 2 * RA ? 2 * RA * I + f : 2 * RB * J + (f - 2 * RA);
      src[q] = (glb_t)(T4 + ((int64_t)(tile < nct ? tile : 0) * KG * 2 + kb0) * 64);
    }
    v16f cV[RA][RB], cD[RA][RB], cH[RA][RB], cHV[RA][RB], cVH[RA][RB];
#pragma unroll
    for (int a = 0; a < RA; a++)
#pragma unroll
      for (int b = 0; b < RB; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) { cV[a][b][r] = 0.f; cD[a][b][r] = 0.f; cH[a][b][r] = 0.f; cHV[a][b][r] = 0.f; cVH[a][b][r] = 0.f; }

    if (kb0 < kb1) {
      const int kl = kb1 - 1;
      // the previous unit's last reads of the ring are behind every wave before anything is written into it again
      __builtin_amdgcn_s_barrier();
      auto DMA = [&](auto Sc, int b) {  // this wave's share of block b -> stage Sc
        uint32_t off = (uint32_t)lane * 16u + (uint32_t)(b - kb0) * 1024u;
        asm("" : "+v"(off));
        lds_t dst = stage(Sc);
#pragma unroll
        for (int q = 0; q < NLD; q++)
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src[q] + off),
                                           (void __attribute__((address_space(3)))*)(dst + (NLD * wv + q) * 1024), 16, 0, 0);
      };
      auto RD = [&](auto Sc, v4u* w) {  // this wave's RA + RB fragments out of stage Sc
        lds_t base = stage(Sc);
#pragma unroll
        for (int t = 0; t < NT; t++) {
          const int f = t < RA ? RA * wa + t : 2 * RA + RB * wb + (t - RA);
          w[t] = *(const v4u __attribute__((address_space(3)))*)(base + f * 1024 + lane * 16);
        }
      };
      // prologue: blocks kb0 .. kb0 + NST - 1 on their way (clamped to the last block of the range: the counts below rely on
      // every step issuing its loads)
      tpg_static_for<NST>([&](auto Sc) {
        constexpr int s = decltype(Sc)::value;
        DMA(Sc, kb0 + s < kb1 ? kb0 + s : kl);
      });
      // The ring is read TWO blocks ahead of the MFMAs, the planes are masked ONE block ahead: right behind the barrier all four
      // waves read at once (24 KiB per block: ~100 cycles of the LDS array + its latency), and with the reads one block ahead
      // the first plane mask of a step -- and every MFMA behind it: a wave issues in order -- waited for them (9.0 - 9.3 ms
      // against 8.4 for the kernel without LDS; measured).  W[q] holds block kb0 + q (mod 2).
      // vmcnt counts this wave's DMAs in order: blocks kb0, kb0 + 1 have landed when at most NLD (NST - 2) younger ones fly
      __builtin_amdgcn_s_waitcnt(tpg_waitcnt_vm(NLD * (NST - 2)));
      __builtin_amdgcn_s_barrier();
      v4u W[2][NT];
      RD(std::integral_constant<int, 0>{}, W[0]);
      RD(std::integral_constant<int, 1>{}, W[1]);
      Frag3 P[2][NT];
#pragma unroll
      for (int t = 0; t < NT; t++) P[0][t] = tpg_planes_of<MASK>(W[0][t], TPG_NIB_V, TPG_NIB_D, TPG_NIB_H);
      auto step = [&](auto Sc, auto Cc, int kb) {  // block kb: stage s, planes in P[cur]; block kb + 1: words in W[cur ^ 1]
        constexpr int s = decltype(Sc)::value, cur = decltype(Cc)::value, nx = cur ^ 1, s2 = (s + 2) % NST;
        // my pieces of block kb + 2 have landed (younger DMAs: blocks kb + 3 .. kb + NST - 1); behind the barrier everybody's
        // have, and everybody has read block kb (two steps ago): its stage takes block kb + NST
        if constexpr (DBGW == 0 || DBGW == 1) __builtin_amdgcn_s_waitcnt(tpg_waitcnt_vm(NLD * (NST - 3)));
        if constexpr (DBGW == 0 || DBGW == 3) __builtin_amdgcn_s_barrier();
        if constexpr (DBGW == 0 || DBGW == 1) DMA(Sc, kb + NST < kb1 ? kb + NST : kl);
        const bool live1 = kb + 1 < kb1;  // past the K range: zero A planes, the tail of the last unrolled body adds nothing
#pragma unroll
        for (int t = 0; t < NT; t++) {
          const bool keep = t >= RA || live1;
          P[nx][t] = tpg_planes_of<MASK>(W[nx][t], keep ? TPG_NIB_V : 0u, keep ? TPG_NIB_D : 0u, keep ? TPG_NIB_H : 0u);
        }
        // (W[cur] held block kb: masked a step ago)
        RD(std::integral_constant<int, s2>{}, W[cur]);
#pragma unroll
        for (int a = 0; a < RA; a++)
#pragma unroll
          for (int b = 0; b < RB; b++) {
            if constexpr (PS::pV) cV[a][b] = MFMA_F4(P[cur][a].v, P[cur][RA + b].v, cV[a][b], sc1, sc1);
            if constexpr (PS::pD) cD[a][b] = MFMA_F4(P[cur][a].d, P[cur][RA + b].d, cD[a][b], sch, sch);
            if constexpr (PS::pH) cH[a][b] = MFMA_F4(P[cur][a].h, P[cur][RA + b].h, cH[a][b], sc2, sc2);
            if constexpr (PS::pA) cHV[a][b] = MFMA_F4(P[cur][a].h, P[cur][RA + b].v, cHV[a][b], sc2, sc1);
            if constexpr (PS::pA) cVH[a][b] = MFMA_F4(P[cur][a].v, P[cur][RA + b].h, cVH[a][b], sc1, sc2);
          }
        // the DMAs and the ring reads spread over the first MFMAs, the plane masks between all of them
        __builtin_amdgcn_sched_group_barrier(SGB_VMEM_READ, NLD, 0);
#pragma unroll
        for (int q = 0; q < NM; q++) {
          __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 1, 0);
          __builtin_amdgcn_sched_group_barrier(SGB_VALU, (NV + NM - 1) / NM + 1, 0);
          // (the first masks wait for the words read a step ago, lgkmcnt(0): no read of THIS step may stand in front of them)
          if (q >= 1 && q <= NT) __builtin_amdgcn_sched_group_barrier(SGB_DS_READ, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      constexpr int U = (NST % 2 == 0) ? NST : 2 * NST;  // stages and plane sets both come round
      for (int kb = kb0; kb < kb1; kb += U)
        tpg_static_for<U>([&](auto Uc) {
          constexpr int u = decltype(Uc)::value;
          // (a body of 2 NST steps walks the ring twice: the stage of step u is u mod NST, its plane set u & 1)
          step(std::integral_constant<int, u % NST>{}, std::integral_constant<int, u & 1>{}, kb + u);
        });
      // drain: nothing of this unit is in flight when the next one (or the end of the kernel) comes
      __builtin_amdgcn_s_waitcnt(tpg_waitcnt_vm(0));
    }
#pragma unroll
    for (int a = 0; a < RA; a++)
#pragma unroll
      for (int b = 0; b < RB; b++) {
        const int rt = 2 * RA * I + RA * wa + a, ct = 2 * RB * J + RB * wb + b;
        if (rt < nct && ct < nct && ct >= rt) {
          const int I3 = rt / TA, a3 = rt - TA * I3;
          int32_t* slab = acc_out + (tpg_pw_unit_index(nst, I3, ct) + rowpad[I3]) * TPG_PW_TILE_INTS + a3 * 16 * 64 + lane;
#pragma unroll
          for (int r = 0; r < 16; r++) {
            if constexpr (PS::pV) atomicAdd(slab + 0 * TPG_PW_PLANE_INTS + r * 64, (int)cV[a][b][r]);
            if constexpr (PS::pD) atomicAdd(slab + 1 * TPG_PW_PLANE_INTS + r * 64, (int)cD[a][b][r]);
            if constexpr (PS::pH) atomicAdd(slab + 2 * TPG_PW_PLANE_INTS + r * 64, (int)cH[a][b][r]);
            if constexpr (PS::pA) atomicAdd(slab + 3 * TPG_PW_PLANE_INTS + r * 64, (int)cHV[a][b][r]);
            if constexpr (PS::pA) atomicAdd(slab + 4 * TPG_PW_PLANE_INTS + r * 64, (int)cVH[a][b][r]);
          }
        }
      }
  }
}

// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
#include "host/host_bands.h"  // pw_bands

static size_t pw_buffer_bytes(int64_t n, int nranks) {
  const int64_t nst = ceil_div(n, 32 * TA);
  std::vector<int32_t> band;
  int64_t chunk;
  pw_bands(nst, nranks, band, chunk);
  return (size_t)(chunk * nranks) * TPG_PW_TILE_INTS * sizeof(int32_t);
}

extern "C" size_t tpg_pairwise_buffer_bytes(int64_t n) { return pw_buffer_bytes(n, 1); }
extern "C" size_t tpg_pairwise_buffer_bytes_sharded(int64_t n, int nranks) { return pw_buffer_bytes(n, nranks < 1 ? 1 : nranks); }

static int pairwise_create(tpg_ctx* ctx, int64_t n, int nranks, int rank, void* ext_buffer, tpg_pairwise** out) {
  TPG_REQUIRE(ctx && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(n > 0 && n < (1 << 22), TPG_EINVAL, "bad n = %lld", (long long)n);
  TPG_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, TPG_EINVAL, "bad rank %d of %d", rank, nranks);
  tpg_pairwise* pw = new tpg_pairwise{ctx, n, ceil_div(n, 32 * TA), 0, nullptr, false, nullptr, 0, 0, 0};
  pw->ntp = TA * pw->nst * (pw->nst + 1) / 2;
  pw->nranks = nranks;
  pw->rank = rank;
  pw_bands(pw->nst, nranks, pw->band, pw->chunk_units);
  {
    // units (I, jt), jt >= 2 I, in patch order: blocks of 16 column tiles, inside a block row after row.  Column
    // tiles that hold only padding (32 jt >= n: all-missing codes, zero products) are left out -- their slabs stay
    // zero and no epilogue reads them.
    const int nst = (int)pw->nst, nct = (int)ceil_div(n, 32);
    std::vector<int2> order;
    order.reserve((size_t)pw->ntp);
    const int PWD = 16;  // 8 ... 32 measured within 2 % of each other
    for (int pc = 0; pc * PWD < nct; pc++) {
      const int c1 = std::min(nct, pc * PWD + PWD);
      for (int I = 0; I < nst && TA * I < c1; I++)
        for (int jt = std::max(pc * PWD, TA * I); jt < c1; jt++) order.push_back(make_int2(I, jt));
    }
    pw->nun = (int64_t)order.size();
    std::vector<int64_t> rowpad((size_t)nst);
    auto off = [&](int64_t I) { return TA * (I * nst - (I * (I - 1)) / 2); };
    for (int r = 0; r < nranks; r++)
      for (int I = pw->band[(size_t)r]; I < pw->band[(size_t)r + 1]; I++)
        rowpad[(size_t)I] = (int64_t)r * pw->chunk_units - off(pw->band[(size_t)r]);
    hipError_t e = tpg_pmalloc(&pw->order, sizeof(int2) * order.size());
    if (e == hipSuccess) e = hipMemcpyAsync(pw->order, order.data(), sizeof(int2) * order.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = tpg_pmalloc((void**)&pw->rowpad, sizeof(int64_t) * rowpad.size());
    if (e == hipSuccess) e = hipMemcpyAsync(pw->rowpad, rowpad.data(), sizeof(int64_t) * rowpad.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the host vectors go out of scope
    if (e != hipSuccess) { tpg_pairwise_free(pw); tpg_set_error("pairwise tables: %s", hipGetErrorString(e)); return TPG_EHIP; }
  }
  if (ext_buffer) {
    if (!tpg_is_device_ptr(ext_buffer)) { tpg_pairwise_free(pw); tpg_set_error("ext_buffer is not device memory"); return TPG_EINVAL; }
    pw->acc = (int32_t*)ext_buffer;
  } else {
    hipError_t e = tpg_pmalloc((void**)&pw->acc, pw_buffer_bytes(n, nranks));
    if (e != hipSuccess) { tpg_pairwise_free(pw); tpg_set_error("hipMalloc pairwise buffer: %s", hipGetErrorString(e)); return TPG_EHIP; }
    pw->owns = true;
  }
  int rc = tpg_pairwise_zero(ctx, pw);
  if (rc != TPG_OK) { tpg_pairwise_free(pw); return rc; }
  *out = pw;
  return TPG_OK;
}

extern "C" int tpg_pairwise_create(tpg_ctx* ctx, int64_t n, void* ext_buffer, tpg_pairwise** out) {
  TpgEnter _enter(ctx);
  return pairwise_create(ctx, n, 1, 0, ext_buffer, out);
}

// Accumulators of one rank of a communicator: same kernel, the buffer laid out for one reduce-scatter
// (tpg_pairwise_reduce); see tpg_pairwise in common.h.
extern "C" int tpg_pairwise_create_sharded(tpg_ctx* ctx, const tpg_comm* comm, int64_t n, tpg_pairwise** out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(comm, TPG_EINVAL, "null communicator");
  TPG_REQUIRE(comm->ctx == ctx, TPG_EINVAL, "the communicator belongs to another context");
  return pairwise_create(ctx, n, comm->nranks, comm->rank, nullptr, out);
}

// Sum the partial cross-products of all ranks: one reduce-scatter of the int32 slabs (exact, order independent).
// Afterwards this rank holds the complete sums of ITS band of super-tile rows only, and the count / epilogue entry
// points write only the part of the N x N outputs that band covers: rows [row0, row1) x columns >= row0 and its
// mirror image (tpg_pairwise_band).
extern "C" int tpg_pairwise_reduce(tpg_ctx* ctx, tpg_comm* comm, tpg_pairwise* pw) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && comm && pw, TPG_EINVAL, "null argument");
  TPG_REQUIRE(comm->ctx == ctx && pw->ctx == ctx, TPG_EINVAL, "communicator / accumulators belong to another context");
  TPG_REQUIRE(pw->nranks == comm->nranks && pw->rank == comm->rank, TPG_EINVAL,
              "accumulators were created for rank %d of %d, the communicator is rank %d of %d", pw->rank, pw->nranks,
              comm->rank, comm->nranks);
  TPG_REQUIRE(!pw->reduced, TPG_EINVAL, "already reduced: zero the accumulators first");
  // the locus count behind the overflow guard is the total over the ranks: known BEFORE the sums are exchanged, so that
  // a panel that is too long is refused with the accumulators intact instead of after they have wrapped
  // With it travels, per product, the number of ranks that did NOT accumulate it since their last zero: the sums of a product
  // are complete only if every rank added it (`have` is rank-local; a rank that ran all five products must not pass pw_need
  // for IBS / KING when another contributed {V, D} alone).
  const int bits[5] = {TPG_PW_V, TPG_PW_D, TPG_PW_H, TPG_PW_A, TPG_PW_DH};
  double word[6] = {(double)pw->loci, 0, 0, 0, 0, 0};
  for (int b = 0; b < 5; b++) word[1 + b] = (pw->have & bits[b]) ? 0.0 : 1.0;
  if (comm->nranks > 1 || comm->nccl) TPG_TRY(tpg_comm_allreduce_f64(ctx, comm, word, 6));
  const double loci = word[0];
  TPG_REQUIRE(loci <= (double)TPG_PW_MAX_LOCI, TPG_EUNSUPPORTED, "%.0f loci over all ranks overflow the int32 pair counts", loci);
  {
    ProfScope ps(ctx, "pairwise_reduce_scatter");
    TPG_TRY(tpg_comm_reduce_scatter_i32(comm, pw->acc, pw->chunk_units * TPG_PW_TILE_INTS));
  }
  pw->loci = (int64_t)loci;
  for (int b = 0; b < 5; b++)
    if (word[1 + b] > 0) pw->have &= ~bits[b];
  pw->reduced = true;
  return TPG_OK;
}

// tpg_pairwise_reduce with the reduce-scatter on the stream of ANOTHER communicator's context (include/tpg.h).  The small
// all-reduce that carries the locus count and the per-product "some rank lacks it" words is host data: it goes first, on the
// side stream, before that stream is made to wait for the accumulate kernels -- otherwise the host would stand still for the
// length of the pairwise kernel and enqueue nothing beside it.
extern "C" int tpg_pairwise_reduce_begin(tpg_ctx* ctx, tpg_comm* side, tpg_pairwise* pw) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && side && pw && side->ctx, TPG_EINVAL, "null argument");
  TPG_REQUIRE(pw->ctx == ctx, TPG_EINVAL, "the accumulators belong to another context");
  TPG_REQUIRE(side->ctx != ctx && side->ctx->device == ctx->device, TPG_EINVAL,
              "the side communicator must sit on another context of the same device");
  TPG_REQUIRE(pw->nranks == side->nranks && pw->rank == side->rank, TPG_EINVAL,
              "accumulators were created for rank %d of %d, the side communicator is rank %d of %d", pw->rank, pw->nranks, side->rank,
              side->nranks);
  TPG_REQUIRE(!pw->reduced && !pw->reducing, TPG_EINVAL, "already reduced, or a reduction is in flight");
  const int bits[5] = {TPG_PW_V, TPG_PW_D, TPG_PW_H, TPG_PW_A, TPG_PW_DH};
  double word[6] = {(double)pw->loci, 0, 0, 0, 0, 0};
  for (int b = 0; b < 5; b++) word[1 + b] = (pw->have & bits[b]) ? 0.0 : 1.0;
  tpg_ctx* sc = side->ctx;
  if (side->nranks > 1 || side->nccl) {
    TpgEnter _side(sc);
    TPG_TRY(tpg_comm_allreduce_f64(sc, side, word, 6));
  }
  TPG_REQUIRE(word[0] <= (double)TPG_PW_MAX_LOCI, TPG_EUNSUPPORTED, "%.0f loci over all ranks overflow the int32 pair counts", word[0]);
  if (!pw->ev_acc) TPG_HIP(hipEventCreateWithFlags(&pw->ev_acc, hipEventDisableTiming));
  if (!pw->ev_red) TPG_HIP(hipEventCreateWithFlags(&pw->ev_red, hipEventDisableTiming));
  TPG_HIP(hipEventRecord(pw->ev_acc, ctx->stream));          // behind the accumulate kernels of pw's context
  TPG_HIP(hipStreamWaitEvent(sc->stream, pw->ev_acc, 0));
  {
    TpgEnter _side(sc);
    TPG_TRY(tpg_comm_reduce_scatter_i32(side, pw->acc, pw->chunk_units * TPG_PW_TILE_INTS));
  }
  TPG_HIP(hipEventRecord(pw->ev_red, sc->stream));
  pw->reducing = true;
  pw->pending_loci = (int64_t)word[0];
  pw->pending_lack = 0;
  for (int b = 0; b < 5; b++)
    if (word[1 + b] > 0) pw->pending_lack |= bits[b];
  return TPG_OK;
}

extern "C" int tpg_pairwise_reduce_end(tpg_ctx* ctx, tpg_comm* side, tpg_pairwise* pw) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && side && pw, TPG_EINVAL, "null argument");
  TPG_REQUIRE(pw->ctx == ctx && pw->reducing, TPG_EINVAL, "no reduction in flight on these accumulators");
  TPG_HIP(hipStreamWaitEvent(ctx->stream, pw->ev_red, 0));  // what follows on pw's context sees the reduced sums
  pw->loci = pw->pending_loci;
  pw->have &= ~pw->pending_lack;
  pw->reduced = true;
  pw->reducing = false;
  return TPG_OK;
}

// the band of rank `rank` when n individuals are shared among nranks ranks (host arithmetic only: no GPU needed)
extern "C" int tpg_pairwise_band_of(int64_t n, int nranks, int rank, int64_t* row0, int64_t* row1) {
  TPG_REQUIRE(row0 && row1 && n > 0 && nranks >= 1 && rank >= 0 && rank < nranks, TPG_EINVAL, "bad argument");
  std::vector<int32_t> band;
  int64_t chunk;
  pw_bands(ceil_div(n, 32 * TA), nranks, band, chunk);
  *row0 = std::min<int64_t>(n, 32 * TA * (int64_t)band[(size_t)rank]);
  *row1 = std::min<int64_t>(n, 32 * TA * (int64_t)band[(size_t)rank + 1]);
  return TPG_OK;
}

// rows [row0, row1) of the individuals whose pairs this rank's outputs cover (all of them before a reduction)
extern "C" int tpg_pairwise_band(const tpg_pairwise* pw, int64_t* row0, int64_t* row1) {
  TPG_REQUIRE(pw && row0 && row1, TPG_EINVAL, "null argument");
  if (!pw->reduced || pw->nranks == 1) { *row0 = 0; *row1 = pw->n; return TPG_OK; }
  *row0 = std::min<int64_t>(pw->n, 32 * TA * (int64_t)pw->band[(size_t)pw->rank]);
  *row1 = std::min<int64_t>(pw->n, 32 * TA * (int64_t)pw->band[(size_t)pw->rank + 1]);
  return TPG_OK;
}

extern "C" void tpg_pairwise_free(tpg_pairwise* pw) {
  if (!pw) return;
  if (pw->ev_acc) (void)hipEventDestroy(pw->ev_acc);
  if (pw->ev_red) (void)hipEventDestroy(pw->ev_red);
  if (pw->owns && pw->acc) tpg_pfree(pw->acc);
  tpg_pfree(pw->order);
  tpg_pfree(pw->rowpad);
  for (auto& o : pw->orders) tpg_pfree(o.second.first);
  delete pw;
}

extern "C" int tpg_pairwise_zero(tpg_ctx* ctx, tpg_pairwise* pw) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && pw, TPG_EINVAL, "null argument");
  ProfScope ps(ctx, "pairwise_zero");
  TPG_HIP(hipMemsetAsync(pw->acc, 0, pw_buffer_bytes(pw->n, pw->nranks), ctx->stream));
  pw->loci = 0;
  TPG_REQUIRE(!pw->reducing, TPG_EINVAL, "a reduction is in flight: tpg_pairwise_reduce_end first");
  pw->reduced = false;
  pw->have = TPG_PW_HAVE_ALL;
  return TPG_OK;
}

// Reference quirk Q1 (src/snp_as.cpp:57-63 with R/snp_allele_sharing.R:55-56): every block of the R
// driver that is one column narrower than the widest adds +1 to EVERY element of the allele-sharing numerator.
// The default (0) is the intended value; a caller who must reproduce a real R run bit for bit passes the number of
// narrower blocks of that run (tpg_as_pad_quirk_blocks) and as_num / allele sharing / GRM come out as R's.
extern "C" int tpg_pairwise_set_as_pad_quirk(tpg_pairwise* pw, int64_t narrow_blocks) {
  TPG_REQUIRE(pw, TPG_EINVAL, "null argument");
  TPG_REQUIRE(narrow_blocks >= 0 && narrow_blocks < (1ll << 31), TPG_EINVAL, "bad block count %lld", (long long)narrow_blocks);
  pw->as_pad_quirk = narrow_blocks;
  return TPG_OK;
}

// blocks narrower than the widest when m loci are cut as the R drivers cut them: CutBySize(m, block.size)
// (R/local_reimplementations.R:13-15) = bigparallelr::split_len(m, nb = ceiling(m / block.size)) (third-party,
// recalled: upper_b = round(b m / nb), R's round-half-even)
extern "C" int64_t tpg_as_pad_quirk_blocks(int64_t m, int64_t block_size) {
  if (m <= 0 || block_size <= 0) return 0;
  const int64_t nb = (m + block_size - 1) / block_size;
  const double step = (double)m / (double)nb;
  int64_t prev = 0, widest = 0;
  std::vector<int64_t> size((size_t)nb);
  for (int64_t b = 0; b < nb; b++) {
    const int64_t up = (int64_t)rint((double)(b + 1) * step);
    size[(size_t)b] = up - prev;
    prev = up;
    widest = std::max(widest, size[(size_t)b]);
  }
  int64_t narrow = 0;
  for (int64_t b = 0; b < nb; b++) narrow += size[(size_t)b] < widest;
  return narrow;
}

// ---- the product-subset kernels: unit tables and launch ----
// units (I, J) of a (32 RA) x (32 RB) wave tile: row group I (row tiles RA I ...), column group J (row tiles RB J ...),
// kept when they hold a tile on or above the diagonal; patch order as for the five-product kernel (blocks of ~16 column
// tiles, inside a block row after row: consecutive units share their A tiles, a run of 128 units is what one XCD's
// waves take in a round)
static int pw_order_for(tpg_ctx* ctx, tpg_pairwise* pw, int RA, int RB, const int2** d_order, int64_t* nun) {
  const int key = 16 * RA + RB;
  auto it = pw->orders.find(key);
  if (it == pw->orders.end()) {
    const int nct = (int)ceil_div(pw->n, 32);
    const int nrg = (int)ceil_div(nct, RA), ncg = (int)ceil_div(nct, RB);
    const int PG = std::max(1, 16 / RB);
    std::vector<int2> order;
    for (int pc = 0; pc * PG < ncg; pc++) {
      const int j1 = std::min(ncg, pc * PG + PG);
      for (int I = 0; I < nrg; I++)
        for (int J = pc * PG; J < j1; J++)
          if (RB * J + RB - 1 >= RA * I) order.push_back(make_int2(I, J));
    }
    void* d = nullptr;
    TPG_HIP(tpg_pmalloc(&d, sizeof(int2) * order.size()));
    hipError_t e = hipMemcpyAsync(d, order.data(), sizeof(int2) * order.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the host vector goes out of scope
    if (e != hipSuccess) { tpg_pfree(d); tpg_set_error("pairwise unit table: %s", hipGetErrorString(e)); return TPG_EHIP; }
    it = pw->orders.emplace(key, std::make_pair(d, (int64_t)order.size())).first;
  }
  *d_order = (const int2*)it->second.first;
  *nun = it->second.second;
  return TPG_OK;
}

// K split of units x S wave-units over the resident waves: a round costs its K range plus the flush of the
// accumulators; t_step in us per K step, t_flush in us
static int pw_ksplit(int64_t nun, int64_t steps, int64_t min_steps_per_unit, int64_t minS, int64_t nwaves, double t_step,
                     double t_flush) {
  int bestS = (int)minS;
  double best = -1;
  const int64_t cap = steps / min_steps_per_unit;
  const int64_t maxS = std::max<int64_t>(minS, cap > 0 ? std::min<int64_t>(cap, 96) : 1);
  for (int64_t S = minS; S <= maxS; S++) {
    const int64_t rounds = ceil_div(nun * S, nwaves);
    const double cost = (double)rounds * ((double)ceil_div(steps, S) * t_step + t_flush);
    if (best < 0 || cost < best * 0.995) { best = cost; bestS = (int)S; }
  }
  return bestS;
}

template <int RA, int RB, int MASK, int NS, int DBG = 0>
static int pw_launch_set(tpg_ctx* ctx, tpg_pairwise* pw, const tpg_view* v, int64_t kg0, int64_t kg1, const char* name) {
  const int2* d_order = nullptr;
  int64_t nun = 0;
  TPG_TRY(pw_order_for(ctx, pw, RA, RB, &d_order, &nun));
  constexpr int NM = RA * RB * PwSet<MASK>::NP;
  const int64_t max_groups = 131072;  // 2^24 loci per wave-unit: FP32 sums of integers stay exact
  int nblk = ctx->num_cu / 8 * 8;
  if (nblk < 8) nblk = 8;
  const int64_t nwaves = 4 * (int64_t)nblk;
  for (int64_t c0 = kg0; c0 < kg1; c0 += 8 * max_groups) {
    const int64_t c1 = std::min(kg1, c0 + 8 * max_groups);
    const int64_t kgs = c1 - c0;
    // per 64-locus block: NM MFMAs at ~18 ns; flush: 16 atomic wave-instructions per accumulator tile (12 us for 15)
    int S = pw_ksplit(nun, 2 * kgs, 16, ceil_div(kgs, max_groups), nwaves, 0.0183 * NM, 0.8 * NM);
    if (const char* e = getenv("TPG_PW_KSPLIT")) S = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(kgs, max_groups), atoi(e)), std::max<int64_t>(1, 2 * kgs));  // (experiments; any S in range gives the same sums)
    if (getenv("TPG_DEBUG")) fprintf(stderr, "[tpg] %s: %d x %d tiles, %lld units, S = %d\n", name, RA, RB, (long long)nun, S);
    TPG_LAUNCH(ctx, name, (tpg_pairwise_set_kernel<RA, RB, MASK, NS, DBG>), dim3((unsigned)nblk), dim3(256), 0,
               (const uint4*)v->T4, v->KG, 2 * c0, 2 * c1, (int)pw->nst, (int)ceil_div(pw->n, 32), d_order, nun, S,
               (const int64_t*)pw->rowpad, pw->acc);
  }
  return TPG_OK;
}

// the workgroup form: units are (64 RA) x (64 RB) blocks of pairs, one per workgroup; TPG_PW_KSPLIT=<S> overrides the K split
template <int RA, int RB, int MASK, int NST, int DBGW = 0>
static int pw_launch_wg(tpg_ctx* ctx, tpg_pairwise* pw, const tpg_view* v, int64_t kg0, int64_t kg1, const char* name) {
  const int2* d_order = nullptr;
  int64_t nun = 0;
  TPG_TRY(pw_order_for(ctx, pw, 2 * RA, 2 * RB, &d_order, &nun));
  constexpr int NM = RA * RB * PwSet<MASK>::NP;
  const int64_t max_groups = 131072;  // 2^24 loci per unit: FP32 sums of integers stay exact
  int nblk = ctx->num_cu / 8 * 8;
  if (nblk < 8) nblk = 8;
  for (int64_t c0 = kg0; c0 < kg1; c0 += 8 * max_groups) {
    const int64_t c1 = std::min(kg1, c0 + 8 * max_groups);
    const int64_t kgs = c1 - c0;
    int S = pw_ksplit(nun, 2 * kgs, 16, ceil_div(kgs, max_groups), nblk, 0.0183 * NM, 0.8 * NM);
    if (const char* e = getenv("TPG_PW_KSPLIT")) S = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(kgs, max_groups), atoi(e)), std::max<int64_t>(1, 2 * kgs));
    if (getenv("TPG_DEBUG")) fprintf(stderr, "[tpg] %s: workgroups of %d x %d tiles, %lld units, S = %d\n", name, 2 * RA, 2 * RB, (long long)nun, S);
    TPG_LAUNCH(ctx, name, (tpg_pairwise_wg_kernel<RA, RB, MASK, NST, DBGW>), dim3((unsigned)nblk), dim3(256), 0,
               (const uint4*)v->T4, v->KG, 2 * c0, 2 * c1, (int)pw->nst, (int)ceil_div(pw->n, 32), d_order, nun, S,
               (const int64_t*)pw->rowpad, pw->acc);
  }
  return TPG_OK;
}

// wave tile and slot count per product set; TPG_PW_VARIANT=<k> picks another instantiation (A/B runs, tools/pw_only.py)
static int pw_variant() {
  const char* e = getenv("TPG_PW_VARIANT");
  return e ? atoi(e) : 0;
}

static int pw_launch_all(tpg_ctx* ctx, tpg_pairwise* pw, const tpg_view* v, int64_t kg0, int64_t kg1) {
  // K split S: units x S wave-units over the resident waves (one workgroup per CU, one wave per SIMD; a multiple of
  // the 8 XCDs), at least 8 K groups (1024 loci) per unit.  Cost model: rounds(S) = ceil(units S / waves) rounds, a
  // round costs its K range (about 0.55 us per 128-locus group: 30 MFMAs at ~34 cycles) plus the flush of the
  // accumulators (240 atomic wave-instructions per wave, all waves at once: ~12 us; S = 5 ... 30 measured within 4 %
  // of each other at 5 000 x 1 000 000, best at 10 - 12).
  // FP32 accumulators: a wave-unit contracts at most 2^24 loci (131072 groups), so that every count it holds is an
  // exactly represented integer; longer ranges go out in several launches of at most 8 x 2^24 loci.
  const int64_t max_groups = 131072;
  int nblk = ctx->num_cu / 8 * 8;
  if (nblk < 8) nblk = 8;
  const int64_t nwaves = 4 * (int64_t)nblk;
  for (int64_t c0 = kg0; c0 < kg1; c0 += 8 * max_groups) {
    const int64_t c1 = std::min(kg1, c0 + 8 * max_groups);
    const int64_t kgs = c1 - c0;
    int bestS = pw_ksplit(pw->nun, kgs, 8, ceil_div(kgs, max_groups), nwaves, 0.55, 12.0);
    // (experiments, tools/pw_ksplit_probe.py: any S in range gives the same sums)
    if (const char* e = getenv("TPG_PW_KSPLIT")) bestS = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(kgs, max_groups), atoi(e)), std::max<int64_t>(1, kgs));
    if (getenv("TPG_DEBUG")) fprintf(stderr, "[tpg] pairwise: %lld units, S = %d\n", (long long)pw->nun, bestS);
    if (pw_variant() == 2)  // A/B: three groups of prefetch
      TPG_LAUNCH(ctx, "pairwise_mfma", tpg_pairwise_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, (const uint4*)v->T4, v->KG,
                 c0, c1, (int)pw->nst, (int)ceil_div(pw->n, 32), (const int2*)pw->order, pw->nun, bestS,
                 (const int64_t*)pw->rowpad, pw->acc);
    else
      TPG_LAUNCH(ctx, "pairwise_mfma", tpg_pairwise_kernel<3>, dim3((unsigned)nblk), dim3(256), 0, (const uint4*)v->T4, v->KG,
                 c0, c1, (int)pw->nst, (int)ceil_div(pw->n, 32), (const int2*)pw->order, pw->nun, bestS,
                 (const int64_t*)pw->rowpad, pw->acc);
  }
  return TPG_OK;
}

extern "C" int tpg_pairwise_accumulate_products(tpg_ctx* ctx, tpg_pairwise* pw, const tpg_view* v, int64_t col_begin,
                                                int64_t col_end, int products) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && pw && v, TPG_EINVAL, "null argument");
  TPG_REQUIRE(pw->n == v->n, TPG_EINVAL, "pairwise n = %lld but view n = %lld", (long long)pw->n, (long long)v->n);
  TPG_REQUIRE(!pw->reduced, TPG_EINVAL, "the accumulators were reduced over the ranks: zero them before accumulating again");
  TPG_REQUIRE(products > 0 && (products & ~(TPG_PW_ALL | TPG_PW_DH)) == 0, TPG_EINVAL, "bad product set 0x%x", products);
  TPG_REQUIRE(!(products & TPG_PW_DH) || (products & ~(TPG_PW_V | TPG_PW_DH)) == 0, TPG_EINVAL,
              "TPG_PW_DH (D + H in one sum) goes with TPG_PW_V only, not with 0x%x", products & ~(TPG_PW_V | TPG_PW_DH));
  if (col_end < 0) col_end = v->m;
  TPG_REQUIRE(col_begin >= 0 && col_begin <= col_end && col_end <= v->m, TPG_EINVAL, "bad locus range [%lld,%lld)",
              (long long)col_begin, (long long)col_end);
  // the kernel works on whole 128-locus groups; a range that is not aligned would need masking
  TPG_REQUIRE(col_begin % 128 == 0 && (col_end % 128 == 0 || col_end == v->m), TPG_EINVAL,
              "locus range must start on a multiple of 128 and end on one (or at m)");
  if (col_begin == col_end) return TPG_OK;
  // int32 accumulators: every product is bounded by the number of loci accumulated (the epilogues form their sums
  // in 64 bits), so the total must stay below 2^31
  TPG_REQUIRE(pw->loci + (col_end - col_begin) <= TPG_PW_MAX_LOCI, TPG_EUNSUPPORTED,
              "%lld loci accumulated + %lld more would overflow the int32 pair counts (limit %lld)", (long long)pw->loci,
              (long long)(col_end - col_begin), (long long)TPG_PW_MAX_LOCI);
  // the kernels that exist: {V, D}, {V, D, H}, {V, D, A}, all five; anything else takes the smallest one that covers it
  int set = products | TPG_PW_V | TPG_PW_D;
  if ((set & TPG_PW_H) && (set & TPG_PW_A)) set = TPG_PW_ALL;
  if (products & TPG_PW_DH) set = TPG_PW_FOR_IBS_ALONE;
  // (pw->loci and pw->have are committed at the end, once every launch has been accepted: a failed accumulate must not leave
  // them claiming work that never ran)
  // the FP4 operand form of the view: written by the pack kernel (tpg_view_create_pair) or made here on first use
  if (!v->T4) {
    TPG_TRY(tpg_view_need_T(ctx, v));
    uint4* t4 = nullptr;
    TPG_HIP(tpg_pmalloc((void**)&t4, 2 * v->bytes_each));
    const int64_t nblocks = 4 * v->Q * v->KG;
    int grid = (int)std::min<int64_t>(ceil_div(nblocks * 64, 256), (int64_t)ctx->num_cu * 32);
    TPG_LAUNCH(ctx, "t4_expand", tpg_t4_expand_kernel, dim3((unsigned)grid), dim3(256), 0, (const uint4*)v->T, t4, nblocks);
    v->T4 = t4;
  }
  const int64_t kg0 = col_begin / 128, kg1 = ceil_div(col_end, 128);
  const int var = pw_variant();
#define PW_SET(RA, RB, MASK, NS, name) TPG_TRY((pw_launch_set<RA, RB, MASK, NS>(ctx, pw, v, kg0, kg1, name)))
  // measured at 5 000 x 1 000 000 (tools/pw_only.py; ms, the same GPU): {V, D} 128 x 64 with 5 / 4 / 3 slots 8.9 / 9.1 / 12.4,
  // 96 x 64 9.8, 64 x 64 11.8; {V, D, H} 64 x 64 with 5 / 4 / 3 / 6 / 7 slots 13.6 / 14.3 / 22.6 / 13.9 / 14.0, 96 x 32 15.7;
  // {V, D, A} 64 x 64 with 5 / 4 / 3 / 6 slots 16.2 / 16.5 / 23.5 / 16.4, 128 x 32 17.9, 96 x 32 17.8; all five through
  // this template (96 x 32, 4 ... 7 slots) 20.2 - 21.8 against 19.5 for the kernel above with its two-block groups
#define PW_WG(RA, RB, MASK, NST, name) TPG_TRY((pw_launch_wg<RA, RB, MASK, NST>(ctx, pw, v, kg0, kg1, name)))
  if (set == TPG_PW_FOR_AS && var >= 10 && var < 20) {  // operands shared through LDS (round 5): 14 / 15 / 16 = 4 / 5 / 6 stages
    if (var == 15) PW_WG(4, 2, TPG_PW_FOR_AS, 5, "pairwise_mfma_as");
    else if (var == 16) PW_WG(4, 2, TPG_PW_FOR_AS, 6, "pairwise_mfma_as");
    else PW_WG(4, 2, TPG_PW_FOR_AS, 4, "pairwise_mfma_as");
  } else if (set == TPG_PW_FOR_IBS && var >= 10 && var < 20) {
    if (var == 16) PW_WG(2, 2, TPG_PW_FOR_IBS, 6, "pairwise_mfma_ibs");
    else PW_WG(2, 2, TPG_PW_FOR_IBS, 4, "pairwise_mfma_ibs");
  } else if (set == TPG_PW_FOR_KING && var >= 10 && var < 20) {
    if (var == 16) PW_WG(2, 2, TPG_PW_FOR_KING, 6, "pairwise_mfma_king");
    else PW_WG(2, 2, TPG_PW_FOR_KING, 4, "pairwise_mfma_king");
  } else if (set == TPG_PW_FOR_AS) {
    // 21 ... 26: timing-only instantiations that give WRONG sums (loads / barriers / LDS-DMA removed: DESIGN.md 3.1 "Round 5").
    // They exist only in a library built with -DTPG_PW_EXPERIMENTS (tools/build_variants.sh); the shipped one refuses them.
    if (var >= 21 && var <= 26) {
#ifdef TPG_PW_EXPERIMENTS
      if (var == 24) TPG_TRY((pw_launch_wg<4, 2, TPG_PW_FOR_AS, 4, 1>(ctx, pw, v, kg0, kg1, "pairwise_mfma_as")));
      else if (var == 25) TPG_TRY((pw_launch_wg<4, 2, TPG_PW_FOR_AS, 4, 2>(ctx, pw, v, kg0, kg1, "pairwise_mfma_as")));
      else if (var == 26) TPG_TRY((pw_launch_wg<4, 2, TPG_PW_FOR_AS, 4, 3>(ctx, pw, v, kg0, kg1, "pairwise_mfma_as")));
      else if (var == 21) TPG_TRY((pw_launch_set<4, 2, TPG_PW_FOR_AS, 5, 1>(ctx, pw, v, kg0, kg1, "pairwise_mfma_as")));
      else if (var == 22) TPG_TRY((pw_launch_set<4, 2, TPG_PW_FOR_AS, 5, 2>(ctx, pw, v, kg0, kg1, "pairwise_mfma_as")));
      else TPG_REQUIRE(false, TPG_EUNSUPPORTED, "TPG_PW_VARIANT=%d does not exist", var);
#else
      TPG_REQUIRE(false, TPG_EUNSUPPORTED, "TPG_PW_VARIANT=%d is a timing experiment with wrong sums: build with -DTPG_PW_EXPERIMENTS", var);
#endif
    }
    else if (var == 1) PW_SET(4, 2, TPG_PW_FOR_AS, 5, "pairwise_mfma_as");
    else if (var == 2) PW_SET(3, 2, TPG_PW_FOR_AS, 5, "pairwise_mfma_as");
    // (96 x 96 = 18 accumulator tiles, 32 of their registers VGPRs: 15.0 ms with three slots, 21.7 with four -- 54 spills --
    // against 8.1; instantiations removed again)
    else PW_SET(4, 2, TPG_PW_FOR_AS, 4, "pairwise_mfma_as");  // (round 6: 8.1 ms with four slots, 8.5 with five; round 4 had it the other way, before the lean scalar stream)
  } else if (set == TPG_PW_FOR_IBS_ALONE) {
    // measured at 5 000 x 1 000 000 (tools/pw_only.py, one GPU job): 128 x 64 with 4 / 5 / 3 slots 10.7 / 10.8 / 13.3 ms, 96 x 64
    // with 4 / 5 slots 11.2 / 11.2, 64 x 64 11.9; the {V, D, H} kernel (three sums per pair, 64 x 64) 12.2
    if (var == 1) PW_SET(4, 2, TPG_PW_FOR_IBS_ALONE, 4, "pairwise_mfma_ibs1");
    else if (var == 2) PW_SET(3, 2, TPG_PW_FOR_IBS_ALONE, 4, "pairwise_mfma_ibs1");
    else if (var == 3) PW_SET(3, 2, TPG_PW_FOR_IBS_ALONE, 5, "pairwise_mfma_ibs1");
    else if (var == 4) PW_SET(2, 2, TPG_PW_FOR_IBS_ALONE, 5, "pairwise_mfma_ibs1");
    else if (var == 5) PW_SET(4, 2, TPG_PW_FOR_IBS_ALONE, 5, "pairwise_mfma_ibs1");
    else if (var == 6) PW_SET(4, 2, TPG_PW_FOR_IBS_ALONE, 3, "pairwise_mfma_ibs1");
    // (96 x 96, 27 MFMAs per 6 fragments: 17.9 ms with three slots, 21.2 with four -- 86 spills; removed again)
    else PW_SET(4, 2, TPG_PW_FOR_IBS_ALONE, 4, "pairwise_mfma_ibs1");
  } else if (set == TPG_PW_FOR_IBS) {
    if (var == 1) PW_SET(2, 2, TPG_PW_FOR_IBS, 4, "pairwise_mfma_ibs");
    else if (var == 2) PW_SET(3, 1, TPG_PW_FOR_IBS, 6, "pairwise_mfma_ibs");
    else PW_SET(2, 2, TPG_PW_FOR_IBS, 5, "pairwise_mfma_ibs");
  } else if (set == TPG_PW_FOR_KING) {
    if (var == 1) PW_SET(2, 2, TPG_PW_FOR_KING, 4, "pairwise_mfma_king");
    else if (var == 2) PW_SET(3, 1, TPG_PW_FOR_KING, 6, "pairwise_mfma_king");
    else PW_SET(2, 2, TPG_PW_FOR_KING, 5, "pairwise_mfma_king");
  } else {
    if (var == 1) PW_SET(3, 1, TPG_PW_ALL, 4, "pairwise_mfma");
    else TPG_TRY(pw_launch_all(ctx, pw, v, kg0, kg1));
  }
#undef PW_SET
#undef PW_WG
  TPG_CHECK_LAUNCH();
  pw->loci += col_end - col_begin;
  // what stays complete: the products of this set; the D + H sum (plane D + plane H) after anything that added D and H, apart or together
  pw->have &= set | (((set & TPG_PW_D) && (set & TPG_PW_H)) ? TPG_PW_DH : 0);
  return TPG_OK;
}

extern "C" int tpg_pairwise_accumulate(tpg_ctx* ctx, tpg_pairwise* pw, const tpg_view* v, int64_t col_begin,
                                       int64_t col_end) {
  return tpg_pairwise_accumulate_products(ctx, pw, v, col_begin, col_end, TPG_PW_ALL);
}

// (TPG_PW_DH is reported only where it is all there is of D and H: a caller that never uses it sees the four bits it always saw)
extern "C" int tpg_pairwise_products(const tpg_pairwise* pw) {
  if (!pw) return 0;
  const bool apart = (pw->have & TPG_PW_D) && (pw->have & TPG_PW_H);
  return (pw->have & TPG_PW_ALL) | ((pw->have & TPG_PW_DH) && !apart ? TPG_PW_DH : 0);
}

// ---------------------------------------------------------------------------
// epilogues.  FP64 arithmetic follows the R drivers' operation order (file compiled with
// -ffp-contract=off).
struct PwCounts {
  int V, D, H, Aij, Aji;  // relative to the OUTPUT element (row i, column j): Aij = loci where i is het and j typed
};

#define TPG_NAN __longlong_as_double(0x7FF8000000000000ll)

// mode: 0 raw counts (six optional outputs), 1 IBS proportion / adjusted counts (scale), 2 KING,
// 3 allele sharing, 4 IBS + KING + allele sharing together.  Sums of products are formed in 64 bits (a single
// product is bounded by the loci accumulated, < 2^31; V + D + H and 2 V are not).  quirk = reference quirk Q1: the
// number of narrower blocks of the R driver, each of which adds 1 to every allele-sharing numerator
// (src/snp_as.cpp:57-63); 0 unless the caller asked for the emulation.
__device__ __forceinline__ void tpg_pw_emit(const PwCounts c, int mode, double scale, long long quirk, int64_t idx,
                                            double* __restrict__ o0, double* __restrict__ o1, double* __restrict__ o2,
                                            double* __restrict__ o3, double* __restrict__ o4, double* __restrict__ o5) {
  const long long V = c.V, D = c.D, H = c.H, Aij = c.Aij, Aji = c.Aji;
  if (mode == 0) {
    if (o0) o0[idx] = (double)(V + D + H);
    if (o1) o1[idx] = (double)(2 * V);
    if (o2) o2[idx] = (double)(D - V + Aij + Aji);
    if (o3) o3[idx] = (double)Aij;
    if (o4) o4[idx] = (double)(D + quirk);
    if (o5) o5[idx] = (double)V;
  } else if (mode == 1) {
    const double prop = (double)(V + D + H) / (double)(2 * V);  // R/snp_ibs.R:88-95
    o0[idx] = scale == 1.0 ? prop : prop * scale;               // :100
  } else if (mode == 2) {
    const double K = (double)(D - V + Aij + Aji), Ni = (double)Aij, Nj = (double)Aji;
    const double mn = Ni < Nj ? Ni : Nj;
    o0[idx] = K / (2 * mn) + 0.5 - 0.25 * (Ni + Nj) / mn;  // R/snp_king.R:86-89
  } else if (mode == 3) {
    const double num = (double)(D + quirk), den = (double)V;
    o0[idx] = V == 0 ? TPG_NAN : 0.5 * (1 + num / den);  // R/snp_allele_sharing.R:79-80
  } else {
    // mode 4: IBS (o0, scale), KING (o1) and allele sharing (o2) from one fetch
    if (o0) {
      const double prop = (double)(V + D + H) / (double)(2 * V);
      o0[idx] = scale == 1.0 ? prop : prop * scale;
    }
    if (o1) {
      const double K = (double)(D - V + Aij + Aji), Ni = (double)Aij, Nj = (double)Aji;
      const double mn = Ni < Nj ? Ni : Nj;
      o1[idx] = K / (2 * mn) + 0.5 - 0.25 * (Ni + Nj) / mn;
    }
    if (o2) o2[idx] = V == 0 ? TPG_NAN : 0.5 * (1 + (double)(D + quirk) / (double)V);
  }
}

// One workgroup per 32 x 32 tile (ti <= tj) of the stored band.  The five count planes of the tile are read once,
// lane index on the slab's lane (contiguous 4-byte loads), into LDS; the tile is then written twice -- as it stands,
// threads running down a column of the column-major outputs, and mirrored for the lower triangle, threads running
// along the slab's column index -- so that every store instruction writes 32 contiguous doubles and no slab element
// is fetched more than once.
__global__ __launch_bounds__(256) void tpg_pairwise_epilogue_kernel(const int32_t* __restrict__ acc,
                                                                    const int64_t* __restrict__ rowpad, int ti0,
                                                                    int nst, int n,
                                                                    int mode, double scale, long long quirk,
                                                                    double* __restrict__ o0,
                                                                    double* __restrict__ o1, double* __restrict__ o2,
                                                                    double* __restrict__ o3, double* __restrict__ o4,
                                                                    double* __restrict__ o5) {
  const int ti = blockIdx.y + ti0, tj = blockIdx.x + ti0;  // tile rows of this rank's band, columns from its first one
  if (ti > tj) return;
  __shared__ int sp[5][32][33];
  const int32_t* p = acc + (tpg_pw_unit_index(nst, ti / TA, tj) + rowpad[ti / TA]) * TPG_PW_TILE_INTS + ((ti % TA) * 16) * 64;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int idx = threadIdx.x + 256 * e, reg = idx >> 6, lane = idx & 63;
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31;
#pragma unroll
    for (int q = 0; q < 5; q++) sp[q][row][col] = p[q * TPG_PW_PLANE_INTS + reg * 64 + lane];
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int idx = threadIdx.x + 256 * e;
    {  // as stored: output row = 32 ti + row (contiguous), column = 32 tj + col
      const int row = idx & 31, col = idx >> 5;
      const int gi = 32 * ti + row, gj = 32 * tj + col;
      if (gi < n && gj < n) {
        const PwCounts c = {sp[0][row][col], sp[1][row][col], sp[2][row][col], sp[3][row][col], sp[4][row][col]};
        tpg_pw_emit(c, mode, scale, quirk, gi + (int64_t)gj * n, o0, o1, o2, o3, o4, o5);
      }
    }
    if (ti != tj) {  // mirrored: output row = 32 tj + col (contiguous), column = 32 ti + row
      const int col = idx & 31, row = idx >> 5;
      const int gi = 32 * ti + row, gj = 32 * tj + col;
      if (gi < n && gj < n) {
        const PwCounts c = {sp[0][row][col], sp[1][row][col], sp[2][row][col], sp[4][row][col], sp[3][row][col]};
        tpg_pw_emit(c, mode, scale, quirk, gj + (int64_t)gi * n, o0, o1, o2, o3, o4, o5);
      }
    }
  }
}

// The part of the N x N outputs a rank's accumulators cover: all of it for unsharded (or not yet reduced)
// accumulators, else rows [r0, r1) x columns [r0, n) (as stored) and its mirror image rows [r0, n) x columns [r0, r1).
struct PwBand {
  int ti0, nti;     // first tile row, number of tile rows
  int64_t r0, r1;   // individuals
  bool whole;
};
static PwBand pw_band(const tpg_pairwise* pw) {
  PwBand b;
  const int nt = (int)ceil_div(pw->n, 32);
  if (!pw->reduced || pw->nranks == 1) return PwBand{0, nt, 0, pw->n, true};
  const int i0 = pw->band[(size_t)pw->rank], i1 = pw->band[(size_t)pw->rank + 1];
  b.ti0 = std::min(nt, TA * i0);
  b.nti = std::min(nt, TA * i1) - b.ti0;
  b.r0 = std::min<int64_t>(pw->n, 32 * TA * (int64_t)i0);
  b.r1 = std::min<int64_t>(pw->n, 32 * TA * (int64_t)i1);
  b.whole = false;
  return b;
}

// device -> caller for an output the caller holds in host memory: everything, or just the band's two rectangles
static int pw_commit(tpg_ctx* ctx, OutBuf& o, int64_t n, const PwBand& b) {
  if (!o.owned || !o.user) return TPG_OK;  // the caller's pointer is device memory: written in place
  if (b.whole) return o.commit(ctx);
  if (b.r1 > b.r0) {
    const size_t pitch = sizeof(double) * (size_t)n, offs = (size_t)b.r0 + (size_t)b.r0 * (size_t)n;
    TPG_HIP(hipMemcpy2DAsync((double*)o.user + offs, pitch, (const double*)o.d + offs, pitch,
                             sizeof(double) * (size_t)(b.r1 - b.r0), (size_t)(n - b.r0), hipMemcpyDeviceToHost, ctx->stream));
    TPG_HIP(hipMemcpy2DAsync((double*)o.user + offs, pitch, (const double*)o.d + offs, pitch,
                             sizeof(double) * (size_t)(n - b.r0), (size_t)(b.r1 - b.r0), hipMemcpyDeviceToHost, ctx->stream));
  }
  TPG_HIP(hipStreamSynchronize(ctx->stream));
  return TPG_OK;
}

// an output may only be formed from products every accumulate since the last zero has added (tpg_pairwise_accumulate_products)
static int pw_need(const tpg_pairwise* pw, int products, const char* what) {
  TPG_REQUIRE((pw->have & products) == products, TPG_EINVAL,
              "%s needs the products 0x%x but only 0x%x were accumulated (TPG_PW_V = 1, D = 2, H = 4, A = 8, D + H as one sum = 16): pass them to "
              "tpg_pairwise_accumulate_products", what, products, pw->have);
  return TPG_OK;
}

static int run_epilogue(tpg_ctx* ctx, const tpg_pairwise* pw, int mode, double scale, double* outs[6]) {
  const size_t bytes = sizeof(double) * (size_t)pw->n * (size_t)pw->n;
  const PwBand band = pw_band(pw);
  OutBuf b[6];
  for (int k = 0; k < 6; k++)
    if (outs[k]) TPG_TRY(b[k].init(outs[k], bytes));
  const unsigned nt = (unsigned)ceil_div(pw->n, 32);
  if (band.nti > 0) {
    TPG_LAUNCH(ctx, "pairwise_epilogue", tpg_pairwise_epilogue_kernel, dim3(nt - (unsigned)band.ti0, (unsigned)band.nti),
               dim3(256), 0, (const int32_t*)pw->acc, (const int64_t*)pw->rowpad, band.ti0, (int)pw->nst, (int)pw->n, mode,
               scale, (long long)pw->as_pad_quirk, b[0].dev<double>(), b[1].dev<double>(), b[2].dev<double>(),
               b[3].dev<double>(), b[4].dev<double>(), b[5].dev<double>());
    TPG_CHECK_LAUNCH();
  }
  for (int k = 0; k < 6; k++)
    if (outs[k]) TPG_TRY(pw_commit(ctx, b[k], pw->n, band));
  return TPG_OK;
}

extern "C" int tpg_pairwise_counts(tpg_ctx* ctx, const tpg_pairwise* pw, double* ibs, double* ibs_valid,
                                   double* king_num, double* n_Aa_i, double* as_num, double* as_den) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && pw, TPG_EINVAL, "null argument");
  if (ibs) TPG_TRY(pw_need(pw, TPG_PW_FOR_IBS_ALONE, "ibs"));
  if (ibs_valid) TPG_TRY(pw_need(pw, TPG_PW_V, "ibs_valid"));
  if (king_num) TPG_TRY(pw_need(pw, TPG_PW_FOR_KING, "king_num"));
  if (n_Aa_i) TPG_TRY(pw_need(pw, TPG_PW_A, "n_Aa_i"));
  if (as_num) TPG_TRY(pw_need(pw, TPG_PW_D, "as_num"));
  if (as_den) TPG_TRY(pw_need(pw, TPG_PW_V, "as_den"));
  double* outs[6] = {ibs, ibs_valid, king_num, n_Aa_i, as_num, as_den};
  return run_epilogue(ctx, pw, 0, 1.0, outs);
}

extern "C" int tpg_pairwise_ibs(tpg_ctx* ctx, const tpg_pairwise* pw, int type, int64_t m, double* out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && pw && out, TPG_EINVAL, "null argument");
  TPG_REQUIRE(type == TPG_IBS_PROPORTION || type == TPG_IBS_ADJUSTED_COUNTS, TPG_EINVAL, "bad IBS type %d", type);
  TPG_TRY(pw_need(pw, TPG_PW_FOR_IBS_ALONE, "IBS"));
  double* outs[6] = {out, nullptr, nullptr, nullptr, nullptr, nullptr};
  return run_epilogue(ctx, pw, 1, type == TPG_IBS_PROPORTION ? 1.0 : (double)m, outs);
}

extern "C" int tpg_pairwise_king(tpg_ctx* ctx, const tpg_pairwise* pw, double* out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && pw && out, TPG_EINVAL, "null argument");
  TPG_TRY(pw_need(pw, TPG_PW_FOR_KING, "KING"));
  double* outs[6] = {out, nullptr, nullptr, nullptr, nullptr, nullptr};
  return run_epilogue(ctx, pw, 2, 1.0, outs);
}

extern "C" int tpg_pairwise_allele_sharing(tpg_ctx* ctx, const tpg_pairwise* pw, double* out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && pw && out, TPG_EINVAL, "null argument");
  TPG_TRY(pw_need(pw, TPG_PW_FOR_AS, "allele sharing"));
  double* outs[6] = {out, nullptr, nullptr, nullptr, nullptr, nullptr};
  return run_epilogue(ctx, pw, 3, 1.0, outs);
}

static int grm_from_as_band(tpg_ctx* ctx, tpg_comm* comm, int n, const PwBand& band, double* d_M);

// IBS, KING, allele sharing and GRM in one pass over the accumulators (any output may be NULL)
static int epilogues_impl(tpg_ctx* ctx, tpg_comm* comm, const tpg_pairwise* pw, int ibs_type, int64_t m, double* ibs,
                          double* king, double* allele_sharing, double* grm) {
  TPG_REQUIRE(ctx && pw, TPG_EINVAL, "null argument");
  TPG_REQUIRE(ibs_type == TPG_IBS_PROPORTION || ibs_type == TPG_IBS_ADJUSTED_COUNTS, TPG_EINVAL, "bad IBS type");
  // every rank has accumulated the same product set (the sharded entry points pass the same mask everywhere)
  if (ibs) TPG_TRY(pw_need(pw, TPG_PW_FOR_IBS_ALONE, "IBS"));
  if (king) TPG_TRY(pw_need(pw, TPG_PW_FOR_KING, "KING"));
  if (allele_sharing || grm) TPG_TRY(pw_need(pw, TPG_PW_FOR_AS, "allele sharing / GRM"));
  const int n = (int)pw->n;
  const PwBand band = pw_band(pw);
  TPG_REQUIRE(band.whole || !grm || comm, TPG_EINVAL,
              "the GRM of reduced, sharded accumulators needs the communicator (its mean runs over all pairs): "
              "use tpg_pairwise_epilogues_sharded");
  const size_t bytes = sizeof(double) * (size_t)n * (size_t)n;
  OutBuf bi, bk, ba, bg;
  auto staging = [&]() -> int {
    if (ibs) TPG_TRY(bi.init(ibs, bytes));
    if (king) TPG_TRY(bk.init(king, bytes));
    if (allele_sharing) TPG_TRY(ba.init(allele_sharing, bytes));
    if (grm) TPG_TRY(bg.init(grm, bytes));
    return TPG_OK;
  };
  int lrc = staging();
  // Host outputs are staged in device buffers whose allocation can fail on one rank alone; the GRM then exchanges its
  // mean over the ranks, so they agree on a status first (every rank passes the same kind of pointers: all callers of the
  // sharded entry point do).  Device outputs allocate nothing: no exchange of a status either.
  const bool staged = (ibs && !tpg_is_device_ptr(ibs)) || (king && !tpg_is_device_ptr(king)) ||
                      (allele_sharing && !tpg_is_device_ptr(allele_sharing)) || (grm && !tpg_is_device_ptr(grm));
  if (grm && comm && !band.whole && staged) lrc = tpg_comm_agree(comm, lrc);
  TPG_TRY(lrc);
  // GRM needs the allele-sharing matrix: write it into the GRM buffer when the caller does not want both
  double* as_dst = allele_sharing ? ba.dev<double>() : bg.dev<double>();
  const unsigned nt = (unsigned)ceil_div(pw->n, 32);
  if (band.nti > 0) {
    TPG_LAUNCH(ctx, "pairwise_epilogue", tpg_pairwise_epilogue_kernel, dim3(nt - (unsigned)band.ti0, (unsigned)band.nti),
               dim3(256), 0, (const int32_t*)pw->acc, (const int64_t*)pw->rowpad, band.ti0, (int)pw->nst, n, 4,
               ibs_type == TPG_IBS_PROPORTION ? 1.0 : (double)m, (long long)pw->as_pad_quirk, bi.dev<double>(),
               bk.dev<double>(), as_dst, (double*)nullptr, (double*)nullptr, (double*)nullptr);
    TPG_CHECK_LAUNCH();
  }
  if (grm) {
    if (band.whole) {
      if (allele_sharing) TPG_HIP(hipMemcpyAsync(bg.dev<double>(), ba.dev<double>(), bytes, hipMemcpyDeviceToDevice, ctx->stream));
      TPG_TRY(grm_from_as_band(ctx, nullptr, n, band, bg.dev<double>()));
    } else {
      if (allele_sharing && band.r1 > band.r0) {  // copy the band's two rectangles
        const size_t pitch = sizeof(double) * (size_t)n, offs = (size_t)band.r0 + (size_t)band.r0 * (size_t)n;
        TPG_HIP(hipMemcpy2DAsync(bg.dev<double>() + offs, pitch, ba.dev<double>() + offs, pitch,
                                 sizeof(double) * (size_t)(band.r1 - band.r0), (size_t)(n - band.r0), hipMemcpyDeviceToDevice, ctx->stream));
        TPG_HIP(hipMemcpy2DAsync(bg.dev<double>() + offs, pitch, ba.dev<double>() + offs, pitch,
                                 sizeof(double) * (size_t)(n - band.r0), (size_t)(band.r1 - band.r0), hipMemcpyDeviceToDevice, ctx->stream));
      }
      TPG_TRY(grm_from_as_band(ctx, comm, n, band, bg.dev<double>()));
    }
  }
  if (ibs) TPG_TRY(pw_commit(ctx, bi, n, band));
  if (king) TPG_TRY(pw_commit(ctx, bk, n, band));
  if (allele_sharing) TPG_TRY(pw_commit(ctx, ba, n, band));
  if (grm) TPG_TRY(pw_commit(ctx, bg, n, band));
  return TPG_OK;
}

extern "C" int tpg_pairwise_epilogues(tpg_ctx* ctx, const tpg_pairwise* pw, int ibs_type, int64_t m, double* ibs,
                                      double* king, double* allele_sharing, double* grm) {
  TpgEnter _enter(ctx);
  return epilogues_impl(ctx, nullptr, pw, ibs_type, m, ibs, king, allele_sharing, grm);
}

// The same on accumulators that tpg_pairwise_reduce left sharded: every rank finishes ITS band (1 / nranks of the
// tiles) and writes only that part of the outputs; the one number that needs all pairs, the mean of the off-diagonal
// allele-sharing values behind the GRM, is summed over the ranks (two doubles).
extern "C" int tpg_pairwise_epilogues_sharded(tpg_ctx* ctx, tpg_comm* comm, const tpg_pairwise* pw, int ibs_type,
                                              int64_t m, double* ibs, double* king, double* allele_sharing, double* grm) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(comm && comm->ctx == ctx, TPG_EINVAL, "bad communicator");
  return epilogues_impl(ctx, comm, pw, ibs_type, m, ibs, king, allele_sharing, grm);
}

// GRM (R/pairwise_grm.R:42-50): mb = mean of the off-diagonal allele-sharing values (na.rm), then
// 2 (M - mb) / (1 - mb).  The mean stays on the device (no host round trip in the middle of the step): per-block
// partial sums in double over the stored elements above the diagonal, doubled (the matrix is symmetric), one thread
// adds the partials in a fixed order.
// Band form (one band = everything for unsharded accumulators): rows [r0, r1), columns > row; the partial sums of
// the ranks are added by an all-reduce; then the band's two rectangles are rescaled.
__global__ __launch_bounds__(256) void tpg_offdiag_sum_band_kernel(const double* __restrict__ M, int n, int r0, int r1,
                                                                   double* __restrict__ part_sum,
                                                                   double* __restrict__ part_cnt) {
  __shared__ double ssum[256], scnt[256];
  double s = 0, c = 0;
  // a workgroup takes whole columns j (the stored elements above the diagonal are the rows i < j of column j: contiguous)
  for (int j = blockIdx.x; j < n; j += gridDim.x) {
    const int i1 = r1 < j ? r1 : j;
    const double* col = M + (int64_t)j * n;
    for (int i = r0 + (int)threadIdx.x; i < i1; i += 256) {
      const double x = col[i];
      if (x == x) { s += x; c += 1; }
    }
  }
  ssum[threadIdx.x] = s;
  scnt[threadIdx.x] = c;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { ssum[threadIdx.x] += ssum[threadIdx.x + w]; scnt[threadIdx.x] += scnt[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part_sum[blockIdx.x] = ssum[0]; part_cnt[blockIdx.x] = scnt[0]; }
}

// out = 2 * (sum a, sum b): 256 threads, every thread a strided share in index order, then a fixed tree (the same sums on
// every run)
__global__ __launch_bounds__(256) void tpg_sum2_kernel(const double* __restrict__ a, const double* __restrict__ b, int nb,
                                                       double* __restrict__ out) {
  __shared__ double ss[256], sc[256];
  double s = 0, c = 0;
  for (int k = threadIdx.x; k < nb; k += 256) { s += a[k]; c += b[k]; }
  ss[threadIdx.x] = s;
  sc[threadIdx.x] = c;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { ss[threadIdx.x] += ss[threadIdx.x + w]; sc[threadIdx.x] += sc[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = 2 * ss[0];  // both triangles
    out[1] = 2 * sc[0];
  }
}

// rows [ra, rb) x columns [ca, cb) of the column-major n x n matrix
__global__ void tpg_grm_rect_kernel(double* __restrict__ M, int n, int ra, int rb, int ca, int cb, const double* __restrict__ sc) {
  const double mb = sc[0] / sc[1];
  const int64_t rows = rb - ra, total = rows * (int64_t)(cb - ca);
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = (ra + idx % rows) + (ca + idx / rows) * (int64_t)n;
    M[o] = (M[o] - mb) / (1 - mb) * 2;
  }
}

static int grm_from_as_band(tpg_ctx* ctx, tpg_comm* comm, int n, const PwBand& band, double* d_M) {
  const int NB = 2048;
  double* d_part = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_part, sizeof(double) * (2 * NB + 2)));
  double* d_sc = d_part + 2 * NB;
  int rc = TPG_OK;
  hipError_t e = hipMemsetAsync(d_part, 0, sizeof(double) * (2 * NB + 2), ctx->stream);
  if (e == hipSuccess) {
    if (band.r1 > band.r0)
      TPG_LAUNCH(ctx, "grm_offdiag_sum", tpg_offdiag_sum_band_kernel, dim3(NB), dim3(256), 0, (const double*)d_M, n,
                 (int)band.r0, (int)band.r1, d_part, d_part + NB);
    TPG_LAUNCH(ctx, "grm_offdiag_sum", tpg_sum2_kernel, dim3(1), dim3(256), 0, (const double*)d_part,
               (const double*)(d_part + NB), NB, d_sc);
    if (comm) rc = tpg_comm_allreduce(comm, d_sc, 2, 1);  // sum and count over all ranks
    if (rc == TPG_OK && band.r1 > band.r0) {
      TPG_LAUNCH(ctx, "grm_scale", tpg_grm_rect_kernel, dim3(1024), dim3(256), 0, d_M, n, (int)band.r0, (int)band.r1,
                 (int)band.r0, n, (const double*)d_sc);  // as stored
      if (n > band.r1)
        TPG_LAUNCH(ctx, "grm_scale", tpg_grm_rect_kernel, dim3(1024), dim3(256), 0, d_M, n, (int)band.r1, n, (int)band.r0,
                   (int)band.r1, (const double*)d_sc);  // mirror image below the band's diagonal block
    }
    e = hipGetLastError();
  }
  tpg_pfree(d_part);  // stream-ordered: the pool hands it out again behind the kernels above
  if (e != hipSuccess) { tpg_set_error("grm (band): %s", hipGetErrorString(e)); return TPG_EHIP; }
  return rc;
}

extern "C" int tpg_pairwise_grm(tpg_ctx* ctx, const tpg_pairwise* pw, double* out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && pw && out, TPG_EINVAL, "null argument");
  return tpg_pairwise_epilogues(ctx, pw, TPG_IBS_PROPORTION, 0, nullptr, nullptr, nullptr, out);
}

// ---------------------------------------------------------------------------
// Block means of an N x N matrix by groups of individuals: the reduction pop_fst / pop_fis (WG17) make of the
// allele-sharing matrix, mean(Mij[p1, p2], na.rm = TRUE) for every pair of populations with the diagonal set
// to NA (R/pop_fst.R:40-63, R/pop_fis.R:151-173).  Deterministic: kernel A sums, for one column j and one
// group g, the rows of g in index order; kernel B sums those per-column results over the columns of a group in
// index order -- the column-major order R's mean() walks the sub-matrix in.
__global__ __launch_bounds__(64) void tpg_block_colsum_kernel(const double* __restrict__ A, int64_t n, int G,
                                                              const int32_t* __restrict__ perm,
                                                              const int32_t* __restrict__ goff, int skip_diag,
                                                              double* __restrict__ colsum, double* __restrict__ colcnt) {
  const int64_t j = blockIdx.x;
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    double s = 0, k = 0;
    for (int t = goff[g]; t < goff[g + 1]; t++) {
      const int64_t i = perm[t];
      const double v = A[i + j * n];
      if (v == v && !(skip_diag && i == j)) { s += v; k += 1; }
    }
    colsum[j * G + g] = s;
    colcnt[j * G + g] = k;
  }
}

__global__ void tpg_block_combine_kernel(const double* __restrict__ colsum, const double* __restrict__ colcnt, int G,
                                         const int32_t* __restrict__ perm, const int32_t* __restrict__ goff,
                                         double* __restrict__ mean, double* __restrict__ count) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G * G) return;
  const int g1 = idx % G, g2 = idx / G;  // rows of g1, columns of g2
  double s = 0, k = 0;
  for (int t = goff[g2]; t < goff[g2 + 1]; t++) {
    const int64_t j = perm[t];
    s += colsum[j * G + g1];
    k += colcnt[j * G + g1];
  }
  mean[idx] = s / k;  // 0 / 0 = NaN = mean(numeric(0))
  if (count) count[idx] = k;
}

extern "C" int tpg_block_means(tpg_ctx* ctx, const double* A, int64_t n, const int32_t* groupIds0, int ngroups,
                               int skip_diag, double* mean, double* count) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && A && groupIds0 && mean, TPG_EINVAL, "null argument");
  TPG_REQUIRE(n > 0 && ngroups > 0, TPG_EINVAL, "bad n = %lld or ngroups = %d", (long long)n, ngroups);
  std::vector<int32_t> goff((size_t)ngroups + 1, 0), perm((size_t)n);
  for (int64_t i = 0; i < n; i++) {
    TPG_REQUIRE(groupIds0[i] >= 0 && groupIds0[i] < ngroups, TPG_EINVAL, "groupIds[%lld] = %d out of [0,%d)",
                (long long)i, groupIds0[i], ngroups);
    goff[(size_t)groupIds0[i] + 1]++;
  }
  for (int g = 0; g < ngroups; g++) goff[(size_t)g + 1] += goff[(size_t)g];
  {
    std::vector<int32_t> fill(goff.begin(), goff.end() - 1);
    for (int64_t i = 0; i < n; i++) perm[(size_t)fill[(size_t)groupIds0[i]]++] = (int32_t)i;  // index order inside a group
  }
  InBuf ia, ip, ig;
  TPG_TRY(ia.init(ctx, A, sizeof(double) * (size_t)n * (size_t)n));
  TPG_TRY(ip.init(ctx, perm.data(), sizeof(int32_t) * (size_t)n));
  TPG_TRY(ig.init(ctx, goff.data(), sizeof(int32_t) * ((size_t)ngroups + 1)));
  const size_t gg = (size_t)ngroups * (size_t)ngroups;
  OutBuf om, oc;
  TPG_TRY(om.init(mean, sizeof(double) * gg));
  if (count) TPG_TRY(oc.init(count, sizeof(double) * gg));
  double *d_cs = nullptr, *d_cc = nullptr;
  hipError_t e = tpg_pmalloc((void**)&d_cs, sizeof(double) * (size_t)n * (size_t)ngroups);
  if (e == hipSuccess) e = tpg_pmalloc((void**)&d_cc, sizeof(double) * (size_t)n * (size_t)ngroups);
  int rc = TPG_OK;
  if (e == hipSuccess) {
    [&]() -> int {
      TPG_LAUNCH(ctx, "block_colsum", tpg_block_colsum_kernel, dim3((unsigned)n), dim3(64), 0, ia.dev<double>(), n,
                 ngroups, ip.dev<int32_t>(), ig.dev<int32_t>(), skip_diag, d_cs, d_cc);
      TPG_LAUNCH(ctx, "block_combine", tpg_block_combine_kernel, dim3((unsigned)ceil_div((int64_t)gg, 256)), dim3(256), 0,
                 (const double*)d_cs, (const double*)d_cc, ngroups, ip.dev<int32_t>(), ig.dev<int32_t>(),
                 om.dev<double>(), oc.dev<double>());
      return TPG_OK;
    }();
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  tpg_pfree(d_cs);
  tpg_pfree(d_cc);
  if (e != hipSuccess) { tpg_set_error("block means: %s", hipGetErrorString(e)); return TPG_EHIP; }
  TPG_TRY(rc);
  TPG_TRY(om.commit(ctx));
  if (count) TPG_TRY(oc.commit(ctx));
  return TPG_OK;
}

// ---------------------------------------------------------------------------
// filter_high_relatedness (R/filter_high_relatedness.R:26-145): greedy pruning of a relatedness (KING) matrix -- for
// every pair above the threshold, met in the order of decreasing mean relatedness, drop the individual whose row mean
// exceeds the mean of ALL other rows (the reference's mean(matrix2[-j, ]), :104).  A sequential, data-dependent loop:
// host logic.  The matrix may live in HBM (the KING matrix tpg_pairwise_king left there) or in host memory.  The
// reference recomputes both means from scratch for every comparison (O(N^2) each); here row sums, counts and the
// grand total are kept up to date in long double, and a comparison whose two means are closer than 1e-12 relative is
// recomputed literally -- R's two-pass long-double mean() over the same entries in the same order -- so the decisions
// are R's.
#include "host/host_addcounts.h"  // tpg_add_counts_u16, tpg_add_counts_i32
#include "host/host_relfilter.h"  // r_mean_ld, tpg_host_filter_high_relatedness

extern "C" int tpg_filter_high_relatedness(tpg_ctx* ctx, const double* matrix, int64_t n, double kings_threshold,
                                           uint8_t* keep, int32_t* new_order0) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && matrix && keep, TPG_EINVAL, "null argument");
  TPG_REQUIRE(n > 0 && n < (1 << 24), TPG_EINVAL, "bad n = %lld", (long long)n);
  TPG_REQUIRE(kings_threshold == kings_threshold, TPG_EINVAL, "kings_threshold is NA");
  const size_t N = (size_t)n;
  std::vector<double> A(N * N);
  if (tpg_is_device_ptr(matrix)) {
    TPG_HIP(hipMemcpyAsync(A.data(), matrix, sizeof(double) * N * N, hipMemcpyDeviceToHost, ctx->stream));
    TPG_HIP(hipStreamSynchronize(ctx->stream));
  } else {
    memcpy(A.data(), matrix, sizeof(double) * N * N);
  }
  std::string err;
  const int rc = tpg_host_filter_high_relatedness(A, n, kings_threshold, keep, new_order0, err);
  TPG_REQUIRE(rc == 0, TPG_EINVAL, "%s", err.c_str());
  return TPG_OK;
}

// ---------------------------------------------------------------------------
// Literal per-block mirrors of increment_{ibs,king,as}_counts.  The R drivers call these once per locus block (38 times
// at 5 000 x 1 000 000 with the default block size) on the same FBM and the same pair of N x N accumulators.
//
// Nothing of the caller's FBM is kept between calls: every call uploads the columns of ITS block (the blocks of a driver
// loop cover the FBM once, so the loop moves the same bytes a whole upload would) -- an FBM whose bytes change between two
// analyses (bigsnpr::snp_fastImputeSimple rewrites the backing file in place, R/gt_impute_simple.R:86), or another array
// at a reused address, can therefore never be served from a stale HBM copy.
// Default: literal semantics, K and K2 are incremented when the call returns (src/snp_ibs.cpp:67-72).  After
// tpg_increment_defer(ctx, 1) every (K, K2) pair gets device accumulators that live across the calls of the block loop and
// the sums reach the caller's matrices at tpg_increment_flush: one download of two N x N matrices per analysis instead of
// per block.
struct ResidentAcc {
  int which;  // 0 IBS, 1 KING, 2 allele sharing
  double *A, *B;
  std::vector<int32_t> rows;
  tpg_pairwise* pw;
};
struct Resident {
  bool defer = false;
  std::vector<ResidentAcc> accs;
  tpg_pairwise* spare = nullptr;  // the accumulators of the last immediate call, reused while n stays the same
  int32_t* stage = nullptr;       // pinned host staging of the two int32 count matrices (add_counts_to_caller)
  size_t stage_ints = 0;
  const double *last_A = nullptr, *last_B = nullptr;  // the matrices of the previous immediate call (never dereferenced)
};

static Resident* resident_of(tpg_ctx* ctx) {
  if (!ctx->resident) ctx->resident = new Resident();
  return (Resident*)ctx->resident;
}

void tpg_resident_release(tpg_ctx* ctx) {  // called by tpg_ctx_destroy and tpg_resident_drop
  Resident* r = (Resident*)ctx->resident;
  if (!r) return;
  for (auto& a : r->accs) tpg_pairwise_free(a.pw);
  tpg_pairwise_free(r->spare);
  if (r->stage) (void)hipHostFree(r->stage);
  delete r;
  ctx->resident = nullptr;
}

// The two count matrices of one increment_* entry point as int32, both triangles, column-major: half the bytes of the
// double outputs of tpg_pairwise_counts on their way to the host.  which 0: IBS = V + D + H, 2 V; 1: KING numerator
// D - V + A + A', A; 2: D (+ quirk), V.  Same tile walk as tpg_pairwise_epilogue_kernel.
// OA / OB = int32_t, or uint16_t where a block is short enough for its counts (+ bias for a signed one) to fit 16 bits: a
// block of the R drivers is 26 843 loci, so IBS and its valid count (<= 2 x loci), N_Aa (<= loci) and the allele-sharing
// numerator (|D| <= loci, biased by 32 768) and denominator travel as 16 bits -- half the bytes again; the KING numerator
// (|x| <= 2 x loci) stays int32.
template <typename OA, typename OB>
__global__ __launch_bounds__(256) void tpg_pairwise_counts2_i32_kernel(const int32_t* __restrict__ acc,
                                                                       const int64_t* __restrict__ rowpad, int nst, int n,
                                                                       int which, int quirk, OA* __restrict__ oA, int biasA,
                                                                       OB* __restrict__ oB) {
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (ti > tj) return;
  __shared__ int sp[5][32][33];
  const int32_t* p = acc + (tpg_pw_unit_index(nst, ti / TA, tj) + rowpad[ti / TA]) * TPG_PW_TILE_INTS + ((ti % TA) * 16) * 64;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int idx = threadIdx.x + 256 * e, reg = idx >> 6, lane = idx & 63;
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31;
#pragma unroll
    for (int q = 0; q < 5; q++) sp[q][row][col] = p[q * TPG_PW_PLANE_INTS + reg * 64 + lane];
  }
  __syncthreads();
  auto emit = [&](int row, int col, int64_t idx, bool mirrored) {
    const int V = sp[0][row][col], D = sp[1][row][col], H = sp[2][row][col];
    const int Aij = mirrored ? sp[4][row][col] : sp[3][row][col], Aji = mirrored ? sp[3][row][col] : sp[4][row][col];
    if (which == 0) { oA[idx] = (OA)(V + D + H + biasA); oB[idx] = (OB)(2 * V); }
    else if (which == 1) { oA[idx] = (OA)(D - V + Aij + Aji + biasA); oB[idx] = (OB)Aij; }
    else { oA[idx] = (OA)(D + quirk + biasA); oB[idx] = (OB)V; }
  };
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int idx = threadIdx.x + 256 * e;
    {
      const int row = idx & 31, col = idx >> 5;
      const int gi = 32 * ti + row, gj = 32 * tj + col;
      if (gi < n && gj < n) emit(row, col, gi + (int64_t)gj * n, false);
    }
    if (ti != tj) {
      const int col = idx & 31, row = idx >> 5;
      const int gi = 32 * ti + row, gj = 32 * tj + col;
      if (gi < n && gj < n) emit(row, col, gj + (int64_t)gi * n, true);
    }
  }
}

// K += sums, K2 += sums: exact (integer-valued doubles), so the order of the blocks does not matter.  This is what an
// unmodified R driver pays PER BLOCK (38 times at 5 000 x 1 000 000), so it is built for that: the two matrices leave the
// device as int32 (2 x 100 MB instead of 2 x 200 MB) into pinned staging buffers the context keeps between calls (no
// 400 MB of fresh pages per block), and a team of threads adds the first to K while the second is still on its way.
// TPG_INCREMENT_TRACE=1: one line per call of an increment_* mirror with the milliseconds of its phases (stderr)
static bool increment_trace() {
  const char* e = getenv("TPG_INCREMENT_TRACE");
  return e && atoi(e) != 0;
}
static double inc_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e3; }

static int add_counts_to_caller(tpg_ctx* ctx, int which, const tpg_pairwise* pw, double* A, double* B) {
  const size_t nn = (size_t)pw->n * (size_t)pw->n;
  Resident* r = resident_of(ctx);
  const bool trace = increment_trace();
  double tr[6] = {inc_now(), 0, 0, 0, 0, 0};
  // every int32 entry is bounded by 2 x loci + quirk
  const bool fits = 2 * pw->loci + pw->as_pad_quirk < (1ll << 31) && pw->nranks == 1;
  const int NT = nn >= (1u << 20) ? (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1;
  auto team = [&](auto body) {
    std::vector<std::thread> th;
    for (int t = 1; t < NT; t++) th.emplace_back(body, t);
    body(0);
    for (auto& t : th) t.join();
  };
  if (!fits) {  // very long panels (or sharded accumulators): the double outputs of tpg_pairwise_counts
    std::vector<double> ta(nn), tb(nn);
    if (which == 0) TPG_TRY(tpg_pairwise_counts(ctx, pw, ta.data(), tb.data(), nullptr, nullptr, nullptr, nullptr));
    else if (which == 1) TPG_TRY(tpg_pairwise_counts(ctx, pw, nullptr, nullptr, ta.data(), tb.data(), nullptr, nullptr));
    else TPG_TRY(tpg_pairwise_counts(ctx, pw, nullptr, nullptr, nullptr, nullptr, ta.data(), tb.data()));
    team([&](int t) {
      for (size_t k = nn * (size_t)t / (size_t)NT; k < nn * (size_t)(t + 1) / (size_t)NT; k++) { A[k] += ta[k]; B[k] += tb[k]; }
    });
    return TPG_OK;
  }
  TPG_TRY(pw_need(pw, which == 0 ? TPG_PW_FOR_IBS_ALONE : which == 1 ? TPG_PW_FOR_KING : TPG_PW_FOR_AS, "increment"));
  // widths on the wire: 16 bits where the block's bounds allow it (see the kernel); TPG_INCREMENT_I32=1: always int32 (A/B)
  static const bool wide = getenv("TPG_INCREMENT_I32") && atoi(getenv("TPG_INCREMENT_I32")) != 0;
  const int64_t bound = 2 * pw->loci + pw->as_pad_quirk;  // of |IBS|, |valid|, |KING numerator|; N_Aa, |D|, V <= loci + quirk
  const bool a16 = !wide && (which == 0 ? bound <= 65535 : which == 2 ? pw->loci + pw->as_pad_quirk <= 32767 : false);
  const bool b16 = !wide && (which == 0 ? bound <= 65535 : pw->loci <= 65535);
  const int biasA = (a16 && which == 2) ? 32768 : 0;
  const size_t bytesA = nn * (a16 ? 2 : 4), bytesB = nn * (b16 ? 2 : 4);
  if (r->stage_ints * sizeof(int32_t) < bytesA + bytesB) {  // pinned staging the context keeps: as large as these widths need
    if (r->stage) (void)hipHostFree(r->stage);              // (pinning costs ~0.2 ms per MB: 100 MB less for an IBS block loop)
    r->stage = nullptr;
    r->stage_ints = 0;
    TPG_HIP(hipHostMalloc((void**)&r->stage, bytesA + bytesB, hipHostMallocDefault));
    r->stage_ints = (bytesA + bytesB) / sizeof(int32_t);
  }
  uint8_t* d_out = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_out, bytesA + bytesB));
  const unsigned nt = (unsigned)ceil_div(pw->n, 32);
  hipError_t e = hipSuccess;
  {
    ProfScope ps(ctx, "pairwise_counts_i32");
    const dim3 grid(nt, nt);
#define CNT2(TA_, TB_)                                                                                                        \
  hipLaunchKernelGGL((tpg_pairwise_counts2_i32_kernel<TA_, TB_>), grid, dim3(256), 0, ctx->stream, (const int32_t*)pw->acc,   \
                     (const int64_t*)pw->rowpad, (int)pw->nst, (int)pw->n, which, (int)pw->as_pad_quirk, (TA_*)d_out, biasA,  \
                     (TB_*)(d_out + bytesA))
    if (a16 && b16) CNT2(uint16_t, uint16_t);
    else if (b16) CNT2(int32_t, uint16_t);
    else if (a16) CNT2(uint16_t, int32_t);
    else CNT2(int32_t, int32_t);
#undef CNT2
    e = hipGetLastError();
  }
  uint8_t* stage = (uint8_t*)r->stage;
  // Each matrix leaves the device in PIECES (an event behind each), and the team adds a piece as soon as it is in the staging
  // buffer: the additions run beside the transfer instead of a whole matrix behind it.  Piece p < NP: of the first matrix,
  // p >= NP: of the second; boundaries on multiples of 16 elements (the vector step of host_addcounts.h).
  constexpr int NP = 4;
  const int npieces = nn >= (1u << 20) ? NP : 1;
  auto piece_lo = [&](int q) { return q >= npieces ? nn : (nn * (size_t)q / (size_t)npieces) & ~(size_t)15; };
  hipEvent_t ev[2 * NP] = {};
  for (int q = 0; q < 2 * npieces && e == hipSuccess; q++) e = hipEventCreateWithFlags(&ev[q], hipEventDisableTiming);
  for (int q = 0; q < 2 * npieces && e == hipSuccess; q++) {
    const bool second = q >= npieces;
    const size_t es = (second ? b16 : a16) ? 2 : 4, lo = piece_lo(q % npieces), hi = piece_lo(q % npieces + 1);
    const size_t off = (second ? bytesA : 0) + lo * es;
    e = hipMemcpyAsync(stage + off, d_out + off, (hi - lo) * es, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(ev[q], ctx->stream);
  }
  tpg_pfree(d_out);  // stream-ordered
  auto add = [&](int t, int q) {  // thread t's stripe of piece q
    const bool second = q >= npieces;
    const size_t lo = piece_lo(q % npieces), hi = piece_lo(q % npieces + 1);
    const size_t k0 = lo + (hi - lo) * (size_t)t / (size_t)NT, k1 = lo + (hi - lo) * (size_t)(t + 1) / (size_t)NT;
    double* dst = second ? B : A;
    const uint8_t* src = stage + (second ? bytesA : 0);
    if (second ? b16 : a16) tpg_add_counts_u16(dst + k0, (const uint16_t*)src + k0, k1 - k0, second ? 0 : biasA);
    else tpg_add_counts_i32(dst + k0, (const int32_t*)src + k0, k1 - k0);
  };
  // ONE team for both matrices (this runs once per block of the R loop).  arrived = pieces in the staging buffer (-1: a
  // transfer failed); thread 0 does the waiting, then adds its own stripe of what it waited for.
  //
  // The FIRST call on a pair of matrices finds their pages unmapped: the accumulators of the R drivers are MAP_SHARED file
  // mappings made just before the loop, and a write fault on one costs 2-25 us (the file system's page_mkwrite; 49 000 pages
  // per matrix at n = 5 000: 75-125 ms per matrix on the pool's overlay file system however many threads share it, but two
  // FILES in parallel take little more than one -- tools/host_rmw_probe.cpp).  So on a pair not seen in the previous call
  // half of the team starts with the second matrix: both files are being faulted in at any time.
  const bool first_touch = r->last_A != A || r->last_B != B;
  r->last_A = A;
  r->last_B = B;
  std::atomic<int> arrived{0};
  if (e != hipSuccess) arrived.store(-1);
  auto wait_for = [&](int need) {
    int sv;
    while ((sv = arrived.load(std::memory_order_acquire)) >= 0 && sv < need) std::this_thread::yield();
    return sv >= 0;
  };
  tr[1] = inc_now();
  team([&](int t) {
    if (t == 0) {
      if (arrived.load() < 0) return;
      if (first_touch) {  // (nobody waits for this thread's stripes, which are slow this time)
        hipError_t w = hipStreamSynchronize(ctx->stream);
        if (w != hipSuccess) { e = w; arrived.store(-1); return; }
        tr[2] = tr[3] = inc_now();
        arrived.store(2 * npieces, std::memory_order_release);
      } else {
        for (int q = 0; q < 2 * npieces; q++) {
          hipError_t w = hipEventSynchronize(ev[q]);
          if (w != hipSuccess) { e = w; arrived.store(-1); return; }
          if (q == 0) tr[2] = inc_now();
          arrived.store(q + 1, std::memory_order_release);
          add(0, q);
        }
        tr[3] = inc_now();
        return;
      }
    }
    const int start = first_touch && t >= (NT + 1) / 2 ? npieces : 0;
    for (int i = 0; i < 2 * npieces; i++) {
      const int q = (start + i) % (2 * npieces);
      if (!wait_for(first_touch ? 2 * npieces : q + 1)) return;
      add(t, q);
    }
  });
  tr[5] = inc_now();
  if (trace)
    fprintf(stderr, "[tpg increment] counts: enqueued %.2f | first piece arrives %.2f | thread 0 done %.2f | team joined %.2f ms (%d threads, "
                    "%d pieces%s)\n", tr[1] - tr[0], tr[2] - tr[1], tr[3] - tr[2], tr[5] - tr[3], NT, 2 * npieces, first_touch ? ", first touch" : "");
  for (int q = 0; q < 2 * NP; q++)
    if (ev[q]) (void)hipEventDestroy(ev[q]);
  if (e != hipSuccess) { tpg_set_error("increment: counts to the caller: %s", hipGetErrorString(e)); return TPG_EHIP; }
  return TPG_OK;
}

// the columns colInd names, in HBM: the covering column range of the FBM when the block is (nearly) contiguous, as the
// blocks of the R drivers are; the m columns gathered on the host when colInd is scattered (an LD-pruned subset ...)
static int upload_block_columns(tpg_ctx* ctx, const uint8_t* fbm_bytes, int64_t nrow, int64_t ncol, const int32_t* colInd1,
                                int64_t m, tpg_fbm** f, std::vector<int32_t>& cols) {
  int32_t lo = colInd1[0], hi = colInd1[0];
  for (int64_t j = 0; j < m; j++) {
    TPG_REQUIRE(colInd1[j] >= 1 && colInd1[j] <= ncol, TPG_EINVAL, "colInd[%lld] = %d out of [1,%lld]", (long long)j,
                colInd1[j], (long long)ncol);
    lo = std::min(lo, colInd1[j]);
    hi = std::max(hi, colInd1[j]);
  }
  const int64_t span = (int64_t)hi - lo + 1;
  cols.resize((size_t)m);
  if (span <= 2 * m + 64) {
    for (int64_t j = 0; j < m; j++) cols[(size_t)j] = colInd1[j] - (lo - 1);
    // (the increment_* functions compare raw bytes: one table, so the block goes up as 2 bits per genotype where its bytes allow it)
    const double* raw = nullptr;
    return tpg_fbm_from_host_for_table(ctx, fbm_bytes + (size_t)(lo - 1) * (size_t)nrow, nrow, span, nullptr, f, &raw);
  }
  std::vector<uint8_t> stage((size_t)nrow * (size_t)m);
  for (int64_t j = 0; j < m; j++) {
    memcpy(stage.data() + (size_t)j * (size_t)nrow, fbm_bytes + (size_t)(colInd1[j] - 1) * (size_t)nrow, (size_t)nrow);
    cols[(size_t)j] = (int32_t)(j + 1);
  }
  const double* raw = nullptr;
  return tpg_fbm_from_host_for_table(ctx, stage.data(), nrow, m, nullptr, f, &raw);  // waited for: the staging buffer may go
}

static int increment_common(tpg_ctx* ctx, int which, double* A, double* B, const uint8_t* fbm_bytes, int64_t nrow,
                            int64_t ncol, const int32_t* rowInd1, int64_t n, const int32_t* colInd1, int64_t m) {
  TPG_REQUIRE(ctx && A && B && fbm_bytes && rowInd1 && colInd1, TPG_EINVAL, "null argument");
  TPG_REQUIRE(n > 0 && m > 0 && nrow > 0 && ncol > 0, TPG_EINVAL, "empty block");
  Resident* r = resident_of(ctx);
  ResidentAcc* acc = nullptr;
  for (auto& e : r->accs)
    if (e.A == A && e.B == B) acc = &e;
  tpg_pairwise* pw = nullptr;
  const bool trace = increment_trace();
  const double t_in = trace ? inc_now() : 0;
  if (acc) {
    TPG_REQUIRE(acc->which == which, TPG_EINVAL, "these accumulators are pending for another increment_* function; flush first");
    TPG_REQUIRE((int64_t)acc->rows.size() == n && memcmp(acc->rows.data(), rowInd1, sizeof(int32_t) * (size_t)n) == 0,
                TPG_EINVAL, "rowInd changed between blocks that accumulate into the same matrices; flush first");
    pw = acc->pw;
  } else if (r->defer) {
    TPG_TRY(tpg_pairwise_create(ctx, n, nullptr, &pw));
    r->accs.push_back(ResidentAcc{which, A, B, std::vector<int32_t>(rowInd1, rowInd1 + n), pw});
  } else {
    if (r->spare && r->spare->n != n) { tpg_pairwise_free(r->spare); r->spare = nullptr; }
    if (r->spare) TPG_TRY(tpg_pairwise_zero(ctx, r->spare));
    else TPG_TRY(tpg_pairwise_create(ctx, n, nullptr, &r->spare));
    pw = r->spare;
  }
  tpg_fbm* f = nullptr;
  tpg_view* v = nullptr;
  std::vector<int32_t> cols;
  const double t0 = trace ? inc_now() : 0;
  int rc = upload_block_columns(ctx, fbm_bytes, nrow, ncol, colInd1, m, &f, cols);
  const double t1 = trace ? inc_now() : 0;
  if (rc == TPG_OK) rc = tpg_view_create(ctx, f, rowInd1, n, cols.data(), m, nullptr /* raw bytes, src/snp_ibs.cpp:47-54 */, &v);
  // the products this entry point's two matrices are made of (src/snp_ibs.cpp:67-72, src/snp_king.cpp:70-72, src/snp_as.cpp:64-65)
  const int products = which == 0 ? TPG_PW_FOR_IBS_ALONE : which == 1 ? TPG_PW_FOR_KING : TPG_PW_FOR_AS;
  if (rc == TPG_OK) rc = tpg_pairwise_accumulate_products(ctx, pw, v, 0, -1, products);
  if (trace) fprintf(stderr, "[tpg increment] %lld loci: columns up %.2f | view + products enqueued %.2f ms\n", (long long)m, t1 - t0, inc_now() - t1);
  if (rc == TPG_OK && pw == r->spare) rc = add_counts_to_caller(ctx, which, pw, A, B);  // immediate: as the reference
  const double t_add = trace ? inc_now() : 0;
  tpg_view_free(v);  // stream-ordered: the blocks return to this context's pool
  tpg_fbm_free(f);
  if (trace) fprintf(stderr, "[tpg increment] accumulators ready %.2f | frees %.2f | whole call %.2f ms\n", t0 - t_in, inc_now() - t_add, inc_now() - t_in);
  return rc;
}

extern "C" int tpg_increment_flush(tpg_ctx* ctx) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx, TPG_EINVAL, "null ctx");
  Resident* r = (Resident*)ctx->resident;
  if (!r) return TPG_OK;
  int rc = TPG_OK;
  for (auto& a : r->accs) {
    if (rc == TPG_OK) rc = add_counts_to_caller(ctx, a.which, a.pw, a.A, a.B);
    tpg_pairwise_free(a.pw);
  }
  r->accs.clear();
  return rc;
}

// on != 0: the increment_* mirrors keep their sums in HBM until tpg_increment_flush; 0 (the default): every call
// increments the caller's matrices before it returns.  Switching it off flushes what is pending.
extern "C" int tpg_increment_defer(tpg_ctx* ctx, int on) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx, TPG_EINVAL, "null ctx");
  Resident* r = resident_of(ctx);
  if (!on && !r->accs.empty()) TPG_TRY(tpg_increment_flush(ctx));
  r->defer = on != 0;
  return TPG_OK;
}

// releases the device memory the increment_* mirrors hold between calls (an error while increments are pending).  No
// copy of a caller's FBM is ever kept, so there is nothing to invalidate when its bytes change.
extern "C" int tpg_resident_drop(tpg_ctx* ctx) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx, TPG_EINVAL, "null ctx");
  Resident* r = (Resident*)ctx->resident;
  TPG_REQUIRE(!r || r->accs.empty(), TPG_EINVAL, "increments are pending: call tpg_increment_flush first");
  const bool defer = r && r->defer;
  tpg_resident_release(ctx);
  if (defer) resident_of(ctx)->defer = true;
  return TPG_OK;
}

// quirk Q1 for the literal mirror: the shim calls this for a block whose scratch matrices are one column wider than
// the block (src/snp_as.cpp:57-63) when the emulation is wanted: +1 on every element of the numerator K (n x n) --
// noted on the pending device accumulators, or added to K right away when nothing is pending for it
extern "C" int tpg_increment_as_note_narrow_block(tpg_ctx* ctx, double* K, int64_t n) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && K && n > 0, TPG_EINVAL, "bad argument");
  Resident* r = (Resident*)ctx->resident;
  if (r)
    for (auto& a : r->accs)
      if (a.A == K) {
        TPG_REQUIRE(a.which == 2 && a.pw->n == n, TPG_EINVAL, "K is pending for another increment_* function or another n");
        a.pw->as_pad_quirk += 1;
        return TPG_OK;
      }
  for (size_t k = 0; k < (size_t)n * (size_t)n; k++) K[k] += 1.0;
  return TPG_OK;
}

extern "C" int tpg_increment_ibs_counts(tpg_ctx* ctx, double* K, double* K2, const uint8_t* fbm_bytes, int64_t nrow,
                                        int64_t ncol, const int32_t* rowInd1, int64_t n, const int32_t* colInd1,
                                        int64_t m) {
  TpgEnter _enter(ctx);
  return increment_common(ctx, 0, K, K2, fbm_bytes, nrow, ncol, rowInd1, n, colInd1, m);
}
extern "C" int tpg_increment_king_numerator(tpg_ctx* ctx, double* K, double* N_Aa_i, const uint8_t* fbm_bytes,
                                            int64_t nrow, int64_t ncol, const int32_t* rowInd1, int64_t n,
                                            const int32_t* colInd1, int64_t m) {
  TpgEnter _enter(ctx);
  return increment_common(ctx, 1, K, N_Aa_i, fbm_bytes, nrow, ncol, rowInd1, n, colInd1, m);
}
extern "C" int tpg_increment_as_counts(tpg_ctx* ctx, double* K, double* K2, const uint8_t* fbm_bytes, int64_t nrow,
                                       int64_t ncol, const int32_t* rowInd1, int64_t n, const int32_t* colInd1,
                                       int64_t m) {
  TpgEnter _enter(ctx);
  return increment_common(ctx, 2, K, K2, fbm_bytes, nrow, ncol, rowInd1, n, colInd1, m);
}
