// pack.hip -- FBM bytes -> 2-bit fragment layouts T and L (common.h), unpack, synthetic panel.
//
// pack: one workgroup per 128 individuals x 128 loci tile.  Generic kernel (row subsets, .bed sources, unaligned
// columns): the tile's bytes are gathered through (rowInd, colInd), decoded by the 256-entry byte -> 2-bit table
// held in LDS, kept as one code byte per genotype in LDS, then emitted twice: four 1-KiB T blocks (contraction
// over loci) and four 1-KiB L blocks (contraction over individuals).  The common case (all rows of a byte FBM)
// takes tpg_pack_fast_kernel below.  HBM-bound: reads n*m bytes, writes n*m/2 bytes.  Replaces the per-block
// byte decode loops of the reference (src/snp_ibs.cpp:45-55, src/snp_king.cpp:45-58, src/snp_as.cpp:44-53 and
// the SubBMCode256Acc accessor in src/alt_freq_dip_pseudo_cpp.cpp:15-16).
#include <type_traits>

#include "common.h"
#include "devfrag.h"
#include "synth_common.h"

#define TILE 128
#define LDS_STRIDE 132  // bytes per locus row in LDS (128 + 4 pad)

// the FBM byte bigsnpr writes for a .bed 2-bit code (getCode(), recalled): 00,01,10,11 -> 2,3,1,0
__device__ __forceinline__ uint8_t tpg_bed_byte(uint8_t packed, int slot) {
  return (uint8_t)((0x00010302u >> (8 * ((packed >> (2 * slot)) & 3))) & 0xFF);
}

__global__ __launch_bounds__(256) void tpg_pack_kernel(const uint8_t* __restrict__ fbm, int64_t nrow, int64_t bed_bpl,
                                                       const int32_t* __restrict__ rows,
                                                       const int32_t* __restrict__ cols, uint8_t* lut_and_flag,
                                                       int64_t n, int64_t m, int64_t Q, int64_t KG,
                                                       uint32_t* __restrict__ T, uint32_t* __restrict__ L) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[256 + TILE * LDS_STRIDE];
  uint8_t* lut = smem;
  uint8_t* codes = smem + 256;  // codes[locus][individual]
  const int tid = threadIdx.x;
  const int64_t bj = blockIdx.x;  // locus group (kg)
  const int64_t bi = blockIdx.y;  // individual chunk (q)
  lut[tid] = lut_and_flag[tid];
  __syncthreads();

  bool bad = false;
  const bool bed_fast = (bed_bpl > 0) && (rows == nullptr) && (bi * TILE + TILE <= n);
  if (bed_fast) {
    // .bed: 128 individuals of one SNP = 32 contiguous bytes; 8 SNPs per pass, byte index on the lanes
    const int b = tid & 31;
#pragma unroll 4
    for (int l = tid >> 5; l < TILE; l += 8) {
      const int64_t j = bj * TILE + l;
      uint8_t pk = 0x55;  // four "missing" codes (01)
      if (j < m) {
        const int64_t src_col = cols ? (int64_t)cols[j] - 1 : j;
        pk = fbm[bi * (TILE / 4) + b + src_col * bed_bpl];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        uint8_t c = j < m ? lut[tpg_bed_byte(pk, q)] : (uint8_t)3;
        if (c == 0xFF) { bad = true; c = 3; }
        codes[l * LDS_STRIDE + 4 * b + q] = c;
      }
    }
  } else {
    const int ii = tid & 127;
    const int64_t i = bi * TILE + ii;
    int64_t src_row = -1;
    if (i < n) src_row = rows ? (int64_t)rows[i] - 1 : i;
    for (int l = tid >> 7; l < TILE; l += 2) {
      const int64_t j = bj * TILE + l;
      uint8_t c = 3;
      if (src_row >= 0 && j < m) {
        const int64_t src_col = cols ? (int64_t)cols[j] - 1 : j;
        const uint8_t raw = bed_bpl ? tpg_bed_byte(fbm[(src_row >> 2) + src_col * bed_bpl], (int)(src_row & 3))
                                    : fbm[src_row + src_col * nrow];
        c = lut[raw];
        if (c == 0xFF) { bad = true; c = 3; }
      }
      codes[l * LDS_STRIDE + ii] = c;
    }
  }
  if (bad) atomicOr((unsigned int*)(lut_and_flag + 256), 1u);
  __syncthreads();

#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int idx = tid + 256 * it;
    const int s = idx & 3, lane = (idx >> 2) & 63, tl = idx >> 8;
    const int r = lane & 31, h = lane >> 5;
    // T: individual 32 tl + r, loci 32 s + 16 h + e
    {
      const int ind = 32 * tl + r;
      uint32_t w = 0;
#pragma unroll
      for (int e = 0; e < 16; e++) w |= (uint32_t)codes[(32 * s + 16 * h + e) * LDS_STRIDE + ind] << tpg_elem_shift(e);
      const int64_t rt = bi * 4 + tl;
      T[((rt * KG + bj) * 64 + lane) * 4 + s] = w;
    }
    // L: locus 32 tl + r, individuals 32 s + 16 h + e
    {
      const int loc = 32 * tl + r;
      uint32_t w = 0;
#pragma unroll
      for (int e = 0; e < 16; e++) w |= (uint32_t)codes[loc * LDS_STRIDE + 32 * s + 16 * h + e] << tpg_elem_shift(e);
      const int64_t lt = bj * 4 + tl;
      L[((lt * Q + bi) * 64 + lane) * 4 + s] = w;
    }
  }
}

// Fast path of the pack: FBM bytes (not .bed), all rows in file order, columns 8-byte aligned.
//   phase 1  thread = (locus, 16 individuals): two 8-byte loads, byte -> code through v_perm_b32 with the first
//            eight table entries as the 8-byte pool (one instruction per four genotypes; a dword holding a byte
//            >= 8 takes the LDS table instead), the L word is the four code dwords shifted and OR-ed together
//            (that is what the element order of the layout is for) and goes straight to HBM; the code bytes go to
//            LDS as one 16-byte write;
//   phase 2  thread = (16 loci, 4 individuals): 16 dword reads, four 4x4 byte transposes (8 v_perm each), four T
//            words.  Rows are rotated by 32 bytes per 16 loci so that both phases are free of bank conflicts.
// ~20 VALU and ~6 LDS instructions per output word pair instead of ~100 and ~50: the kernel sits on the HBM
// roofline (reads n*m bytes, writes n*m/2).
// NV = 2: the same bytes decoded through TWO code tables into two views from one read of the FBM (the raw view of the
// pairwise statistics and the imputed view of the PCA: R/gt_has_imputed.R:101-106 switches between exactly these two
// tables on one FBM) -- 10 GB of traffic instead of 15 GB at 5 000 x 1 000 000.
// T0 / T1 may be NULL (the view gets no T layout: tpg_view_need_T makes it from L if it is ever wanted); T4 (may be
// NULL) = the FP4 operand layout of view 0 for the pairwise kernel (pairwise.hip), written instead of being expanded
// from T later: every T word goes out as the two T4 words of the same lane.
#ifndef PACK_NSUB
#define PACK_NSUB 2  // individual chunks per workgroup, all their loads issued up front
#endif
__device__ __forceinline__ int64_t tpg_pack_uniform64(int64_t x) {  // a wave-uniform value the compiler keeps in SGPRs
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)x >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
typedef uint32_t pk_u4 __attribute__((ext_vector_type(4)));
typedef pk_u4 pk_u4a8 __attribute__((aligned(8)));
// BED (round 5): the store is a PLINK .bed payload (tpg_fbm::bed_bpl bytes per SNP, 4 genotypes per byte, `nrow` = bed_bpl).
// Only the front of phase 1 differs: a thread's 16 individuals of a locus are ONE dword (any byte address: a SNP is
// ceil(n / 4) bytes), byte q of it four individuals; two shift-or-mask steps spread its four 2-bit fields over four
// bytes, and one v_perm_b32 with the four composed entries table[bigsnpr's byte of .bed code b] turns them into codes.  The
// generic kernel did this route until now: two passes + a T -> T4 expansion + the counts kernel over L, 5.5 ms more per step
// at 5 000 x 1 000 000 than the byte FBM.
template <int NV, bool BED = false>
__global__ __launch_bounds__(256) void tpg_pack_fast_kernel(const uint8_t* __restrict__ fbm, int64_t nrow,
                                                            const int32_t* __restrict__ cols, uint8_t* lut_and_flag,
                                                            int64_t n, int64_t m, int64_t Q, int64_t KG,
                                                            uint32_t* __restrict__ T0, uint32_t* __restrict__ L0,
                                                            uint32_t* __restrict__ T1, uint32_t* __restrict__ L1,
                                                            uint32_t* __restrict__ T4, int xcd_map,
                                                            uint32_t* __restrict__ P0, uint32_t* __restrict__ P1) {
  // dynamic: the tables + one TILE x TILE array of code bytes per view that gets a T or T4 layout (tpg_pack_lds_bytes).  A view
  // with L only -- the imputed view of the bench's pair -- needs no pass through LDS at all, and with 17 instead of 33 KiB a CU
  // holds six workgroups instead of four: the kernel is bound by the bytes it has in flight, not by HBM or the VALU
  // (rocprofv3 --pmc: the waves issue VALU 11 % of their cycles; FETCH_SIZE at the algorithmic 5.0 GB with the XCD map)
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  bool need_[NV];
  int slot_[NV];
#pragma unroll
  for (int vw = 0; vw < NV; vw++) {
    need_[vw] = vw ? T1 != nullptr : (T0 != nullptr || T4 != nullptr);
    slot_[vw] = vw ? (need_[0] ? 1 : 0) : 0;
  }
  // lut_and_flag: NV tables of 256 bytes, each followed by its 16-byte flag area
  uint8_t* codes_all = smem + NV * (256 + 16);  // codes[view][locus][(individual + 32 * (locus >> 4)) & 127]
  const int tid = threadIdx.x;
  // 1-D grid, individual chunk fastest: neighbouring workgroups read neighbouring pieces of the same 128 columns
  // (columns are only 8-byte aligned, so the pieces share cache lines and DRAM pages).  A workgroup does NSUB
  // individual chunks one after the other with all their loads issued up front (16 x 8 B in flight per thread).
  constexpr int NSUB = PACK_NSUB;
  const int64_t QB = (Q + NSUB - 1) / NSUB;
  int64_t bj = blockIdx.x / QB;            // locus group (kg)
  int64_t bi0 = (blockIdx.x % QB) * NSUB;  // first individual chunk (q)
  if (xcd_map) {  // workgroup b runs on XCD b % 8: the chunks of one locus group stay on one XCD, whose L2 then holds the lines two of them share
    const int64_t k = blockIdx.x >> 3;
    bj = (k / QB) * 8 + (blockIdx.x & 7);
    bi0 = (k % QB) * NSUB;
    if (bj >= KG) return;
  }
#pragma unroll
  for (int vw = 0; vw < NV; vw++) smem[vw * (256 + 16) + tid] = lut_and_flag[vw * (256 + 16) + tid];
  __syncthreads();
  uint32_t lo_[NV], hi_[NV];
  bool bad_[NV];
#pragma unroll
  for (int vw = 0; vw < NV; vw++) {
    lo_[vw] = *reinterpret_cast<const uint32_t*>(smem + vw * (256 + 16));
    hi_[vw] = *reinterpret_cast<const uint32_t*>(smem + vw * (256 + 16) + 4);
    bad_[vw] = false;
  }
  if constexpr (BED) {  // .bed code b -> the byte bigsnpr would have stored (00, 01, 10, 11 -> 2, 3, 1, 0) -> this view's code
#pragma unroll
    for (int vw = 0; vw < NV; vw++) {
      const uint8_t* lut = smem + vw * (256 + 16);
      lo_[vw] = (uint32_t)lut[2] | ((uint32_t)lut[3] << 8) | ((uint32_t)lut[1] << 16) | ((uint32_t)lut[0] << 24);
      hi_[vw] = lo_[vw];
    }
  }
  auto conv = [&](uint32_t w, int vw) -> uint32_t {
    const uint8_t* lut = smem + vw * (256 + 16);
    const uint32_t lo = lo_[vw], hi = hi_[vw];
    bool& bad = bad_[vw];
    uint32_t c;
    if ((w & 0xF8F8F8F8u) == 0) c = __builtin_amdgcn_perm(hi, lo, w);
    else
      c = (uint32_t)lut[w & 0xFF] | ((uint32_t)lut[(w >> 8) & 0xFF] << 8) | ((uint32_t)lut[(w >> 16) & 0xFF] << 16) |
          ((uint32_t)lut[w >> 24] << 24);
    if (c & 0x80808080u) {  // 0xFF = value the code table does not map: flag it, treat as missing
      bad = true;
      const uint32_t msk = ((c & 0x80808080u) >> 7) * 0xFFu;
      c = (c & ~msk) | (0x03030303u & msk);
    }
    return c;
  };
  const int c16 = tid & 7;
  uint2 va[NSUB][4], vb[NSUB][4];
#pragma unroll
  for (int sub = 0; sub < NSUB; sub++) {
    const int64_t i0 = (bi0 + sub) * TILE + 16 * c16;
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int64_t j = bj * TILE + (tid >> 3) + 32 * it;
      va[sub][it] = make_uint2(0, 0);
      vb[sub][it] = make_uint2(0, 0);
      if constexpr (BED) {
        if (j < m && bi0 + sub < Q && i0 < n) {
          const int64_t src_col = cols ? (int64_t)cols[j] - 1 : j;
          const uint8_t* p = fbm + (i0 >> 2) + src_col * nrow;  // (nrow = bytes per SNP)
          typedef uint32_t pk_u1a1 __attribute__((aligned(1)));
          if ((i0 >> 2) + 4 <= nrow) {
            va[sub][it].x = *reinterpret_cast<const pk_u1a1*>(p);
          } else {  // the last bytes of a SNP whose individuals do not fill 16: never past the record
            uint32_t w = 0;
            for (int b = 0; (i0 >> 2) + b < nrow; b++) w |= (uint32_t)p[b] << (8 * b);
            va[sub][it].x = w;
          }
        }
      } else if (j < m && bi0 + sub < Q) {
        const int64_t src_col = cols ? (int64_t)cols[j] - 1 : j;
        const uint8_t* p = fbm + i0 + src_col * nrow;
        // (columns are 8-byte aligned only; the hardware takes a 16-byte load at any dword address: half the load instructions)
        if (i0 + 16 <= n) {
          const pk_u4 t = *reinterpret_cast<const pk_u4a8*>(p);
          va[sub][it] = make_uint2(t[0], t[1]);
          vb[sub][it] = make_uint2(t[2], t[3]);
        } else if (i0 + 8 <= n) {
          va[sub][it] = *reinterpret_cast<const uint2*>(p);
        }
      }
    }
  }
  // One decision per WAVE instead of two divergent tests per dword (64 conv() calls per thread: ~700 v_cmp / s_and_saveexec /
  // s_cbranch_execz / s_or_b64 around 224 v_perm_b32): if no byte this wave loaded is >= 8 -- every genotype panel --
  // every dword takes the v_perm_b32 and the "code the table does not map" flag is looked at once at the end (a view with
  // that flag set is refused by the caller, so what its words hold does not matter).
  uint32_t anyw = 0;
#pragma unroll
  for (int sub = 0; sub < NSUB; sub++)
#pragma unroll
    for (int it = 0; it < 4; it++) anyw |= va[sub][it].x | va[sub][it].y | vb[sub][it].x | vb[sub][it].y;
  const bool fastw = BED || __builtin_amdgcn_ballot_w64((anyw & 0xF8F8F8F8u) != 0) == 0;  // wave-uniform
  uint32_t cacc[NV];
#pragma unroll
  for (int vw = 0; vw < NV; vw++) cacc[vw] = 0;
  // genotype counts of this thread's 16 (then, over the chunks, 32) individuals per locus and view, three 10-bit fields:
  // codes with bit 0 set, with bit 1 set, with both (tpg_view::lc_part)
  uint32_t cnt3[NV][4];
#pragma unroll
  for (int vw = 0; vw < NV; vw++)
#pragma unroll
    for (int it = 0; it < 4; it++) cnt3[vw][it] = 0;
  const bool cnt_on = P0 != nullptr;
#pragma unroll
  for (int sub = 0; sub < NSUB; sub++) {
  const int64_t bi = bi0 + sub;
  if (bi >= Q) break;
  if (sub) __syncthreads();  // the previous chunk's readers are done with `codes`
  // FULL (wave-uniform, with FAST): every thread of the wave is inside the view with all its 16 individuals -- no selects, and
  // every address is a base that was made once (uniform part in SGPRs, the thread's part a 32-bit offset) + an immediate:
  // written the plain way, the L store alone cost five 64-bit VALU operations per word.
  auto phase1 = [&](auto fastc, auto fullc) {
    constexpr bool FAST = decltype(fastc)::value, FULL = decltype(fullc)::value;
    const int64_t i0 = bi * TILE + 16 * c16;
    const int l0 = tid >> 3;  // locus l = l0 + 32 it, so l >> 5 = it, l & 31 = l0, l >> 4 = (l0 >> 4) + 2 it
    const uint32_t rot0 = (uint32_t)(16 * c16 + 32 * (l0 >> 4)) & 127u;
    const uint32_t cb[2] = {(uint32_t)l0 * TILE + rot0, (uint32_t)l0 * TILE + (rot0 ^ 64u)};  // `codes` offset of it even / odd
    const uint32_t loff = (uint32_t)(l0 + 32 * (c16 & 1)) * 16u + (uint32_t)(c16 >> 1) * 4u;  // inside a 1-KiB L block
    typedef __attribute__((address_space(1))) char gchar;
    typedef __attribute__((address_space(1))) uint32_t gu32;
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int l = l0 + 32 * it;
      const bool inside = bj * TILE + l < m;
#pragma unroll
      for (int vw = 0; vw < NV; vw++) {
        uint8_t* codes = codes_all + slot_[vw] * TILE * TILE;
        uint32_t c[4] = {0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u};
        if constexpr (FAST && BED) {
          const uint32_t wb = va[sub][it].x;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            // the four 2-bit fields of byte q, one per byte: individual 4 q + e in byte e (two shift-or-mask steps whose terms do
            // not overlap; a multiplication by 0x41041 would carry between them)
            const uint32_t Bq = (wb >> (8 * q)) & 0xFFu, xq = (Bq | (Bq << 12)) & 0x000F000Fu;
            const uint32_t sel = (xq | (xq << 6)) & 0x03030303u;
            uint32_t cq = __builtin_amdgcn_perm(hi_[vw], lo_[vw], sel);
            if constexpr (!FULL) {  // individuals past n (the padding bits of the last byte, or no byte at all), loci past m
              uint32_t keep = 0;
#pragma unroll
              for (int e = 0; e < 4; e++) keep |= (inside && i0 + 4 * q + e < n) ? (0xFFu << (8 * e)) : 0u;
              cq = (cq & keep) | (0x03030303u & ~keep);
            }
            c[q] = cq;
          }
          cacc[vw] |= (c[0] | c[1]) | (c[2] | c[3]);
        } else if constexpr (FAST) {
          const uint32_t w[4] = {va[sub][it].x, va[sub][it].y, vb[sub][it].x, vb[sub][it].y};
          const bool h0 = inside && i0 + 8 <= n, h1 = inside && i0 + 16 <= n;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint32_t cq = __builtin_amdgcn_perm(hi_[vw], lo_[vw], w[q]);
            c[q] = FULL || (q < 2 ? h0 : h1) ? cq : 0x03030303u;
          }
          cacc[vw] |= (c[0] | c[1]) | (c[2] | c[3]);
        } else {
          if (inside && i0 + 8 <= n) { c[0] = conv(va[sub][it].x, vw); c[1] = conv(va[sub][it].y, vw); }
          if (inside && i0 + 16 <= n) { c[2] = conv(vb[sub][it].x, vw); c[3] = conv(vb[sub][it].y, vw); }
        }
        if (need_[vw]) *reinterpret_cast<uint4*>(codes + cb[it & 1] + it * 32 * TILE) = make_uint4(c[0], c[1], c[2], c[3]);
        gchar* Lb = (gchar*)tpg_pack_uniform64((int64_t)((vw ? L1 : L0) + ((bj * 4 + it) * Q + bi) * 256));  // L block (lt = 4 bj + it, bi)
        const uint32_t lw = c[0] | (c[1] << 2) | (c[2] << 4) | (c[3] << 6);
        *(gu32*)(Lb + loff) = lw;
        if (cnt_on) {  // (the padding code 3 of individuals past n, loci past m, is not a genotype: masked out)
          uint32_t wc = lw;
          if constexpr (!FULL && BED) {  // individual i0 + 4 k + b sits at bits 8 b + 2 k of the L word
            uint32_t vm = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
              for (int b = 0; b < 4; b++) vm |= (inside && i0 + 4 * k + b < n) ? (3u << (8 * b + 2 * k)) : 0u;
            wc = lw & vm;
          } else if constexpr (!FULL) wc = (inside && i0 + 16 <= n) ? lw : (inside && i0 + 8 <= n) ? (lw & 0x0F0F0F0Fu) : 0u;
          cnt3[vw][it] += (uint32_t)__popc(wc & 0x55555555u) | ((uint32_t)__popc(wc & 0xAAAAAAAAu) << 10) |
                          ((uint32_t)__popc(wc & (wc >> 1) & 0x55555555u) << 20);
        }
      }
    }
  };
  // (wave-uniform) all 64 threads x 4 loci of this wave inside the view with 16 individuals each
  const bool fullw = __builtin_amdgcn_ballot_w64(!(bj * TILE + (tid >> 3) + 96 < m && bi * TILE + 16 * c16 + 16 <= n)) == 0;
  if (fastw && fullw) phase1(std::true_type{}, std::true_type{});
  else if (fastw) phase1(std::true_type{}, std::false_type{});
  else phase1(std::false_type{}, std::false_type{});
  __syncthreads();
  {
    const int wv = tid >> 6, t = tid & 63;
    const int g = t >> 3, iqq = t & 7;  // loci 16 g .. 16 g + 15, individuals 32 wv + 4 iqq .. + 3
#pragma unroll
    for (int vw = 0; vw < NV; vw++) {
    const uint8_t* codes = codes_all + slot_[vw] * TILE * TILE;
    uint32_t* T = vw ? T1 : T0;
    uint32_t* T4v = vw ? nullptr : T4;
    if (!T && !T4v) continue;
    uint32_t d[16];
#pragma unroll
    for (int e = 0; e < 16; e++)
      d[e] = *reinterpret_cast<const uint32_t*>(codes + (16 * g + e) * TILE + ((32 * wv + 4 * iqq + 32 * g) & 127));
    const int64_t rt = bi * 4 + wv;
    typedef __attribute__((address_space(1))) char gchar2;
    if (T) {
      uint32_t W[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t x0 = d[4 * k], x1 = d[4 * k + 1], x2 = d[4 * k + 2], x3 = d[4 * k + 3];
        const uint32_t t0 = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t1 = __builtin_amdgcn_perm(x1, x0, 0x07030602u);
        const uint32_t t2 = __builtin_amdgcn_perm(x3, x2, 0x05010400u), t3 = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
        W[0] |= __builtin_amdgcn_perm(t2, t0, 0x05040100u) << (2 * k);
        W[1] |= __builtin_amdgcn_perm(t2, t0, 0x07060302u) << (2 * k);
        W[2] |= __builtin_amdgcn_perm(t3, t1, 0x05040100u) << (2 * k);
        W[3] |= __builtin_amdgcn_perm(t3, t1, 0x07060302u) << (2 * k);
      }
      uint32_t* dst = T + ((rt * KG + bj) * 64 + 4 * iqq + 32 * (g & 1)) * 4 + (g >> 1);
#pragma unroll
      for (int b = 0; b < 4; b++) dst[b * 4] = W[b];
      if (T4v) {  // T word s = g >> 1 of lane l -> words 2 (s & 1), 2 (s & 1) + 1 of lane l in T4 block 2 kg + (s >> 1)
        const int sT = g >> 1;
        uint32_t* dst4 = T4v + (((rt * KG + bj) * 2 + (sT >> 1)) * 64 + 4 * iqq + 32 * (g & 1)) * 4 + 2 * (sT & 1);
#pragma unroll
        for (int b = 0; b < 4; b++) {
          uint32_t lo, hi;
          tpg_t4_words(W[b], lo, hi);
          *reinterpret_cast<uint2*>(dst4 + b * 4) = make_uint2(lo, hi);
        }
      }
    } else {
      // T4 only (the bench's raw view): the transposed code bytes go through the nibble table as they are -- the T word they
      // would be OR-ed into is never made (16 v_perm_b32 + 8 v_lshl_or_b32 where the detour took 16 + 52)
      uint32_t N[4][4];  // [k][b]
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t x0 = d[4 * k], x1 = d[4 * k + 1], x2 = d[4 * k + 2], x3 = d[4 * k + 3];
        const uint32_t t0 = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t1 = __builtin_amdgcn_perm(x1, x0, 0x07030602u);
        const uint32_t t2 = __builtin_amdgcn_perm(x3, x2, 0x05010400u), t3 = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
        N[k][0] = (uint32_t)tpg_lut(TPG_NIB_LUT, __builtin_amdgcn_perm(t2, t0, 0x05040100u));
        N[k][1] = (uint32_t)tpg_lut(TPG_NIB_LUT, __builtin_amdgcn_perm(t2, t0, 0x07060302u));
        N[k][2] = (uint32_t)tpg_lut(TPG_NIB_LUT, __builtin_amdgcn_perm(t3, t1, 0x05040100u));
        N[k][3] = (uint32_t)tpg_lut(TPG_NIB_LUT, __builtin_amdgcn_perm(t3, t1, 0x07060302u));
      }
      const int sT = g >> 1;
      gchar2* B4 = (gchar2*)tpg_pack_uniform64((int64_t)(T4v + ((rt * KG + bj) * 2) * 256));  // the two T4 blocks of (rt, kg)
      const uint32_t o4 = (uint32_t)(sT >> 1) * 1024u + (uint32_t)(4 * iqq + 32 * (g & 1)) * 16u + (uint32_t)(sT & 1) * 8u;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        typedef uint32_t pk_u2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(1))) pk_u2 gu2;
        *(gu2*)(B4 + o4 + b * 16) = pk_u2{N[0][b] | (N[1][b] << 4), N[2][b] | (N[3][b] << 4)};
      }
    }
    }  // view
  }
  }  // sub
#pragma unroll
  for (int vw = 0; vw < NV; vw++)
    if (bad_[vw] || (cacc[vw] & 0x80808080u)) atomicOr((unsigned int*)(lut_and_flag + vw * (256 + 16) + 256), 1u);
  if (cnt_on) {  // the eight threads of a locus (16 individuals each) are eight neighbouring lanes
    const int64_t Mp = KG * TILE;
#pragma unroll
    for (int vw = 0; vw < NV; vw++) {
      uint32_t* P = vw ? P1 : P0;
#pragma unroll
      for (int it = 0; it < 4; it++) {
        uint32_t pk = cnt3[vw][it];
        pk += (uint32_t)__shfl_xor((int)pk, 1);
        pk += (uint32_t)__shfl_xor((int)pk, 2);
        pk += (uint32_t)__shfl_xor((int)pk, 4);
        if (c16 == 0) P[(bi0 / NSUB) * Mp + bj * TILE + (tid >> 3) + 32 * it] = pk;
      }
    }
  }
}

// d_lut: one (v2 == NULL) or two consecutive {256-byte table, 16-byte flag area} records
int tpg_launch_pack(tpg_ctx* ctx, const tpg_fbm* fbm, const int32_t* d_rows, const int32_t* d_cols,
                    const uint8_t* d_lut, tpg_view* v, tpg_view* v2) {
  TPG_REQUIRE(v->KG < 2147483647ll && v->Q <= 65535, TPG_EINVAL, "view too large for the pack grid");
  dim3 grid((unsigned)v->KG, (unsigned)v->Q);
  const bool fast_bytes = fbm->bed_bpl == 0 && (fbm->nrow & 7) == 0 && (((uintptr_t)fbm->d_bytes) & 7) == 0;
  const bool fast_bed = fbm->bed_bpl > 0 && !(getenv("TPG_PACK_BED_GENERIC") && atoi(getenv("TPG_PACK_BED_GENERIC")) != 0);  // (A/B)
  if ((fast_bytes || fast_bed) && d_rows == nullptr && !getenv("TPG_PACK_GENERIC")) {
    TPG_REQUIRE((v->KG + 8) * v->Q < 2147483647ll, TPG_EINVAL, "view too large for the pack grid");
    const int xmap = getenv("TPG_PACK_XCD") ? atoi(getenv("TPG_PACK_XCD")) : 1;
    const dim3 g1((unsigned)((xmap ? (v->KG + 7) / 8 * 8 : v->KG) * ((v->Q + PACK_NSUB - 1) / PACK_NSUB)));
    // per-chunk genotype counts beside the layouts (tpg_view::lc_part; TPG_PACK_COUNTS=0: not, the counts kernel reads L)
    static const bool counts_on = !(getenv("TPG_PACK_COUNTS") && atoi(getenv("TPG_PACK_COUNTS")) == 0);
    const int64_t qb = (v->Q + PACK_NSUB - 1) / PACK_NSUB;
    static_assert(PACK_NSUB * 8 * 16 < 1024, "a 10-bit field holds the counts of a chunk: NSUB x 8 threads x 16 individuals");
    if (counts_on)
      for (tpg_view* w : {v, v2}) {
        if (!w || w->lc_part) continue;
        if (tpg_pmalloc((void**)&w->lc_part, sizeof(uint32_t) * (size_t)qb * (size_t)w->KG * TILE) != hipSuccess) { w->lc_part = nullptr; continue; }
        w->lc_chunks = (int)qb;
        w->lc_row = w->KG * TILE;
      }
    uint32_t* lp0 = v->lc_part && (!v2 || v2->lc_part) ? v->lc_part : nullptr;
    uint32_t* lp1 = lp0 && v2 ? v2->lc_part : nullptr;
    if (!lp0)
      for (tpg_view* w : {v, v2})
        if (w && w->lc_part) { tpg_pfree(w->lc_part); w->lc_part = nullptr; w->lc_chunks = 0; }
    auto lds_bytes = [&](int nv, bool t_a, bool t_b) { return (size_t)nv * (256 + 16) + (size_t)((t_a ? 1 : 0) + (t_b ? 1 : 0)) * TILE * TILE; };
    if (v2 && fast_bed)
      TPG_LAUNCH(ctx, "pack2", (tpg_pack_fast_kernel<2, true>), g1, dim3(256), lds_bytes(2, v->T || v->T4, v2->T != nullptr), fbm->d_bytes, fbm->bed_bpl, d_cols, (uint8_t*)d_lut,
                 v->n, v->m, v->Q, v->KG, (uint32_t*)v->T, (uint32_t*)v->L, (uint32_t*)v2->T, (uint32_t*)v2->L,
                 (uint32_t*)v->T4, xmap, lp0, lp1);
    else if (fast_bed)
      TPG_LAUNCH(ctx, "pack", (tpg_pack_fast_kernel<1, true>), g1, dim3(256), lds_bytes(1, v->T || v->T4, false), fbm->d_bytes, fbm->bed_bpl, d_cols, (uint8_t*)d_lut,
                 v->n, v->m, v->Q, v->KG, (uint32_t*)v->T, (uint32_t*)v->L, (uint32_t*)nullptr, (uint32_t*)nullptr,
                 (uint32_t*)v->T4, xmap, lp0, lp1);
    else if (v2)
      TPG_LAUNCH(ctx, "pack2", tpg_pack_fast_kernel<2>, g1, dim3(256), lds_bytes(2, v->T || v->T4, v2->T != nullptr), fbm->d_bytes, fbm->nrow, d_cols, (uint8_t*)d_lut,
                 v->n, v->m, v->Q, v->KG, (uint32_t*)v->T, (uint32_t*)v->L, (uint32_t*)v2->T, (uint32_t*)v2->L,
                 (uint32_t*)v->T4, xmap, lp0, lp1);
    else
      TPG_LAUNCH(ctx, "pack", tpg_pack_fast_kernel<1>, g1, dim3(256), lds_bytes(1, v->T || v->T4, false), fbm->d_bytes, fbm->nrow, d_cols, (uint8_t*)d_lut,
                 v->n, v->m, v->Q, v->KG, (uint32_t*)v->T, (uint32_t*)v->L, (uint32_t*)nullptr, (uint32_t*)nullptr,
                 (uint32_t*)v->T4, xmap, lp0, lp1);
    TPG_CHECK_LAUNCH();
    return TPG_OK;
  }
  // the generic kernel writes T and L only: a view created without T gets it here, and a T4 it was given is dropped
  // (tpg_pairwise_accumulate makes it from T when it is needed)
  for (tpg_view* w : {v, v2}) {
    if (!w) continue;
    if (!w->T) TPG_HIP(tpg_pmalloc((void**)&w->T, w->bytes_each));
    if (w->T4) { tpg_pfree(w->T4); w->T4 = nullptr; }
  }
  TPG_LAUNCH(ctx, "pack", tpg_pack_kernel, grid, dim3(256), 0, fbm->d_bytes, fbm->nrow, fbm->bed_bpl, d_rows, d_cols,
             (uint8_t*)d_lut, v->n, v->m, v->Q, v->KG, (uint32_t*)v->T, (uint32_t*)v->L);
  if (v2)  // no fused form of the generic kernel: a second pass with the second table
    TPG_LAUNCH(ctx, "pack", tpg_pack_kernel, grid, dim3(256), 0, fbm->d_bytes, fbm->nrow, fbm->bed_bpl, d_rows, d_cols,
               (uint8_t*)d_lut + 256 + 16, v2->n, v2->m, v2->Q, v2->KG, (uint32_t*)v2->T, (uint32_t*)v2->L);
  TPG_CHECK_LAUNCH();
  return TPG_OK;
}

// ---------------------------------------------------------------------------
__global__ void tpg_unpack_kernel(const uint32_t* __restrict__ T, const uint32_t* __restrict__ L, int from_L,
                                  int64_t n, int64_t m, int64_t Q, int64_t KG, uint8_t* __restrict__ out) {
  const int64_t total = n * m;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx % n, j = idx / n;
    uint32_t w;
    int e;
    if (!from_L) {
      const int64_t rt = i >> 5, kg = j >> 7;
      const int r = (int)(i & 31), jj = (int)(j & 127);
      const int s = jj >> 5, h = (jj >> 4) & 1;
      e = jj & 15;
      w = T[((rt * KG + kg) * 64 + (r + 32 * h)) * 4 + s];
    } else {
      const int64_t lt = j >> 5, q = i >> 7;
      const int r = (int)(j & 31), ii = (int)(i & 127);
      const int s = ii >> 5, h = (ii >> 4) & 1;
      e = ii & 15;
      w = L[((lt * Q + q) * 64 + (r + 32 * h)) * 4 + s];
    }
    out[idx] = (uint8_t)((w >> tpg_elem_shift(e)) & 3u);
  }
}

// T from L: one workgroup per 128 x 128 tile, the four L blocks of the tile unpacked into LDS code bytes and emitted as
// its four T blocks (the second half of the generic pack kernel).  Only runs for a view that was packed without T
// (tpg_view_create_pair) and is then handed to something that contracts over loci on the 2-bit layout.
__global__ __launch_bounds__(256) void tpg_l2t_kernel(const uint32_t* __restrict__ L, int64_t Q, int64_t KG,
                                                      uint32_t* __restrict__ T) {
  __shared__ __attribute__((aligned(16))) uint8_t codes[TILE * LDS_STRIDE];  // codes[locus][individual]
  const int tid = threadIdx.x;
  const int64_t bj = blockIdx.x, bi = blockIdx.y;
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int idx = tid + 256 * it;
    const int s = idx & 3, lane = (idx >> 2) & 63, tl = idx >> 8;
    const int r = lane & 31, h = lane >> 5;
    const uint32_t w = L[(((bj * 4 + tl) * Q + bi) * 64 + lane) * 4 + s];  // locus 32 tl + r, individuals 32 s + 16 h + e
#pragma unroll
    for (int e = 0; e < 16; e++) codes[(32 * tl + r) * LDS_STRIDE + 32 * s + 16 * h + e] = (uint8_t)((w >> tpg_elem_shift(e)) & 3u);
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int idx = tid + 256 * it;
    const int s = idx & 3, lane = (idx >> 2) & 63, tl = idx >> 8;
    const int r = lane & 31, h = lane >> 5;
    const int ind = 32 * tl + r;  // T: individual 32 tl + r, loci 32 s + 16 h + e
    uint32_t w = 0;
#pragma unroll
    for (int e = 0; e < 16; e++) w |= (uint32_t)codes[(32 * s + 16 * h + e) * LDS_STRIDE + ind] << tpg_elem_shift(e);
    T[(((bi * 4 + tl) * KG + bj) * 64 + lane) * 4 + s] = w;
  }
}

int tpg_view_need_T(tpg_ctx* ctx, const tpg_view* v) {
  if (v->T) return TPG_OK;
  TPG_REQUIRE(v->KG < 2147483647ll && v->Q <= 65535, TPG_EINVAL, "view too large for the pack grid");
  uint4* t = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&t, v->bytes_each));
  TPG_LAUNCH(ctx, "l2t", tpg_l2t_kernel, dim3((unsigned)v->KG, (unsigned)v->Q), dim3(256), 0, (const uint32_t*)v->L, v->Q,
             v->KG, (uint32_t*)t);
  v->T = t;
  TPG_CHECK_LAUNCH();
  return TPG_OK;
}

int tpg_launch_unpack(tpg_ctx* ctx, const tpg_view* v, uint8_t* d_codes, int from_L) {
  if (!from_L) TPG_TRY(tpg_view_need_T(ctx, v));
  TPG_LAUNCH(ctx, "unpack", tpg_unpack_kernel, dim3(2048), dim3(256), 0, (const uint32_t*)v->T,
             (const uint32_t*)v->L, from_L, v->n, v->m, v->Q, v->KG, d_codes);
  TPG_CHECK_LAUNCH();
  return TPG_OK;
}

// ---------------------------------------------------------------------------
// One workgroup per locus: the npop population frequencies are computed once into LDS.
__global__ __launch_bounds__(256) void tpg_synth_kernel(uint8_t* __restrict__ out, uint64_t seed, int64_t nrow,
                                                        int64_t ncol, int64_t j0, int npop, uint32_t miss_thresh,
                                                        int imputed_bytes) {
  __shared__ uint32_t pjg[1024];
  for (int64_t j = blockIdx.x; j < ncol; j += gridDim.x) {
    __syncthreads();
    for (int g = threadIdx.x; g < npop; g += blockDim.x) pjg[g] = tpg_synth_pjg(seed, (uint64_t)(j0 + j), (uint32_t)g, (uint32_t)npop);
    __syncthreads();
    for (int64_t i = threadIdx.x; i < nrow; i += blockDim.x)
      out[i + j * nrow] = tpg_synth_geno(seed, (uint64_t)i, (uint64_t)(j0 + j), pjg[i % npop], miss_thresh, imputed_bytes);
  }
}

int tpg_launch_synth(tpg_ctx* ctx, uint8_t* d_bytes, uint64_t seed, int64_t nrow, int64_t ncol, int64_t j0,
                     int npop, uint32_t miss_thresh, int imputed_bytes) {
  unsigned grid = (unsigned)(ncol < 65536 ? ncol : 65536);
  TPG_LAUNCH(ctx, "synth", tpg_synth_kernel, dim3(grid), dim3(256), 0, d_bytes, seed, nrow, ncol, j0, npop,
             miss_thresh, imputed_bytes);
  TPG_CHECK_LAUNCH();
  return TPG_OK;
}
