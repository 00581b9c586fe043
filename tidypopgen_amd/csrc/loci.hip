// loci.hip -- per-locus sweeps over the locus-tiled layout L.
//
//  * tpg_loci_counts_kernel: genotype counts per locus (n0,n1,n2,nNA).  One wave per 32-locus
//    tile streams that tile's Q contiguous 1-KiB blocks with 16-B coalesced loads; lane (r,h)
//    owns locus r and popcounts its own dwords, so the only cross-lane step is one lane^32
//    exchange at the end.  HBM-bound: n/4 bytes per locus in, 16 B out.  (A view made by the fast pack kernel carries
//    per-chunk counts from the pack itself: tpg_loci_counts_sum_kernel adds those up and L is not read.)
//  * tpg_grouped_counts_kernel: per locus x class counts as an int8 MFMA contraction over
//    individuals, D[locus][class] = sum_i plane[i][locus] * onehot[i][class] for the planes
//    {het, hom-alt, valid}.  Exact in int32, arbitrary class assignment, no atomics.
//  * finalize kernels turn counts into the doubles the reference returns
//    (src/alt_freq_dip_pseudo_cpp.cpp:43-57, src/grouped_alt_freq_dip_pseudo_cpp.cpp:46-55,
//    src/grouped_missingness_cpp.cpp:23-31, src/grouped_summaries_dip_pseudo_cpp.cpp:50-57).
//    The reference's sums of x*mult and ploidy are sums of {0, 0.5, 1, 2}-multiples, exact in
//    double in any order, so integer counts + one conversion reproduce them bit for bit.
#include <string.h>

#include "common.h"
#include <type_traits>
#include "devfrag.h"

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tpg_loci_counts_kernel(const uint4* __restrict__ L, int64_t n_lt,
                                                              int64_t Q, int64_t n, int64_t m,
                                                              int4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t lt = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (lt >= n_lt) return;
  const uint4* p = L + (lt * Q) * 64 + lane;
  int c_lo = 0, c_hi = 0, c_both = 0;
  auto acc = [&](uint32_t w) {
    const uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
    c_lo += __popc(lo);
    c_hi += __popc(hi);
    c_both += __popc(lo & hi);
  };
  int64_t q = 0;
  for (; q + 4 <= Q; q += 4) {
    uint4 a0 = p[(q + 0) * 64], a1 = p[(q + 1) * 64], a2 = p[(q + 2) * 64], a3 = p[(q + 3) * 64];
    acc(a0.x); acc(a0.y); acc(a0.z); acc(a0.w);
    acc(a1.x); acc(a1.y); acc(a1.z); acc(a1.w);
    acc(a2.x); acc(a2.y); acc(a2.z); acc(a2.w);
    acc(a3.x); acc(a3.y); acc(a3.z); acc(a3.w);
  }
  for (; q < Q; q++) {
    uint4 a0 = p[q * 64];
    acc(a0.x); acc(a0.y); acc(a0.z); acc(a0.w);
  }
  c_lo += __shfl_xor(c_lo, 32);
  c_hi += __shfl_xor(c_hi, 32);
  c_both += __shfl_xor(c_both, 32);
  const int64_t j = lt * 32 + (lane & 31);
  if (lane < 32 && j < m) {
    const int n3 = c_both, n1 = c_lo - c_both, n2 = c_hi - c_both;
    const int npad = (int)(Q * 128 - n);
    out[j] = make_int4((int)(Q * 128) - n1 - n2 - n3, n1, n2, n3 - npad);
  }
}

// the same counts from what the fast pack kernel left beside the layouts (tpg_view::lc_part): per locus the sum over the
// chunks of 256 individuals of {codes with bit 0 set, with bit 1 set, with both}, 4 B per chunk and locus instead of n / 4
__global__ __launch_bounds__(256) void tpg_loci_counts_sum_kernel(const uint32_t* __restrict__ part, int chunks, int64_t row,
                                                                  int64_t n, int64_t m, int4* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  int lo = 0, hi = 0, both = 0;
  for (int c = 0; c < chunks; c++) {
    const uint32_t w = part[(int64_t)c * row + j];
    lo += (int)(w & 1023u);
    hi += (int)((w >> 10) & 1023u);
    both += (int)(w >> 20);
  }
  const int n1 = lo - both, n2 = hi - both;
  out[j] = make_int4((int)n - n1 - n2 - both, n1, n2, both);
}

int tpg_launch_loci_counts(tpg_ctx* ctx, const tpg_view* v, int32_t* d_counts) {
  if (v->lc_part) {
    TPG_LAUNCH(ctx, "loci_counts", tpg_loci_counts_sum_kernel, dim3((unsigned)ceil_div(v->m, 256)), dim3(256), 0,
               (const uint32_t*)v->lc_part, v->lc_chunks, v->lc_row, v->n, v->m, (int4*)d_counts);
    TPG_CHECK_LAUNCH();
    return TPG_OK;
  }
  const int64_t n_lt = v->KG * 4;
  TPG_LAUNCH(ctx, "loci_counts", tpg_loci_counts_kernel, dim3((unsigned)ceil_div(n_lt, 4)), dim3(256), 0,
             (const uint4*)v->L, n_lt, v->Q, v->n, v->m, (int4*)d_counts);
  TPG_CHECK_LAUNCH();
  return TPG_OK;
}

extern "C" int tpg_loci_counts(tpg_ctx* ctx, const tpg_view* v, int32_t* out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && out, TPG_EINVAL, "null argument");
  OutBuf o;
  TPG_TRY(o.init(out, sizeof(int32_t) * 4 * (size_t)v->m));
  TPG_TRY(tpg_launch_loci_counts(ctx, v, o.dev<int32_t>()));
  return o.commit(ctx);
}

// per-individual counts: the same kernel on the individual-tiled layout (rows = individuals, contraction over loci)
extern "C" int tpg_indiv_counts(tpg_ctx* ctx, const tpg_view* v, int32_t* out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && out, TPG_EINVAL, "null argument");
  OutBuf o;
  TPG_TRY(o.init(out, sizeof(int32_t) * 4 * (size_t)v->n));
  TPG_TRY(tpg_view_need_T(ctx, v));
  const int64_t n_rt = v->Q * 4;
  TPG_LAUNCH(ctx, "indiv_counts", tpg_loci_counts_kernel, dim3((unsigned)ceil_div(n_rt, 4)), dim3(256), 0,
             (const uint4*)v->T, n_rt, v->KG, v->m, v->n, o.dev<int4>());
  TPG_CHECK_LAUNCH();
  return o.commit(ctx);
}

// src/gt_ind_hetero.cpp:11-42: row 0 = heterozygous loci, row 1 = missing loci, per individual
__global__ void tpg_ind_hetero_kernel(const int4* __restrict__ counts, int64_t n, int32_t* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    out[2 * i] = counts[i].y;
    out[2 * i + 1] = counts[i].w;
  }
}

extern "C" int tpg_gt_ind_hetero(tpg_ctx* ctx, const tpg_view* v, int32_t* out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && out, TPG_EINVAL, "null argument");
  int32_t* d_counts = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_counts, sizeof(int32_t) * 4 * (size_t)v->n));
  int rc = tpg_indiv_counts(ctx, v, d_counts);
  OutBuf o;
  if (rc == TPG_OK) rc = o.init(out, sizeof(int32_t) * 2 * (size_t)v->n);
  if (rc == TPG_OK) {
    TPG_LAUNCH(ctx, "ind_hetero", tpg_ind_hetero_kernel, dim3(256), dim3(256), 0, (const int4*)d_counts, v->n,
               o.dev<int32_t>());
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { tpg_set_error("ind_hetero: %s", hipGetErrorString(e)); rc = TPG_EHIP; }
  }
  tpg_pfree(d_counts);
  TPG_TRY(rc);
  return o.commit(ctx);
}

// src/gt_pi_diploid.cpp:28-34
__global__ void tpg_pi_kernel(const int4* __restrict__ counts, int64_t m, double* __restrict__ pi) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = counts[j];
    const double cnt = (double)(c.y + 2 * c.z), valid = (double)(2 * (c.x + c.y + c.z));
    pi[j] = valid > 0 ? (cnt * (valid - cnt) / (valid * (valid - 1) / 2)) : __longlong_as_double(0x7FF8000000000000ll);
  }
}

extern "C" int tpg_gt_pi_diploid(tpg_ctx* ctx, const tpg_view* v, double* pi) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && pi, TPG_EINVAL, "null argument");
  int32_t* d_counts = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_counts, sizeof(int32_t) * 4 * (size_t)v->m));
  int rc = tpg_launch_loci_counts(ctx, v, d_counts);
  OutBuf o;
  if (rc == TPG_OK) rc = o.init(pi, sizeof(double) * (size_t)v->m);
  if (rc == TPG_OK) {
    TPG_LAUNCH(ctx, "pi_diploid", tpg_pi_kernel, dim3(1024), dim3(256), 0, (const int4*)d_counts, v->m, o.dev<double>());
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { tpg_set_error("pi: %s", hipGetErrorString(e)); rc = TPG_EHIP; }
  }
  tpg_pfree(d_counts);
  TPG_TRY(rc);
  return o.commit(ctx);
}

// ---------------------------------------------------------------------------
// one-hot B fragments of the grouped counts, as FP4 operands (nibble 0x2 = 1.0): OH[q][S][gt][lane = (c, h)][16 B] for the
// two 64-individual steps S of a 128-individual group q.  The A side of that kernel takes the dwords s = 2 S, 2 S + 1 of an
// L block apart into their even and odd 2-bit slots (P & 0x33333333, (P >> 2) & 0x33333333), so operand dword d = 2 s' + odd
// holds, at nibble j, element e = (j >> 1) + 8 (j & 1) + 4 odd of source dword s = 2 S + s' (tpg_elem_shift), i.e.
// individual 128 q + 32 s + 16 h + e.
__global__ void tpg_onehot_kernel(const int32_t* __restrict__ cls, int64_t n, int64_t Q, int GT,
                                  uint4* __restrict__ OH) {
  const int64_t total = Q * 2 * GT * 64;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    const int64_t t = idx >> 6;
    const int gt = (int)(t % GT);
    const int64_t qS = t / GT;
    const int S = (int)(qS & 1);
    const int64_t q = qS >> 1;
    const int c = 32 * gt + (lane & 31), h = lane >> 5;
    uint32_t w[4] = {0, 0, 0, 0};
    for (int d = 0; d < 4; d++)
      for (int j = 0; j < 8; j++) {
        const int e = (j >> 1) + 8 * (j & 1) + 4 * (d & 1);
        const int64_t i = 128 * q + 32 * (2 * S + (d >> 1)) + 16 * h + e;
        if (i < n && cls[i] == c) w[d] |= 2u << (4 * j);
      }
    OH[idx] = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// cnt[plane][locus][class] (row-major by locus, Cpad classes), planes: 0 het, 1 hom-alt, 2 valid.
// One wave owns GC_NLT consecutive 32-locus tiles and GT class tiles (3 planes x GT x GC_NLT accumulator tiles); the
// four waves of a workgroup share the one-hot fragments of a 128-individual group through double-buffered LDS (each
// wave fetches a quarter, one barrier per group), so a 1-KiB fragment fetched from L2 feeds 4 x 3 x GC_NLT MFMAs.
// (With a fragment per MFMA straight from global memory, every wave re-read the whole one-hot array: 10 GB through the
// L1s per launch at 5 000 x 1 000 000 x 51 groups, 1.07 ms for 1.25 GB of genotypes.)  One locus tile per wave and three
// workgroups per CU: a workgroup lives for 40 groups, and with one resident workgroup per CU (two tiles per wave, 432
// registers) its prologue, its barriers and its flush were all exposed -- 0.84 ms against 0.59.
//
// The products are 0 / 1, so they run on the FP4 matrix cores (v_mfma_scale_f32_32x32x64_f8f6f4: 64 individuals per
// instruction where the int8 form takes 32; sums of at most n ones, exact in FP32 below 2^24).  The 2-bit code IS the
// operand: with X a dword of eight codes in nibbles, Y = X & 0x11111111 (bit 0: heterozygous or missing, FP4 0.5, block
// scale 2), Z = X & 0x22222222 (bit 1: homozygous alt or missing, FP4 1.0) and W = Y & (Z >> 1) (missing) are three
// operand dwords for four instructions; het = sum Y - sum W, hom-alt = sum Z - sum W, valid = class size - sum W.
#define GC_NLT 1
#define GC_D 4
template <int C, class F>
__device__ __forceinline__ void gc_static_for(F&& f) {
  if constexpr (C > 0) {
    gc_static_for<C - 1>(f);
    f(std::integral_constant<int, C - 1>{});
  }
}
typedef int gc_v8i __attribute__((ext_vector_type(8)));
typedef float gc_v16f __attribute__((ext_vector_type(16)));
#define GC_MFMA(a, b, c, sa) \
  __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((gc_v8i){(int)(a)[0], (int)(a)[1], (int)(a)[2], (int)(a)[3], 0, 0, 0, 0}, \
                                                  (gc_v8i){(int)(b).x, (int)(b).y, (int)(b).z, (int)(b).w, 0, 0, 0, 0}, (c), 4, 4, 0, (sa), 0, 0x7F7F7F7F)
__device__ __forceinline__ int64_t tpg_gc_uniform64(int64_t x) {  // a wave-uniform value the compiler keeps in SGPRs
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)x >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <int GT>
__global__ __launch_bounds__(256, 3) void tpg_grouped_counts_kernel(const uint4* __restrict__ L,
                                                                    const uint4* __restrict__ OH, int64_t n_lt,
                                                                    int64_t Q, int gt0, int GT_total,
                                                                    const int32_t* __restrict__ csize,
                                                                    int32_t* __restrict__ cnt, int64_t Mpad,
                                                                    int Cpad) {
  __shared__ __attribute__((aligned(16))) uint4 ohb[2][2 * GT][64];  // [buffer][step * GT + class tile][lane]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t lt0 = ((int64_t)blockIdx.x * 4 + wv) * GC_NLT;  // may lie past n_lt: the wave still serves the LDS
  gc_v16f acc[3][GC_NLT][GT];
#pragma unroll
  for (int p = 0; p < 3; p++)
#pragma unroll
    for (int t = 0; t < GC_NLT; t++)
#pragma unroll
      for (int g = 0; g < GT; g++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[p][t][g][r] = 0.f;
  // (wave-uniform bases in SGPRs + a 32-bit lane offset per group: 32-bit group indices, no 64-bit per-lane pointers)
  const int Qi = (int)Q;
  const int64_t lt0u = tpg_gc_uniform64(lt0);
  const char* pa[GC_NLT];
#pragma unroll
  for (int t = 0; t < GC_NLT; t++) pa[t] = (const char*)(L + ((lt0u + t < n_lt ? lt0u + t : 0) * Q) * 64);  // past the end: a copy
  auto LDG = [&](const char* p, int q) {
    uint32_t off = (uint32_t)lane * 16u + (uint32_t)q * 1024u;
    asm("" : "+v"(off));
    return *(const uint4*)(p + off);
  };
  // this wave's share of a group's 2 * GT one-hot fragments: items wv, wv + 4, ... (fragment (S, g) of group q sits at
  // OH[((q * 2 + S) * GT_total + gt0 + g) * 64 + lane])
  constexpr int NIT = (2 * GT + 3) / 4;
  const char* po = (const char*)(OH + (int64_t)gt0 * 64);
  auto frag = [&](int q, int it) {
    uint32_t off = (uint32_t)lane * 16u + (uint32_t)(((q * 2 + it / GT) * GT_total + it % GT)) * 1024u;
    return *(const uint4*)(po + off);
  };
  // genotype blocks of groups q .. q + GC_D - 1 in GC_D rotating register slots (the HBM stream is fetched GC_D - 1 groups
  // ahead; the loop is unrolled by GC_D so that no slot is copied -- a copy would wait for the load just issued)
  // the one-hot fragments of group q' wait in register slot q' & 1, fetched TWO groups ahead of the barrier behind which they
  // are used (one group ahead they had not arrived when the group's MFMAs were done: the wave waited, three others with it)
  uint4 R[GC_D][GC_NLT], on[2][NIT];
#pragma unroll
  for (int d = 0; d < GC_D - 1; d++)
#pragma unroll
    for (int t = 0; t < GC_NLT; t++) R[d][t] = LDG(pa[t], d < Qi ? d : Qi - 1);
#pragma unroll
  for (int j = 0; j < NIT; j++) {
    const int it = wv + 4 * j;
    if (it < 2 * GT) {
      ohb[0][it][lane] = frag(0, it);
      on[1][j] = frag(Qi > 1 ? 1 : 0, it);
    }
  }
  tpg_lds_barrier();
  auto group = [&](auto Cc, auto Mm, int q) {
    constexpr int C = decltype(Cc)::value, M = decltype(Mm)::value;
    static_assert(GC_D % 2 == 0, "the slot of a group's fragments is its parity, known at compile time in the unrolled loop");
    const int qn2 = q + GC_D - 1 < Qi ? q + GC_D - 1 : Qi - 1, qo2 = q + 2 < Qi ? q + 2 : Qi - 1;
    const int cur = q & 1;
#pragma unroll
    for (int t = 0; t < GC_NLT; t++) R[M][t] = LDG(pa[t], qn2);
#pragma unroll
    for (int j = 0; j < NIT; j++) {
      const int it = wv + 4 * j;
      if (it < 2 * GT) on[C & 1][j] = frag(qo2, it);
    }
#pragma unroll
    for (int S = 0; S < 2; S++) {
      uint32_t fy[GC_NLT][4], fz[GC_NLT][4], fw[GC_NLT][4];
#pragma unroll
      for (int t = 0; t < GC_NLT; t++) {
        const uint32_t P[2] = {S == 0 ? R[C][t].x : R[C][t].z, S == 0 ? R[C][t].y : R[C][t].w};
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const uint32_t X = (d & 1) ? (P[d >> 1] >> 2) & 0x33333333u : P[d >> 1] & 0x33333333u;
          fy[t][d] = X & 0x11111111u;
          fz[t][d] = X & 0x22222222u;
          fw[t][d] = fy[t][d] & (fz[t][d] >> 1);
        }
      }
#pragma unroll
      for (int g = 0; g < GT; g++) {
        const uint4 b = ohb[cur][S * GT + g][lane];
#pragma unroll
        for (int t = 0; t < GC_NLT; t++) {
          acc[0][t][g] = GC_MFMA(fy[t], b, acc[0][t][g], (int)0x80808080);
          acc[1][t][g] = GC_MFMA(fz[t], b, acc[1][t][g], 0x7F7F7F7F);
          acc[2][t][g] = GC_MFMA(fw[t], b, acc[2][t][g], (int)0x80808080);
        }
      }
    }
    // the other buffer was last read in group q - 1, which every wave left through the barrier below
#pragma unroll
    for (int j = 0; j < NIT; j++) {
      const int it = wv + 4 * j;
      if (it < 2 * GT) ohb[cur ^ 1][it][lane] = on[(C & 1) ^ 1][j];
    }
    tpg_lds_barrier();
  };
  for (int q = 0; q < Qi; q += GC_D)  // Q is the same for every wave: all of them meet every barrier
    gc_static_for<GC_D>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      if (q + k < Qi) group(std::integral_constant<int, k>{}, std::integral_constant<int, (k + GC_D - 1) % GC_D>{}, q + k);
    });
#pragma unroll
  for (int t = 0; t < GC_NLT; t++) {
    if (lt0 + t >= n_lt) break;
#pragma unroll
    for (int g = 0; g < GT; g++) {
      const int cs = csize[32 * (gt0 + g) + (lane & 31)];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int64_t row = (lt0 + t) * 32 + tpg_cd_row(r, lane);
        const int64_t o = row * Cpad + 32 * (gt0 + g) + (lane & 31);
        const int nmiss = (int)acc[2][t][g][r];
        cnt[o] = (int)acc[0][t][g][r] - nmiss;
        cnt[Mpad * Cpad + o] = (int)acc[1][t][g][r] - nmiss;
        cnt[2 * Mpad * Cpad + o] = cs - nmiss;
      }
    }
  }
}

GroupedCounts::~GroupedCounts() {
  if (cnt && !borrowed) tpg_pfree(cnt);
}

int tpg_grouped_counts(tpg_ctx* ctx, const tpg_view* v, const int32_t* h_cls, int nclass, GroupedCounts* out) {
  TPG_REQUIRE(nclass > 0, TPG_EINVAL, "no classes");
  for (int64_t i = 0; i < v->n; i++)
    TPG_REQUIRE(h_cls[i] >= 0 && h_cls[i] < nclass, TPG_EINVAL, "class id %d of individual %lld out of [0,%d)",
                h_cls[i], (long long)i, nclass);
  const int GT = (int)ceil_div(nclass, 32);
  const int64_t n_lt = v->KG * 4;
  if (v->gc_cache.cnt && v->gc_cache.nclass == nclass && v->gc_cls.size() == (size_t)v->n &&
      memcmp(v->gc_cls.data(), h_cls, sizeof(int32_t) * (size_t)v->n) == 0) {
    *out = v->gc_cache;  // shallow copy of the cached buffer
    out->borrowed = true;
    return TPG_OK;
  }
  if (v->gc_cache.cnt) { tpg_pfree(v->gc_cache.cnt); v->gc_cache.cnt = nullptr; }
  out->Mpad = n_lt * 32;
  out->Cpad = GT * 32;
  out->nclass = nclass;
  TPG_HIP(tpg_pmalloc((void**)&out->cnt, sizeof(int32_t) * 3 * (size_t)out->Mpad * (size_t)out->Cpad));
  TPG_REQUIRE(v->n < (1 << 24), TPG_EUNSUPPORTED, "grouped counts of more than 2^24 individuals");  // FP32 sums of ones
  // class sizes (the valid count of a class is its size minus its missing genotypes), behind the class ids
  std::vector<int32_t> h_up((size_t)v->n + (size_t)GT * 32, 0);
  memcpy(h_up.data(), h_cls, sizeof(int32_t) * (size_t)v->n);
  for (int64_t i = 0; i < v->n; i++) h_up[(size_t)v->n + (size_t)h_cls[i]]++;
  int32_t* d_cls = nullptr;
  uint4* d_oh = nullptr;
  TPG_HIP(tpg_pmalloc((void**)&d_cls, sizeof(int32_t) * h_up.size()));
  hipError_t e = tpg_pmalloc((void**)&d_oh, (size_t)v->Q * 2 * GT * 1024);
  if (e != hipSuccess) { tpg_pfree(d_cls); tpg_set_error("hipMalloc one-hot: %s", hipGetErrorString(e)); return TPG_EHIP; }
  int rc = TPG_OK;
  e = tpg_h2d_async(ctx, d_cls, h_up.data(), sizeof(int32_t) * h_up.size());
  if (e != hipSuccess) { tpg_set_error("class upload: %s", hipGetErrorString(e)); rc = TPG_EHIP; }
  if (rc == TPG_OK) {
    TPG_LAUNCH(ctx, "onehot", tpg_onehot_kernel, dim3(1024), dim3(256), 0, d_cls, v->n, v->Q, GT, d_oh);
    const unsigned grid = (unsigned)ceil_div(n_lt, 4 * GC_NLT);
    int g0 = 0;
    while (g0 < GT) {
      if (GT - g0 >= 2) {
        TPG_LAUNCH(ctx, "grouped_counts", tpg_grouped_counts_kernel<2>, dim3(grid), dim3(256), 0, (const uint4*)v->L,
                   (const uint4*)d_oh, n_lt, v->Q, g0, GT, (const int32_t*)(d_cls + v->n), out->cnt, out->Mpad, out->Cpad);
        g0 += 2;
      } else {
        TPG_LAUNCH(ctx, "grouped_counts", tpg_grouped_counts_kernel<1>, dim3(grid), dim3(256), 0, (const uint4*)v->L,
                   (const uint4*)d_oh, n_lt, v->Q, g0, GT, (const int32_t*)(d_cls + v->n), out->cnt, out->Mpad, out->Cpad);
        g0 += 1;
      }
    }
    e = hipGetLastError();  // d_cls / d_oh go back to this context's pool below: reuse is stream-ordered, no wait needed
    if (e != hipSuccess) { tpg_set_error("grouped counts: %s", hipGetErrorString(e)); rc = TPG_EHIP; }
  }
  tpg_pfree(d_cls);
  tpg_pfree(d_oh);
  if (rc == TPG_OK) {  // the view keeps the buffer; the caller borrows it
    v->gc_cache = *out;
    v->gc_cache.borrowed = false;
    v->gc_cls.assign(h_cls, h_cls + v->n);
    out->borrowed = true;
  }
  return rc;
}

// ---------------------------------------------------------------------------
// Class scheme shared by the grouped entry points: with pseudohaploids present class = 2*g + (ploidy==1),
// otherwise class = g.  (ploidy is 1 or 2: the reference's dip_pseudo kernels assume it too.)
struct ClassPlan {
  std::vector<int32_t> cls;
  std::vector<int32_t> group_size;
  int nclass = 0;
  int has_hap = 0;
};

static int make_class_plan(const tpg_view* v, const int32_t* groupIds0, int ngroups, const double* ploidy,
                           ClassPlan* cp) {
  TPG_REQUIRE(ngroups > 0, TPG_EINVAL, "ngroups must be positive");
  cp->has_hap = 0;
  if (ploidy)
    for (int64_t i = 0; i < v->n; i++) {
      TPG_REQUIRE(ploidy[i] == 1.0 || ploidy[i] == 2.0, TPG_EUNSUPPORTED,
                  "ploidy[%lld] = %g: only diploid (2) and pseudohaploid (1) individuals are supported",
                  (long long)i, ploidy[i]);
      if (ploidy[i] == 1.0) cp->has_hap = 1;
    }
  cp->nclass = ngroups * (cp->has_hap ? 2 : 1);
  cp->cls.resize((size_t)v->n);
  cp->group_size.assign((size_t)ngroups, 0);
  for (int64_t i = 0; i < v->n; i++) {
    const int g = groupIds0 ? groupIds0[i] : 0;
    TPG_REQUIRE(g >= 0 && g < ngroups, TPG_EINVAL, "groupIds[%lld] = %d out of [0,%d)", (long long)i, g, ngroups);
    cp->group_size[(size_t)g]++;
    cp->cls[(size_t)i] = cp->has_hap ? 2 * g + (ploidy[i] == 1.0 ? 1 : 0) : g;
  }
  return TPG_OK;
}

struct GroupVals {
  double alt, valid, het2, nvalid_ind;
};

__device__ __forceinline__ GroupVals tpg_group_vals(const int32_t* __restrict__ cnt, int64_t Mpad, int Cpad,
                                                    int64_t j, int g, int has_hap) {
  GroupVals r;
  const int64_t plane = Mpad * Cpad;
  if (!has_hap) {
    const int64_t o = j * Cpad + g;
    const int n1 = cnt[o], n2 = cnt[plane + o], nv = cnt[2 * plane + o];
    r.alt = (double)(n1 + 2 * n2);
    r.valid = (double)(2 * nv);
    r.het2 = (double)(2 * n1);
    r.nvalid_ind = (double)nv;
  } else {
    const int64_t o = j * Cpad + 2 * g;
    const int n1d = cnt[o], n2d = cnt[plane + o], nvd = cnt[2 * plane + o];
    const int n1h = cnt[o + 1], n2h = cnt[plane + o + 1], nvh = cnt[2 * plane + o + 1];
    r.alt = (double)(n1d + 2 * n2d) + 0.5 * (double)(n1h + 2 * n2h);  // x * 1/(3-ploidy)
    r.valid = (double)(2 * nvd + nvh);                                 // sum of ploidy
    r.het2 = (double)(2 * (n1d + n1h));                                // +2 per x == 1
    r.nvalid_ind = (double)(nvd + nvh);
  }
  return r;
}

// mode 0: grouped_alt_freq (out m x 2G; as_counts), 1: grouped_missingness (out m x G),
// 2: grouped_summaries (o0..o3 m x G, each may be null).
// A workgroup owns 64 consecutive loci: the count rows are read with the class index on the lanes
// (contiguous 4-byte loads), results are staged in LDS and written with the locus index on the lanes
// (contiguous 8-byte stores into the column-major outputs).
__global__ __launch_bounds__(256) void tpg_grouped_finalize_kernel(const int32_t* __restrict__ cnt, int64_t Mpad,
                                                                   int Cpad, int64_t m, int G, int has_hap, int mode,
                                                                   int as_counts,
                                                                   const int32_t* __restrict__ group_size,
                                                                   double* __restrict__ o0, double* __restrict__ o1,
                                                                   double* __restrict__ o2, double* __restrict__ o3) {
  // [output][group chunk][locus]; only as many outputs as the mode writes (2 / 1 / 4): 33 KiB instead of 66 for the
  // allele frequencies, i.e. four workgroups per CU instead of two
  extern __shared__ double tile_raw[];
  double (*tile)[32][65] = (double (*)[32][65])tile_raw;
  const int64_t j0 = (int64_t)blockIdx.x * 64;
  for (int g0 = 0; g0 < G; g0 += 32) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {
      const int gl = idx & 31, l = idx >> 5;
      const int g = g0 + gl;
      const int64_t j = j0 + l;
      if (g < G && j < m) {
        const GroupVals gv = tpg_group_vals(cnt, Mpad, Cpad, j, g, has_hap);
        if (mode == 0) {
          tile[0][gl][l] = as_counts ? gv.alt : gv.alt / gv.valid;
          tile[1][gl][l] = gv.valid;
        } else if (mode == 1) {
          tile[0][gl][l] = (double)group_size[g] - gv.nvalid_ind;
        } else {
          const double f = gv.alt / gv.valid;
          tile[0][gl][l] = f;
          tile[1][gl][l] = 1 - f;
          tile[2][gl][l] = gv.valid;
          tile[3][gl][l] = gv.het2 / gv.valid;
        }
      }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {
      const int l = idx & 63, gl = idx >> 6;
      const int g = g0 + gl;
      const int64_t j = j0 + l;
      if (g >= G || j >= m) continue;
      if (mode == 0) {
        o0[j + (int64_t)g * m] = tile[0][gl][l];
        o0[j + (int64_t)(G + g) * m] = tile[1][gl][l];
      } else if (mode == 1) {
        o0[j + (int64_t)g * m] = tile[0][gl][l];
      } else {
        if (o0) o0[j + (int64_t)g * m] = tile[0][gl][l];
        if (o1) o1[j + (int64_t)g * m] = tile[1][gl][l];
        if (o2) o2[j + (int64_t)g * m] = tile[2][gl][l];
        if (o3) o3[j + (int64_t)g * m] = tile[3][gl][l];
      }
    }
  }
}

// src/gt_grouped_pi_diploid.cpp:24-38: per locus x group, pi = x (v - x) / (v (v - 1) / 2), no NA guard
__global__ void tpg_grouped_pi_kernel(const int32_t* __restrict__ cnt, int64_t Mpad, int Cpad, int64_t m, int G,
                                      double* __restrict__ pi, double* __restrict__ nvalid) {
  const int64_t total = m * G;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(idx % G);  // class index on the lanes: contiguous count reads
    const int64_t j = idx / G;
    const GroupVals gv = tpg_group_vals(cnt, Mpad, Cpad, j, g, 0);
    pi[j + (int64_t)g * m] = gv.alt * (gv.valid - gv.alt) / (gv.valid * (gv.valid - 1) / 2);
    if (nvalid) nvalid[j + (int64_t)g * m] = gv.valid;
  }
}

extern "C" int tpg_gt_grouped_pi_diploid(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups,
                                         double* pi, double* n) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && groupIds0 && pi, TPG_EINVAL, "null argument");
  ClassPlan cp;
  TPG_TRY(make_class_plan(v, groupIds0, ngroups, nullptr, &cp));
  GroupedCounts gc;
  TPG_TRY(tpg_grouped_counts(ctx, v, cp.cls.data(), cp.nclass, &gc));
  const size_t bytes = sizeof(double) * (size_t)v->m * (size_t)ngroups;
  OutBuf op, on;
  TPG_TRY(op.init(pi, bytes));
  if (n) TPG_TRY(on.init(n, bytes));
  TPG_LAUNCH(ctx, "grouped_pi", tpg_grouped_pi_kernel, dim3(2048), dim3(256), 0, (const int32_t*)gc.cnt, gc.Mpad, gc.Cpad,
             v->m, ngroups, op.dev<double>(), on.dev<double>());
  TPG_CHECK_LAUNCH();
  TPG_HIP(hipStreamSynchronize(ctx->stream));
  TPG_TRY(op.commit(ctx));
  if (n) TPG_TRY(on.commit(ctx));
  return TPG_OK;
}

// genotype counts per locus x group, the table gt_grouped_hwe fills before each exact test
// (src/hwe.cpp:238-250): out[k][j + g m] = individuals of group g with k alternate alleles at locus j
__global__ void tpg_grouped_genotype_counts_kernel(const int32_t* __restrict__ cnt, int64_t Mpad, int Cpad, int64_t m,
                                                   int G, int32_t* __restrict__ out) {
  const int64_t total = m * G, plane = Mpad * Cpad;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(idx % G);
    const int64_t j = idx / G;
    const int64_t o = j * Cpad + g;
    const int n1 = cnt[o], n2 = cnt[plane + o], nv = cnt[2 * plane + o];
    out[j + (int64_t)g * m] = nv - n1 - n2;
    out[total + j + (int64_t)g * m] = n1;
    out[2 * total + j + (int64_t)g * m] = n2;
  }
}

extern "C" int tpg_grouped_genotype_counts(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups,
                                           int32_t* out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && groupIds0 && out, TPG_EINVAL, "null argument");
  ClassPlan cp;
  TPG_TRY(make_class_plan(v, groupIds0, ngroups, nullptr, &cp));
  GroupedCounts gc;
  TPG_TRY(tpg_grouped_counts(ctx, v, cp.cls.data(), cp.nclass, &gc));
  OutBuf o;
  TPG_TRY(o.init(out, sizeof(int32_t) * 3 * (size_t)v->m * (size_t)ngroups));
  TPG_LAUNCH(ctx, "grouped_genotype_counts", tpg_grouped_genotype_counts_kernel, dim3(2048), dim3(256), 0,
             (const int32_t*)gc.cnt, gc.Mpad, gc.Cpad, v->m, ngroups, o.dev<int32_t>());
  TPG_CHECK_LAUNCH();
  TPG_HIP(hipStreamSynchronize(ctx->stream));
  return o.commit(ctx);
}

// R/pop_global_stats.R:129-197 (hierfstat::basic.stats arithmetic), one thread per locus; out = m x 10,
// column-major, columns Ho Hs Ht Dst Htp Dstp Fst Fstp Fis Dest.  np / mn as src/compute_np_mn.cpp:8-34 (n is
// never NA there: a group with no typed individual has n = 0, counts in np and makes mn = 0).
#define TPG_GS_NAN __longlong_as_double(0x7FF8000000000000ll)
__global__ void tpg_global_stats_kernel(const int32_t* __restrict__ cnt, int64_t Mpad, int Cpad, int64_t m, int G,
                                        double* __restrict__ out) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
    double np = 0, denom = 0, sho = 0, ssp2 = 0, sfa = 0, sfr = 0;
    int nho = 0, nsp2 = 0;
    for (int g = 0; g < G; g++) {
      const GroupVals gv = tpg_group_vals(cnt, Mpad, Cpad, j, g, 0);
      const double n = gv.valid / 2;
      const double fa = gv.alt / gv.valid, fr = 1 - fa, ho = gv.het2 / gv.valid;
      np += 1.0;
      denom += 1.0 / n;
      if (ho == ho) { sho += ho; nho++; }             // rowMeans(sHo, na.rm = TRUE)
      const double sp2 = fa * fa + fr * fr;
      if (sp2 == sp2) { ssp2 += sp2; nsp2++; }        // rowMeans(sp2, na.rm = TRUE)
      sfa += fa;                                      // rowMeans(freq) without na.rm: NaN propagates
      sfr += fr;
    }
    const double mn = denom > 0.0 ? np / denom : TPG_GS_NAN;
    const double mHo = sho / nho, msp2 = ssp2 / nsp2;  // 0 / 0 = NaN, as mean(numeric(0))
    const double mfa = sfa / G, mfr = sfr / G;
    const double mp2 = mfa * mfa + mfr * mfr;
    const double mHs = mn / (mn - 1) * (1 - msp2 - mHo / 2 / mn);
    const double Ht = 1 - mp2 + mHs / mn / np - mHo / 2 / mn / np;
    const double mFis = 1 - mHo / mHs;
    const double Dst = Ht - mHs;
    const double Dstp = np / (np - 1) * Dst;
    const double Htp = mHs + Dstp;
    const double Fst = Dst / Ht, Fstp = Dstp / Htp, Dest = Dstp / (1 - mHs);
    out[j] = mHo; out[m + j] = mHs; out[2 * m + j] = Ht; out[3 * m + j] = Dst; out[4 * m + j] = Htp;
    out[5 * m + j] = Dstp; out[6 * m + j] = Fst; out[7 * m + j] = Fstp; out[8 * m + j] = mFis; out[9 * m + j] = Dest;
  }
}

// per column: sum and count of the non-NaN (drop_inf: finite) entries, in a fixed order (block partials, then the host)
__global__ __launch_bounds__(256) void tpg_finite_colsum_kernel(const double* __restrict__ x, int64_t m, int drop_inf,
                                                                double* __restrict__ part) {
  __shared__ double ssum[256], scnt[256];
  const int c = blockIdx.y;
  const int64_t chunk = (m + gridDim.x - 1) / gridDim.x;
  const int64_t a = blockIdx.x * chunk, b = a + chunk < m ? a + chunk : m;
  double s = 0, k = 0;
  for (int64_t j = a + threadIdx.x; j < b; j += 256) {
    const double v = x[(int64_t)c * m + j];
    if (v == v && !(drop_inf && fabs(v) == HUGE_VAL)) { s += v; k += 1; }
  }
  ssum[threadIdx.x] = s; scnt[threadIdx.x] = k;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { ssum[threadIdx.x] += ssum[threadIdx.x + o]; scnt[threadIdx.x] += scnt[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[((int64_t)c * gridDim.x + blockIdx.x) * 2] = ssum[0];
    part[((int64_t)c * gridDim.x + blockIdx.x) * 2 + 1] = scnt[0];
  }
}

extern "C" int tpg_pop_global_stats(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups,
                                    const double* ploidy, double* by_locus, double* overall) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && groupIds0 && (by_locus || overall), TPG_EINVAL, "null argument");
  if (ploidy)
    for (int64_t i = 0; i < v->n; i++)  // stopifnot_diploid(.x), R/pop_global_stats.R:117
      TPG_REQUIRE(ploidy[i] == 2.0, TPG_EINVAL, "pop_global_stats only works on diploid data");
  ClassPlan cp;
  TPG_TRY(make_class_plan(v, groupIds0, ngroups, nullptr, &cp));
  GroupedCounts gc;
  TPG_TRY(tpg_grouped_counts(ctx, v, cp.cls.data(), cp.nclass, &gc));
  const int64_t m = v->m;
  OutBuf ob;
  double* d_tmp = nullptr;
  if (by_locus) TPG_TRY(ob.init(by_locus, sizeof(double) * 10 * (size_t)m));
  else TPG_HIP(tpg_pmalloc((void**)&d_tmp, sizeof(double) * 10 * (size_t)m));
  double* d_loc = by_locus ? ob.dev<double>() : d_tmp;
  const int NB = 64;
  double* d_part = nullptr;
  hipError_t e = tpg_pmalloc((void**)&d_part, sizeof(double) * 2 * 10 * NB);
  std::vector<double> part((size_t)2 * 10 * NB);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(tpg_global_stats_kernel, dim3(1024), dim3(256), 0, ctx->stream, (const int32_t*)gc.cnt, gc.Mpad,
                       gc.Cpad, m, ngroups, d_loc);
    if (overall) {
      hipLaunchKernelGGL(tpg_finite_colsum_kernel, dim3(NB, 10), dim3(256), 0, ctx->stream, (const double*)d_loc, m, 1, d_part);
      e = tpg_download(ctx, part.data(), d_part, sizeof(double) * part.size());  // (small: the mailbox) waits for the stream
    } else {
      e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipGetLastError();
  }
  tpg_pfree(d_part);
  tpg_pfree(d_tmp);
  if (e != hipSuccess) { tpg_set_error("pop_global_stats: %s", hipGetErrorString(e)); return TPG_EHIP; }
  if (overall) {
    // is.na(res) <- is.infinite(res); colMeans(res, na.rm = TRUE); then the ratios of means (:203-210)
    for (int c = 0; c < 10; c++) {
      double s = 0, k = 0;
      for (int b = 0; b < NB; b++) { s += part[((size_t)c * NB + b) * 2]; k += part[((size_t)c * NB + b) * 2 + 1]; }
      overall[c] = s / k;
    }
    overall[6] = overall[3] / overall[2];
    overall[7] = overall[5] / overall[4];
    overall[8] = 1 - overall[0] / overall[1];
    overall[9] = overall[5] / (1 - overall[1]);
  }
  if (by_locus) TPG_TRY(ob.commit(ctx));
  return TPG_OK;
}

// Window statistics of per-locus values: the core of windows_stats_generic (R/windows_stats_generic.R:113-176),
// i.e. runner::sum_run / mean_run with na_rm = TRUE over the loci lo[w] .. hi[w]-1 of each window, for every column
// of x (m x ncol, column-major).  One wave per (window, column); lanes stride the window, fixed-order reduction.
// stat = NaN where no value is present (the reference's NA) or where fewer than min_loci are.
__global__ __launch_bounds__(64) void tpg_window_stats_kernel(const double* __restrict__ x, int64_t m,
                                                              const int64_t* __restrict__ lo,
                                                              const int64_t* __restrict__ hi,
                                                              const uint8_t* __restrict__ pad_na, int64_t nw, int op,
                                                              int min_loci, double* __restrict__ stat,
                                                              int32_t* __restrict__ n_loci) {
  const int64_t w = blockIdx.x;
  const int c = blockIdx.y;
  const int lane = threadIdx.x;
  double s = 0;
  int k = 0;
  const bool pad = pad_na && pad_na[w];
  if (!pad)
    for (int64_t j = lo[w] + lane; j < hi[w]; j += 64) {
      const double v = x[j + (int64_t)c * m];
      if (v == v) { s += v; k++; }
    }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); k += __shfl_xor(k, o); }
  if (lane == 0) {
    double r = TPG_GS_NAN;
    if (!pad && k > 0 && k >= min_loci) r = op == 0 ? s / k : s;
    stat[w + (int64_t)c * nw] = r;
    if (n_loci) n_loci[w + (int64_t)c * nw] = pad ? -1 : k;  // -1 = NA (incomplete window with complete = TRUE)
  }
}

extern "C" int tpg_window_stats(tpg_ctx* ctx, const double* x, int64_t m, int ncol, const int64_t* lo,
                                const int64_t* hi, const uint8_t* pad_na, int64_t nw, int op, int min_loci,
                                double* stat, int32_t* n_loci) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && x && lo && hi && stat, TPG_EINVAL, "null argument");
  TPG_REQUIRE(op == 0 || op == 1, TPG_EINVAL, "operator must be 0 (mean) or 1 (sum)");
  TPG_REQUIRE(m >= 0 && ncol >= 1 && ncol <= 65535 && nw >= 0 && nw < 2147483647ll, TPG_EINVAL, "bad sizes");
  if (nw == 0) return TPG_OK;
  for (int64_t w = 0; w < nw; w++)
    TPG_REQUIRE(lo[w] >= 0 && lo[w] <= hi[w] && hi[w] <= m, TPG_EINVAL, "window %lld = [%lld, %lld) outside [0, %lld]",
                (long long)w, (long long)lo[w], (long long)hi[w], (long long)m);
  InBuf ix, il, ih, ip;
  TPG_TRY(ix.init(ctx, x, sizeof(double) * (size_t)m * (size_t)ncol));
  TPG_TRY(il.init(ctx, lo, sizeof(int64_t) * (size_t)nw));
  TPG_TRY(ih.init(ctx, hi, sizeof(int64_t) * (size_t)nw));
  if (pad_na) TPG_TRY(ip.init(ctx, pad_na, (size_t)nw));
  OutBuf os, on;
  TPG_TRY(os.init(stat, sizeof(double) * (size_t)nw * (size_t)ncol));
  if (n_loci) TPG_TRY(on.init(n_loci, sizeof(int32_t) * (size_t)nw * (size_t)ncol));
  TPG_LAUNCH(ctx, "window_stats", tpg_window_stats_kernel, dim3((unsigned)nw, (unsigned)ncol), dim3(64), 0,
             ix.dev<double>(), m, il.dev<int64_t>(), ih.dev<int64_t>(), pad_na ? ip.dev<uint8_t>() : (const uint8_t*)nullptr,
             nw, op, min_loci, os.dev<double>(), on.dev<int32_t>());
  TPG_CHECK_LAUNCH();
  TPG_HIP(hipStreamSynchronize(ctx->stream));
  TPG_TRY(os.commit(ctx));
  if (n_loci) TPG_TRY(on.commit(ctx));
  return TPG_OK;
}

// pop_het_obs / pop_het_exp / pop_fis(method = "Nei87") by locus x population (R/pop_het_obs.R:78-91,
// R/pop_het_exp.R:85-103, R/pop_fis.R:108-115): which 0 = Ho, 1 = Hs, 2 = Fis
__global__ void tpg_pop_basic_kernel(const int32_t* __restrict__ cnt, int64_t Mpad, int Cpad, int64_t m, int G, int which,
                                     double* __restrict__ out) {
  const int64_t total = m * G;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(idx % G);
    const int64_t j = idx / G;
    const GroupVals gv = tpg_group_vals(cnt, Mpad, Cpad, j, g, 0);
    const double sHo = gv.het2 / gv.valid;
    double r = sHo;
    if (which != 0) {
      const double n = gv.valid / 2, fa = gv.alt / gv.valid, fr = 1 - fa;
      const double sp2 = fa * fa + fr * fr;
      double Hs = (1 - sp2 - sHo / 2 / n);
      Hs = n / (n - 1) * Hs;
      r = which == 1 ? Hs : 1 - sHo / Hs;
    }
    out[j + (int64_t)g * m] = r;
  }
}

extern "C" int tpg_pop_basic_stats(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups,
                                   const double* ploidy, int which, double* by_locus, double* colmeans) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && groupIds0 && (by_locus || colmeans), TPG_EINVAL, "null argument");
  TPG_REQUIRE(which >= 0 && which <= 2, TPG_EINVAL, "which must be 0 (Ho), 1 (Hs) or 2 (Fis)");
  if (ploidy)
    for (int64_t i = 0; i < v->n; i++)  // stopifnot_diploid(.x)
      TPG_REQUIRE(ploidy[i] == 2.0, TPG_EINVAL, "this statistic only works on diploid data");
  ClassPlan cp;
  TPG_TRY(make_class_plan(v, groupIds0, ngroups, nullptr, &cp));
  GroupedCounts gc;
  TPG_TRY(tpg_grouped_counts(ctx, v, cp.cls.data(), cp.nclass, &gc));
  const int64_t m = v->m;
  const size_t bytes = sizeof(double) * (size_t)m * (size_t)ngroups;
  OutBuf ob;
  double* d_tmp = nullptr;
  if (by_locus) TPG_TRY(ob.init(by_locus, bytes));
  else TPG_HIP(tpg_pmalloc((void**)&d_tmp, bytes));
  double* d_loc = by_locus ? ob.dev<double>() : d_tmp;
  const int NB = 64;
  double* d_part = nullptr;
  hipError_t e = tpg_pmalloc((void**)&d_part, sizeof(double) * 2 * (size_t)ngroups * NB);
  std::vector<double> part((size_t)2 * ngroups * NB);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(tpg_pop_basic_kernel, dim3(2048), dim3(256), 0, ctx->stream, (const int32_t*)gc.cnt, gc.Mpad, gc.Cpad,
                       m, ngroups, which, d_loc);
    if (colmeans) {
      hipLaunchKernelGGL(tpg_finite_colsum_kernel, dim3(NB, (unsigned)ngroups), dim3(256), 0, ctx->stream,
                         (const double*)d_loc, m, 0, d_part);
      e = tpg_download(ctx, part.data(), d_part, sizeof(double) * part.size());  // (small: the mailbox) waits for the stream
    } else {
      e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipGetLastError();
  }
  tpg_pfree(d_part);
  tpg_pfree(d_tmp);
  if (e != hipSuccess) { tpg_set_error("pop_basic_stats: %s", hipGetErrorString(e)); return TPG_EHIP; }
  if (colmeans)
    for (int g = 0; g < ngroups; g++) {  // colMeans(x, na.rm = TRUE)
      double sm = 0, k = 0;
      for (int b = 0; b < NB; b++) { sm += part[((size_t)g * NB + b) * 2]; k += part[((size_t)g * NB + b) * 2 + 1]; }
      colmeans[g] = sm / k;
    }
  if (by_locus) TPG_TRY(ob.commit(ctx));
  return TPG_OK;
}

// ungrouped diploid: counts (m x 4) -> m x 2 doubles
__global__ void tpg_alt_freq_finalize_kernel(const int4* __restrict__ counts, int64_t m, int as_counts,
                                             double* __restrict__ out) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = counts[j];
    const double alt = (double)(c.y + 2 * c.z), valid = (double)(2 * (c.x + c.y + c.z));
    double f = alt;
    if (!as_counts) f = valid > 0 ? alt / valid : __longlong_as_double(0x7FF8000000000000ll);
    out[j] = f;
    out[m + j] = valid;
  }
}

extern "C" int tpg_alt_freq_dip_pseudo(tpg_ctx* ctx, const tpg_view* v, const double* ploidy, int as_counts,
                                       double* out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && out, TPG_EINVAL, "null argument");
  bool all_dip = true;
  if (ploidy)
    for (int64_t i = 0; i < v->n; i++) {
      TPG_REQUIRE(ploidy[i] == 1.0 || ploidy[i] == 2.0, TPG_EUNSUPPORTED,
                  "ploidy[%lld] = %g: only 1 and 2 are supported", (long long)i, ploidy[i]);
      if (ploidy[i] != 2.0) all_dip = false;
    }
  OutBuf o;
  TPG_TRY(o.init(out, sizeof(double) * 2 * (size_t)v->m));
  if (all_dip) {
    int32_t* d_counts = nullptr;
    TPG_HIP(tpg_pmalloc((void**)&d_counts, sizeof(int32_t) * 4 * (size_t)v->m));
    int rc = tpg_launch_loci_counts(ctx, v, d_counts);
    if (rc == TPG_OK) {
      TPG_LAUNCH(ctx, "alt_freq_finalize", tpg_alt_freq_finalize_kernel, dim3(1024), dim3(256), 0,
                 (const int4*)d_counts, v->m, as_counts, o.dev<double>());
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) { tpg_set_error("alt_freq: %s", hipGetErrorString(e)); rc = TPG_EHIP; }
    }
    tpg_pfree(d_counts);
    TPG_TRY(rc);
    return o.commit(ctx);
  }
  // mixed ploidy: one group, two ploidy classes
  ClassPlan cp;
  TPG_TRY(make_class_plan(v, nullptr, 1, ploidy, &cp));
  GroupedCounts gc;
  TPG_TRY(tpg_grouped_counts(ctx, v, cp.cls.data(), cp.nclass, &gc));
  // reuse the grouped finalize (mode 0 with G = 1 has the m x 2 layout wanted), then NA guard on the host side
  (void)hipFuncSetAttribute((const void*)tpg_grouped_finalize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32 * 65 * 8);
  TPG_LAUNCH(ctx, "grouped_finalize", tpg_grouped_finalize_kernel, dim3((unsigned)ceil_div(v->m, 64)), dim3(256), 2 * 32 * 65 * 8,
             gc.cnt, gc.Mpad, gc.Cpad, v->m, 1, cp.has_hap, 0, as_counts, (const int32_t*)nullptr, o.dev<double>(),
             (double*)nullptr, (double*)nullptr, (double*)nullptr);
  TPG_CHECK_LAUNCH();
  TPG_HIP(hipStreamSynchronize(ctx->stream));
  // 0/0 is already NaN, which is what the ungrouped reference returns as NA_REAL (:52)
  return o.commit(ctx);
}

static int grouped_common(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups,
                          const double* ploidy, int mode, int as_counts, double* o0, size_t o0_count, double* o1,
                          double* o2, double* o3) {
  ClassPlan cp;
  TPG_TRY(make_class_plan(v, groupIds0, ngroups, ploidy, &cp));
  GroupedCounts gc;
  TPG_TRY(tpg_grouped_counts(ctx, v, cp.cls.data(), cp.nclass, &gc));
  const size_t mg = (size_t)v->m * (size_t)ngroups;
  OutBuf b0, b1, b2, b3;
  if (o0) TPG_TRY(b0.init(o0, sizeof(double) * o0_count));
  if (o1) TPG_TRY(b1.init(o1, sizeof(double) * mg));
  if (o2) TPG_TRY(b2.init(o2, sizeof(double) * mg));
  if (o3) TPG_TRY(b3.init(o3, sizeof(double) * mg));
  InBuf gs;
  TPG_TRY(gs.init(ctx, cp.group_size.data(), sizeof(int32_t) * (size_t)ngroups));
  (void)hipFuncSetAttribute((const void*)tpg_grouped_finalize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32 * 65 * 8);
  TPG_LAUNCH(ctx, "grouped_finalize", tpg_grouped_finalize_kernel, dim3((unsigned)ceil_div(v->m, 64)), dim3(256),
             (size_t)(mode == 0 ? 2 : mode == 1 ? 1 : 4) * 32 * 65 * 8,
             gc.cnt, gc.Mpad, gc.Cpad, v->m, ngroups, cp.has_hap, mode, as_counts, gs.dev<int32_t>(), b0.dev<double>(),
             b1.dev<double>(), b2.dev<double>(), b3.dev<double>());
  TPG_CHECK_LAUNCH();  // outputs in device memory are ready in stream order; host outputs are waited for in commit()
  if (o0) TPG_TRY(b0.commit(ctx));
  if (o1) TPG_TRY(b1.commit(ctx));
  if (o2) TPG_TRY(b2.commit(ctx));
  if (o3) TPG_TRY(b3.commit(ctx));
  return TPG_OK;
}

extern "C" int tpg_grouped_alt_freq_dip_pseudo(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0,
                                               int ngroups, const double* ploidy, int as_counts, double* out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && groupIds0 && out, TPG_EINVAL, "null argument");
  return grouped_common(ctx, v, groupIds0, ngroups, ploidy, 0, as_counts, out, (size_t)v->m * 2 * (size_t)ngroups,
                        nullptr, nullptr, nullptr);
}

extern "C" int tpg_grouped_missingness(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0, int ngroups,
                                       double* out) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && groupIds0 && out, TPG_EINVAL, "null argument");
  return grouped_common(ctx, v, groupIds0, ngroups, nullptr, 1, 0, out, (size_t)v->m * (size_t)ngroups, nullptr,
                        nullptr, nullptr);
}

extern "C" int tpg_grouped_summaries_dip_pseudo(tpg_ctx* ctx, const tpg_view* v, const int32_t* groupIds0,
                                                int ngroups, const double* ploidy, double* freq_alt,
                                                double* freq_ref, double* n, double* het_obs) {
  TpgEnter _enter(ctx);
  TPG_REQUIRE(ctx && v && groupIds0, TPG_EINVAL, "null argument");
  return grouped_common(ctx, v, groupIds0, ngroups, ploidy, 2, 0, freq_alt, (size_t)v->m * (size_t)ngroups,
                        freq_ref, n, het_obs);
}
