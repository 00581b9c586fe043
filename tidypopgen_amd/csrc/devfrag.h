// devfrag.h -- device-side helpers for the 2-bit fragment layout (see common.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef double v4d __attribute__((ext_vector_type(4)));

// v_perm_b32 used as a 4-entry byte lookup: selector bytes 0..3 pick bytes of `lut`.
// LUT byte c = value for code c (0,1,2 = dosage, 3 = missing).
#define TPG_LUT_V 0x00010101u   // valid:        1,1,1,0
#define TPG_LUT_H 0x00000100u   // heterozygous: 0,1,0,0
#define TPG_LUT_E2 0x00010000u  // hom alt:      0,0,1,0
#define TPG_LUT_D 0x000100FFu   // dosage - 1:  -1,0,1,0
#define TPG_LUT_G 0x00020100u   // dosage:       0,1,2,0

__device__ __forceinline__ uint32_t tpg_codes(uint32_t w, int k) { return (w >> (2 * k)) & 0x03030303u; }

__device__ __forceinline__ int tpg_lut(uint32_t lut, uint32_t codes) {
  return (int)__builtin_amdgcn_perm(0u, lut, codes);
}

// FP4 operand nibble of the pairwise kernel (pairwise.hip): one magnitude bit per operand plane -- 0x1 = FP4 0.5, 0x2 = 1.0,
// 0x4 = 2.0 -- for heterozygous (h), typed (v) and homozygous (d), bit 3 = the sign of d (dosage 0).  Which plane takes which
// magnitude does not change a single sum (the E8M0 block scales undo it: TPG_T4_SC_*), only the bit patterns the matrix cores
// multiply -- and their clock under the MFMAs is a POWER limit that depends on those (tools/pw_power_probe.py), so the
// assignment is a compile-time choice that was measured (TPG_T4_ENC, tools/enc_ab.py): 0 = h 0.5, v 1, d 2 (rounds 2 - 4).
#ifndef TPG_T4_ENC
#define TPG_T4_ENC 0
#endif
#if TPG_T4_ENC == 0
#define TPG_T4_MH 1u
#define TPG_T4_MV 2u
#define TPG_T4_MD 4u
#elif TPG_T4_ENC == 1
#define TPG_T4_MH 1u
#define TPG_T4_MV 4u
#define TPG_T4_MD 2u
#elif TPG_T4_ENC == 2
#define TPG_T4_MH 2u
#define TPG_T4_MV 1u
#define TPG_T4_MD 4u
#elif TPG_T4_ENC == 3
#define TPG_T4_MH 2u
#define TPG_T4_MV 4u
#define TPG_T4_MD 1u
#elif TPG_T4_ENC == 4
#define TPG_T4_MH 4u
#define TPG_T4_MV 1u
#define TPG_T4_MD 2u
#else
#define TPG_T4_MH 4u
#define TPG_T4_MV 2u
#define TPG_T4_MD 1u
#endif
// 2-bit code -> nibble: dosage 0 -> v | d | sign, 1 -> v | h, 2 -> v | d, missing -> 0  (0x0006030E for TPG_T4_ENC = 0)
#ifndef TPG_T4_VMISS
#define TPG_NIB_LUT ((TPG_T4_MV | TPG_T4_MD | 8u) | ((TPG_T4_MV | TPG_T4_MH) << 8) | ((TPG_T4_MV | TPG_T4_MD) << 16))
#else
// TIMING EXPERIMENT (round 6, wrong sums): the "v" bit marks the MISSING genotypes instead of the typed ones, i.e. the plane
// that three of the five products multiply is 2 % ones instead of 98 % -- what the complement form V = L - m_i - m_j + MM,
// HV = H_ii - HM would feed the matrix cores -- to see what the power-limited clock does with it (DESIGN.md 3.1 "Round 6")
#define TPG_NIB_LUT ((TPG_T4_MD | 8u) | ((TPG_T4_MH) << 8) | ((TPG_T4_MD) << 16) | ((TPG_T4_MV) << 24))
#endif
// E8M0 block scale (all four bytes equal) that turns a plane of magnitude bit M into 0 / +-1: 0.5 x 2, 1 x 1, 2 x 0.5
#define TPG_T4_SC(M) ((M) == 1u ? (int)0x80808080 : (M) == 2u ? 0x7f7f7f7f : 0x7e7e7e7e)
// one T dword (16 codes) -> two T4 dwords (16 nibbles)
__device__ __forceinline__ void tpg_t4_words(uint32_t w, uint32_t& lo, uint32_t& hi) {
  uint32_t nb[4];
#pragma unroll
  for (int k = 0; k < 4; k++) nb[k] = (uint32_t)tpg_lut(TPG_NIB_LUT, tpg_codes(w, k));
  lo = nb[0] | (nb[1] << 4);
  hi = nb[2] | (nb[3] << 4);
}

// bit position of element e (0..15) inside a packed dword
__host__ __device__ __forceinline__ int tpg_elem_shift(int e) { return 8 * (e & 3) + 2 * (e >> 2); }

// MFMA 32x32 C/D register -> row inside the 32x32 tile (col = lane & 31)
__device__ __forceinline__ int tpg_cd_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() also waits for every global load in
// flight (s_waitcnt vmcnt(0)), i.e. for the prefetch a K loop has just issued for its next iteration: a full memory
// latency per iteration.  The workgroup-scope fences order the LDS accesses (s_waitcnt lgkmcnt(0)) and leave vmcnt alone.
__device__ __forceinline__ void tpg_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
