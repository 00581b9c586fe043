// host_bits2.h -- the register-level transposition of tpg_gcls_gather_kernel: sixteen dwords of sixteen 2-bit fields each,
// rows and columns exchanged in 96 instructions (two rounds of byte permutes, two of masked shifts).  Compiled for the
// device by gramcls.hip and, with the one builtin spelled out, for the host by tests/host/host_pieces.cpp.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define TPG_BITS_FN __host__ __device__ __forceinline__
#else
#define TPG_BITS_FN static inline
#endif

// bytes of the result picked by the selector bytes: 0 .. 3 = bytes of lo, 4 .. 7 = bytes of hi (v_perm_b32)
TPG_BITS_FN uint32_t tpg_byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_amdgcn_perm(hi, lo, sel);
#else
  const uint64_t v = ((uint64_t)hi << 32) | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) r |= (uint32_t)((v >> (8 * ((sel >> (8 * i)) & 7u))) & 0xFFu) << (8 * i);
  return r;
#endif
}

// W[k] field p (bits 2 p, 2 p + 1)  ->  W[p] field k, for k, p = 0 .. 15
TPG_BITS_FN void tpg_transpose16_2bit(uint32_t (&W)[16]) {
  // bytes: word 4 a + c, byte b  ->  word 4 b + c, byte a
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const uint32_t x0 = W[c], x1 = W[4 + c], x2 = W[8 + c], x3 = W[12 + c];
    const uint32_t l01 = tpg_byte_perm(x1, x0, 0x05010400u), h01 = tpg_byte_perm(x1, x0, 0x07030602u);
    const uint32_t l23 = tpg_byte_perm(x3, x2, 0x05010400u), h23 = tpg_byte_perm(x3, x2, 0x07030602u);
    W[c] = tpg_byte_perm(l23, l01, 0x05040100u);
    W[4 + c] = tpg_byte_perm(l23, l01, 0x07060302u);
    W[8 + c] = tpg_byte_perm(h23, h01, 0x05040100u);
    W[12 + c] = tpg_byte_perm(h23, h01, 0x07060302u);
  }
  // inside every byte: word 4 b + c, field f  ->  word 4 b + f, field c (v_lshlrev / v_lshrrev + v_bfi_b32 per word and round)
#pragma unroll
  for (int b = 0; b < 4; b++) {
    uint32_t* U = W + 4 * b;
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const uint32_t lo = (U[c] & 0x0F0F0F0Fu) | ((U[c + 2] << 4) & 0xF0F0F0F0u);
      const uint32_t hi = ((U[c] >> 4) & 0x0F0F0F0Fu) | (U[c + 2] & 0xF0F0F0F0u);
      U[c] = lo;
      U[c + 2] = hi;
    }
#pragma unroll
    for (int x = 0; x < 4; x += 2) {
      const uint32_t lo = (U[x] & 0x33333333u) | ((U[x + 1] << 2) & 0xCCCCCCCCu);
      const uint32_t hi = ((U[x] >> 2) & 0x33333333u) | (U[x + 1] & 0xCCCCCCCCu);
      U[x] = lo;
      U[x + 1] = hi;
    }
  }
}
