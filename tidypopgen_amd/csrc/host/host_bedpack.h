// host_bedpack.h -- FBM bytes -> 2 bits per genotype ON THE HOST, in the layout of a PLINK .bed payload (SNP-major, four
// individuals per byte, individual 4 q + r at bits [2 r, 2 r + 1]), for a streamed run that needs ONE code table
// (stream.hip): a quarter of the store's bytes cross PCIe (half of what the nibble pack of host_nibpack.h sends) and the
// device packs its views from them with the .bed front end of the fast pack kernel (pack.hip).
//   out = 2-bit field lut16[in byte] per genotype; lut16 has 16 entries: a byte >= 16 cannot be looked up by pshufb and is
//   reported in the returned OR of all input bytes (the caller then sends that block as bytes).
// tpg_bedpack_col packs ONE column (n bytes -> ceil(n / 4) bytes; the unused bit pairs of the last byte are 0).  A run of
// columns is contiguous in both layouts when nrow is a multiple of 4: the caller packs it with one call.
// Plain C++ (no HIP): also built under -fsanitize=address,undefined by tests/test_host_sanitizers.py.
// (tools/hostpack_probe.cpp: 5 GB out of the page cache -> 1.25 GB in 27 ms with 8 - 16 threads.)
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
__attribute__((target("avx2"))) static inline uint8_t tpg_bedpack_avx2(const uint8_t* in, uint8_t* out, size_t n /* multiple of 128 */,
                                                                         const uint8_t* lut16) {
  const __m128i l = _mm_loadu_si128((const __m128i*)lut16);
  const __m256i lut = _mm256_broadcastsi128_si256(l);
  const __m256i m4 = _mm256_set1_epi16(0x0401), m16 = _mm256_set1_epi16(0x1001);
  __m256i seen = _mm256_setzero_si256();
  for (size_t i = 0; i < n; i += 128) {
    __m256i v[4];
    for (int k = 0; k < 4; k++) {
      const __m256i x = _mm256_loadu_si256((const __m256i*)(in + i + 32 * k));
      seen = _mm256_or_si256(seen, x);
      v[k] = _mm256_shuffle_epi8(lut, x);  // (a byte with its top bit set gives 0; it is reported through `seen`)
    }
    // bytes (c0, c1) of a 16-bit lane -> c0 + 4 c1, lanes to bytes; then (p0, p1) -> p0 + 16 p1: four codes per byte, in order
    const __m256i q0 = _mm256_permute4x64_epi64(_mm256_packus_epi16(_mm256_maddubs_epi16(v[0], m4), _mm256_maddubs_epi16(v[1], m4)), 0xD8);
    const __m256i q1 = _mm256_permute4x64_epi64(_mm256_packus_epi16(_mm256_maddubs_epi16(v[2], m4), _mm256_maddubs_epi16(v[3], m4)), 0xD8);
    _mm256_storeu_si256((__m256i*)(out + i / 4),
                        _mm256_permute4x64_epi64(_mm256_packus_epi16(_mm256_maddubs_epi16(q0, m16), _mm256_maddubs_epi16(q1, m16)), 0xD8));
  }
  uint8_t s[32];
  _mm256_storeu_si256((__m256i*)s, seen);
  uint8_t r = 0;
  for (int k = 0; k < 32; k++) r |= s[k];
  return r;
}
#endif

static inline uint8_t tpg_bedpack_scalar(const uint8_t* in, uint8_t* out, size_t n, const uint8_t* lut16) {
  uint8_t seen = 0;
  for (size_t i = 0; i < n; i += 4) {
    uint8_t b = 0;
    for (size_t r = 0; r < 4 && i + r < n; r++) {
      seen |= in[i + r];
      b |= (uint8_t)((lut16[in[i + r] & 15] & 3) << (2 * r));
    }
    out[i / 4] = b;
  }
  return seen;
}

// n bytes of genotypes -> ceil(n / 4) bytes; returns the OR of the input bytes (>= 16: the packed bytes must not be used)
static inline uint8_t tpg_bedpack(const uint8_t* in, uint8_t* out, size_t n, const uint8_t* lut16) {
  size_t done = 0;
  uint8_t seen = 0;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_cpu_supports("avx2")) {
    done = n / 128 * 128;
    seen = tpg_bedpack_avx2(in, out, done, lut16);
  }
#endif
  return (uint8_t)(seen | tpg_bedpack_scalar(in + done, out + done / 4, n - done, lut16));
}
