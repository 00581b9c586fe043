// host_nibpack.h -- FBM bytes -> 4 bits per genotype ON THE HOST, inside the upload team (runtime.hip: tpg_upload): a
// bigstatsr FBM.code256 of genotypes holds bytes 0 .. 2 (dosages), 3 (missing) and, once imputed, 4 .. 6; two of them fit a
// byte, so half the bytes cross PCIe and a trivial kernel (tpg_nib_expand_kernel) writes the byte FBM the pack kernels
// read.  out[i] = in[2 i] | in[2 i + 1] << 4; returns the OR of all input bytes, so that the caller sees a byte >= 16
// (a dosage table, arbitrary user bytes) and sends that chunk as it is.  Plain C++ (no HIP): built under
// -fsanitize=address,undefined by tests/test_host_sanitizers.py too.  What the host gives (tools/hostpack_probe.cpp, the
// box of a one-GPU job, 5 GB out of the page cache): 38 ms with 8 - 16 threads (140 GB/s of input) -- against 90 - 160 ms for
// the 5 GB over PCIe.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
__attribute__((target("avx2"))) static inline uint8_t tpg_nibpack_avx2(const uint8_t* in, uint8_t* out, size_t n /* multiple of 64 */) {
  const __m256i mul = _mm256_set1_epi16(0x1001);
  __m256i seen = _mm256_setzero_si256();
  for (size_t i = 0; i < n; i += 64) {
    const __m256i a = _mm256_loadu_si256((const __m256i*)(in + i)), b = _mm256_loadu_si256((const __m256i*)(in + i + 32));
    seen = _mm256_or_si256(seen, _mm256_or_si256(a, b));
    // bytes (x0, x1) of a 16-bit lane -> x0 * 1 + x1 * 16 (<= 255 while both are < 16), then the 16-bit lanes to bytes
    const __m256i pa = _mm256_maddubs_epi16(a, mul), pb = _mm256_maddubs_epi16(b, mul);
    _mm256_storeu_si256((__m256i*)(out + i / 2), _mm256_permute4x64_epi64(_mm256_packus_epi16(pa, pb), 0xD8));
  }
  uint8_t s[32];
  _mm256_storeu_si256((__m256i*)s, seen);
  uint8_t r = 0;
  for (int k = 0; k < 32; k++) r |= s[k];
  return r;
}
#endif

static inline uint8_t tpg_nibpack_scalar(const uint8_t* in, uint8_t* out, size_t n /* even */) {
  uint8_t seen = 0;
  for (size_t i = 0; i < n; i += 2) {
    seen |= (uint8_t)(in[i] | in[i + 1]);
    out[i / 2] = (uint8_t)((in[i] & 15) | (in[i + 1] << 4));
  }
  return seen;
}

// n even; returns the OR of the input bytes (a result >= 16 means the packed bytes must not be used)
static inline uint8_t tpg_nibpack(const uint8_t* in, uint8_t* out, size_t n) {
  size_t done = 0;
  uint8_t seen = 0;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_cpu_supports("avx2")) {
    done = n / 64 * 64;
    seen = tpg_nibpack_avx2(in, out, done);
  }
#endif
  return (uint8_t)(seen | tpg_nibpack_scalar(in + done, out + done / 2, n - done));
}
