// host_addcounts.h -- K[k] += count[k] ON THE HOST: what the literal increment_* mirrors do to the caller's N x N double
// matrices once per block of an R driver loop (pairwise.hip add_counts_to_caller; src/snp_ibs.cpp:67-72, src/snp_king.cpp:70-72,
// src/snp_as.cpp:64-65 add FP64 products there).  The counts arrive as uint16 (+ bias for a signed one) or int32; every sum is
// an integer below 2^53, so the order of the additions does not matter and the result equals the reference's.
// Plain C++ (no HIP): also built under -fsanitize=address,undefined by tests/test_host_sanitizers.py.
// (tools/host_rmw_probe.cpp: 200 MB of doubles + 50 MB of counts on 16 threads of the pool's hosts: 1.0-1.4 ms with AVX2,
// 1.7-1.9 ms with the scalar loop.)
#pragma once
#include <stddef.h>
#include <stdint.h>

static inline void tpg_add_counts_u16_scalar(double* dst, const uint16_t* q, size_t n, int bias) {
  for (size_t k = 0; k < n; k++) dst[k] += (double)((int)q[k] - bias);
}
static inline void tpg_add_counts_i32_scalar(double* dst, const int32_t* q, size_t n) {
  for (size_t k = 0; k < n; k++) dst[k] += (double)q[k];
}

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
__attribute__((target("avx2"))) static inline void tpg_add_counts_u16_avx2(double* dst, const uint16_t* q, size_t n /* multiple of 16 */,
                                                                             int bias) {
  const __m256i b = _mm256_set1_epi32(bias);
  for (size_t k = 0; k < n; k += 16) {
    const __m256i w = _mm256_loadu_si256((const __m256i*)(q + k));
    const __m256i lo = _mm256_sub_epi32(_mm256_cvtepu16_epi32(_mm256_castsi256_si128(w)), b);
    const __m256i hi = _mm256_sub_epi32(_mm256_cvtepu16_epi32(_mm256_extracti128_si256(w, 1)), b);
    _mm256_storeu_pd(dst + k, _mm256_add_pd(_mm256_loadu_pd(dst + k), _mm256_cvtepi32_pd(_mm256_castsi256_si128(lo))));
    _mm256_storeu_pd(dst + k + 4, _mm256_add_pd(_mm256_loadu_pd(dst + k + 4), _mm256_cvtepi32_pd(_mm256_extracti128_si256(lo, 1))));
    _mm256_storeu_pd(dst + k + 8, _mm256_add_pd(_mm256_loadu_pd(dst + k + 8), _mm256_cvtepi32_pd(_mm256_castsi256_si128(hi))));
    _mm256_storeu_pd(dst + k + 12, _mm256_add_pd(_mm256_loadu_pd(dst + k + 12), _mm256_cvtepi32_pd(_mm256_extracti128_si256(hi, 1))));
  }
}
__attribute__((target("avx2"))) static inline void tpg_add_counts_i32_avx2(double* dst, const int32_t* q, size_t n /* multiple of 8 */) {
  for (size_t k = 0; k < n; k += 8) {
    const __m256i w = _mm256_loadu_si256((const __m256i*)(q + k));
    _mm256_storeu_pd(dst + k, _mm256_add_pd(_mm256_loadu_pd(dst + k), _mm256_cvtepi32_pd(_mm256_castsi256_si128(w))));
    _mm256_storeu_pd(dst + k + 4, _mm256_add_pd(_mm256_loadu_pd(dst + k + 4), _mm256_cvtepi32_pd(_mm256_extracti128_si256(w, 1))));
  }
}
#endif

// dst[k] += q[k] - bias, k < n
static inline void tpg_add_counts_u16(double* dst, const uint16_t* q, size_t n, int bias) {
  size_t done = 0;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_cpu_supports("avx2")) {
    done = n / 16 * 16;
    tpg_add_counts_u16_avx2(dst, q, done, bias);
  }
#endif
  tpg_add_counts_u16_scalar(dst + done, q + done, n - done, bias);
}
static inline void tpg_add_counts_i32(double* dst, const int32_t* q, size_t n) {
  size_t done = 0;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_cpu_supports("avx2")) {
    done = n / 8 * 8;
    tpg_add_counts_i32_avx2(dst, q, done);
  }
#endif
  tpg_add_counts_i32_scalar(dst + done, q + done, n - done);
}
