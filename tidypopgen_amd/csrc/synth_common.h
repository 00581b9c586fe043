// synth_common.h -- deterministic synthetic genotype panel.
//
// Integer-only, counter-based: every byte of the panel is a pure function of
// (seed, individual i, locus j), so any shard can be generated independently
// on the host (oracle/tpg_oracle.c: orc_synth_fbm) or on the device
// (synth.hip) and the two agree bit for bit.  No floating point is used, so
// there is no libm / FMA difference between host and device.
//
// Model: ancestral frequency p_j ~ U(0.05, 0.95); population frequency
// p_jg = clamp(p_j + sd_jg * z_jg, 0.001, 0.999) with sd_jg = sqrt(F_g p_j (1-p_j)),
// population-specific drift F_g rising linearly from 0.01 (g = 0) to 0.20 (g = npop-1) -- equal F for
// all populations would give npop-1 nearly equal leading eigenvalues, a degenerate spectrum that real
// panels (HGDP-like hierarchical structure) do not have -- and z_jg an Irwin-Hall(12) approximation of N(0,1) -- the normal
// approximation of the Balding-Nichols Beta with the same mean and variance
// (documented deviation from the Beta draw named in SURVEY.md §8d: a Beta
// sampler needs floating-point transcendental functions, which would break the
// host/device bit-for-bit property).  Genotype ~ Binomial(2, p_jg); individuals
// are assigned to populations round-robin (g = i % npop); a genotype is missing
// with probability miss_thresh / 2^32.  Missing entries are written as FBM byte
// 3 (NA under CODE_012), or -- when imputed_bytes != 0 -- as 4 + genotype, i.e.
// the bytes bigsnpr's imputation writes and CODE_IMPUTE_PRED decodes back to
// 0/1/2 (R/gt_has_imputed.R:101-106).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define TPG_HD __host__ __device__ static inline
#else
#define TPG_HD static inline
#endif

TPG_HD uint64_t tpg_mix64(uint64_t x) {  // splitmix64 finalizer
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

TPG_HD uint64_t tpg_isqrt64(uint64_t v) {  // floor(sqrt(v)), bitwise (no fp)
  uint64_t r = 0, bit = 1ull << 62;
  while (bit > v) bit >>= 2;
  while (bit) {
    if (v >= r + bit) { v -= r + bit; r = (r >> 1) + bit; }
    else r >>= 1;
    bit >>= 2;
  }
  return r;
}

// population allele frequency of locus j in population g, Q32 fixed point
TPG_HD uint32_t tpg_synth_pjg(uint64_t seed, uint64_t j, uint32_t g, uint32_t npop) {
  const uint64_t LO = 214748365ull;    // 0.05 * 2^32
  const uint64_t RANGE = 3865470566ull;  // 0.90 * 2^32
  const uint64_t FMIN = 42949673ull, FSPAN = 816043786ull;  // 0.01, 0.19 (* 2^32)
  const uint64_t F = FMIN + (npop > 1 ? (FSPAN * (uint64_t)g) / (uint64_t)(npop - 1) : 0);
  uint64_t hj = tpg_mix64(seed ^ tpg_mix64(j * 2 + 1));
  uint64_t p = LO + (((hj >> 32) * RANGE) >> 32);          // Q32
  uint64_t pq = (p * (4294967296ull - p)) >> 32;           // Q32
  uint64_t var = (pq * F) >> 32;                           // Q32
  uint64_t sd = tpg_isqrt64(var << 32);                    // Q32
  // Irwin-Hall(12) from three 64-bit hashes (12 x 16-bit uniforms)
  int64_t s = 0;
  for (int t = 0; t < 3; t++) {
    uint64_t h = tpg_mix64(hj ^ tpg_mix64(((uint64_t)g << 8) + (uint64_t)t + 0x5151ull));
    s += (int64_t)(h & 0xFFFF) + (int64_t)((h >> 16) & 0xFFFF) + (int64_t)((h >> 32) & 0xFFFF) + (int64_t)(h >> 48);
  }
  s -= 393210;  // 6 * 65535
  int64_t pg = (int64_t)p + (((int64_t)sd * s) >> 16);
  const int64_t PMIN = 4294967, PMAX = 4290672329ll;  // 0.001, 0.999
  if (pg < PMIN) pg = PMIN;
  if (pg > PMAX) pg = PMAX;
  return (uint32_t)pg;
}

// FBM byte of individual i at locus j given its population's frequency
TPG_HD uint8_t tpg_synth_geno(uint64_t seed, uint64_t i, uint64_t j, uint32_t pjg,
                              uint32_t miss_thresh, int imputed_bytes) {
  uint64_t h = tpg_mix64(tpg_mix64(seed + 0xA5A5A5A5ull + j) ^ (i * 0xD1B54A32D192ED03ull));
  uint32_t u1 = (uint32_t)h, u2 = (uint32_t)(h >> 32);
  uint8_t g = (uint8_t)((u1 < pjg) + (u2 < pjg));
  uint32_t u3 = (uint32_t)(tpg_mix64(h ^ 0x1234567ull) >> 32);
  if (u3 < miss_thresh) return imputed_bytes ? (uint8_t)(4 + g) : (uint8_t)3;
  return g;
}

TPG_HD uint8_t tpg_synth_byte(uint64_t seed, uint64_t i, uint64_t j, int npop, uint32_t miss_thresh,
                              int imputed_bytes) {
  uint32_t pjg = tpg_synth_pjg(seed, j, (uint32_t)(i % (uint64_t)npop), (uint32_t)npop);
  return tpg_synth_geno(seed, i, j, pjg, miss_thresh, imputed_bytes);
}
