// host_relfilter.h -- the sequential, data-dependent loop of filter_high_relatedness (R/filter_high_relatedness.R:26-145) on
// a host copy of the relatedness matrix (pairwise.hip fetches it from HBM and calls this).  Plain C++ (no HIP):
// tests/test_host_sanitizers.py builds it with -fsanitize=address,undefined and compares its decisions with a literal
// restatement of the R loop.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <string>
#include <vector>

#define TPG_HOST_REQUIRE(cond, code, ...)          \
  do {                                             \
    if (!(cond)) {                                 \
      char _b[512];                                \
      snprintf(_b, sizeof(_b), __VA_ARGS__);       \
      err = _b;                                    \
      return (code);                               \
    }                                              \
  } while (0)

static double r_mean_ld(const std::vector<double>& x) {  // R's mean() of the non-NA values (summary.c, recalled)
  const size_t n = x.size();
  if (n == 0) return NAN;
  long double s = 0;
  for (double v : x) s += v;
  s /= (long double)n;
  long double t = 0;
  for (double v : x) t += (long double)v - s;
  s += t / (long double)n;
  return (double)s;
}

// A: n x n column-major relatedness matrix (consumed); keep[n], new_order0[n] (may be NULL) as tpg_filter_high_relatedness
// documents them; returns 0, or 1 with the R error message in `err`
static inline int tpg_host_filter_high_relatedness(std::vector<double>& A, int64_t n, double kings_threshold, uint8_t* keep,
                                                   int32_t* new_order0, std::string& err) {
  const size_t N = (size_t)n;
  if (n == 1) {  // :46-52
    keep[0] = 1;
    if (new_order0) new_order0[0] = 0;
    return 0;
  }
  for (double& x : A) x = fabs(x);  // :55 (NaN stays NaN)
  // :69-73 column means without the diagonal, R's mean(); order(decreasing = TRUE) is stable and puts NA last
  std::vector<double> cm(N);
  {
    std::vector<double> col;
    col.reserve(N);
    for (size_t j = 0; j < N; j++) {
      col.clear();
      for (size_t i = 0; i < N; i++)
        if (i != j && A[i + j * N] == A[i + j * N]) col.push_back(A[i + j * N]);
      cm[j] = r_mean_ld(col);
    }
  }
  std::vector<int32_t> ord(N);
  for (size_t j = 0; j < N; j++) ord[j] = (int32_t)j;
  std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
    const double x = cm[(size_t)a], y = cm[(size_t)b];
    const bool xn = x != x, yn = y != y;
    if (xn || yn) return !xn && yn;  // NA last
    return x > y;
  });
  std::vector<double> M(N * N);  // :76 matrix[order, order]
  for (size_t b = 0; b < N; b++)
    for (size_t a = 0; a < N; a++) M[a + b * N] = A[(size_t)ord[a] + (size_t)ord[b] * N];
  std::vector<double>().swap(A);
  // matrix2 = M with the diagonal (and the rows / columns of deleted individuals) set to NA, kept implicitly
  std::vector<uint8_t> alive(N, 1);
  std::vector<long double> rs(N, 0);
  std::vector<int64_t> rc(N, 0);
  long double S = 0;
  int64_t C = 0, above = 0;
  for (size_t l = 0; l < N; l++)
    for (size_t k = 0; k < N; k++) {
      const double x = M[k + l * N];
      if (k == l || x != x) continue;
      rs[k] += x;
      rc[k]++;
      if (x > kings_threshold) above++;
    }
  for (size_t k = 0; k < N; k++) { S += rs[k]; C += rc[k]; }
  auto in_m2 = [&](size_t k, size_t l) { return k != l && alive[k] && alive[l] && M[k + l * N] == M[k + l * N]; };
  std::vector<double> tmp;
  for (size_t i = 0; i + 1 < N; i++) {  // :90
    if (above == 0) break;              // :91-96 !any(matrix2 > threshold)
    if (!alive[i]) continue;            // :97
    for (size_t j = i + 1; j < N; j++) {
      if (!alive[i] || !alive[j]) continue;  // :101
      const double x = M[i + j * N];
      TPG_HOST_REQUIRE(x == x, 1, "missing value where TRUE/FALSE needed (relatedness of individuals %d and %d is NA)",
                  ord[i] + 1, ord[j] + 1);
      if (!(x > kings_threshold)) continue;  // :102
      TPG_HOST_REQUIRE(rc[i] > 0 && C - rc[j] > 0, 1, "missing value where TRUE/FALSE needed (empty mean)");
      double mn1 = (double)(rs[i] / (long double)rc[i]);
      double mn2 = (double)((S - rs[j]) / (long double)(C - rc[j]));
      const double scale = std::max(fabs(mn1), fabs(mn2));
      if (fabs(mn1 - mn2) <= 1e-12 * scale) {  // too close to call from running sums: R's own arithmetic, literally
        tmp.clear();
        for (size_t l = 0; l < N; l++)
          if (in_m2(i, l)) tmp.push_back(M[i + l * N]);  // mean(matrix2[i, ], na.rm = TRUE), :103
        mn1 = r_mean_ld(tmp);
        tmp.clear();
        for (size_t l = 0; l < N; l++)  // mean(matrix2[-j, ], na.rm = TRUE), :104: column-major walk without row j
          for (size_t k = 0; k < N; k++)
            if (k != j && in_m2(k, l)) tmp.push_back(M[k + l * N]);
        mn2 = r_mean_ld(tmp);
      }
      const size_t d = mn1 > mn2 ? i : j;  // :119-131
      // matrix2[d, ] <- NA; matrix2[, d] <- NA
      for (size_t k = 0; k < N; k++) {
        if (!in_m2(k, d)) continue;
        const double c1 = M[k + d * N];  // column d, row k
        rs[k] -= c1; rc[k]--;
        S -= c1; C--;
        if (c1 > kings_threshold) above--;
      }
      for (size_t l = 0; l < N; l++)
        if (in_m2(d, l) && M[d + l * N] > kings_threshold) above--;  // row d
      S -= rs[d]; C -= rc[d];
      rs[d] = 0; rc[d] = 0;
      alive[d] = 0;
    }
  }
  for (size_t k = 0; k < N; k++) {
    keep[(size_t)ord[k]] = alive[k];
    if (new_order0) new_order0[k] = ord[k];
  }
  return 0;
}
