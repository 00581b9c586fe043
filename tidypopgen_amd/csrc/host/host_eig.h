// host_eig.h -- the b x b (b <= 64) host linear algebra of the eigen solver (pca.hip): Cholesky for CholQR2, the inverse of
// an upper triangle, and the symmetric eigen-decomposition of the Rayleigh-Ritz matrix (Householder tridiagonalisation +
// implicit QL).  Plain C++ with no HIP in it, so that tests/test_host_sanitizers.py can build it with
// -fsanitize=address,undefined and check it without a GPU (the reference checks its native code under valgrind,
// .github/workflows/R-CMD-check-valgrind.yaml:50-51).
#pragma once
#include <math.h>
#include <stddef.h>

#include <algorithm>
#include <vector>

// ---- small host linear algebra (b <= 64) ----
static bool host_cholesky_upper(std::vector<double>& G, int b) {  // G = R'R, R upper, in place (column-major)
  for (int j = 0; j < b; j++) {
    double s = G[j + (size_t)j * b];
    for (int k = 0; k < j; k++) s -= G[k + (size_t)j * b] * G[k + (size_t)j * b];
    if (!(s > 0)) return false;
    const double rjj = sqrt(s);
    G[j + (size_t)j * b] = rjj;
    for (int c = j + 1; c < b; c++) {
      double t = G[j + (size_t)c * b];
      for (int k = 0; k < j; k++) t -= G[k + (size_t)j * b] * G[k + (size_t)c * b];
      G[j + (size_t)c * b] = t / rjj;
    }
    for (int i = j + 1; i < b; i++) G[i + (size_t)j * b] = 0;
  }
  return true;
}

static void host_upper_inverse(const std::vector<double>& R, int b, std::vector<double>& Ri) {
  Ri.assign((size_t)b * b, 0.0);
  for (int j = 0; j < b; j++) {
    Ri[j + (size_t)j * b] = 1.0 / R[j + (size_t)j * b];
    for (int i = j - 1; i >= 0; i--) {
      double s = 0;
      for (int k = i + 1; k <= j; k++) s += R[i + (size_t)k * b] * Ri[k + (size_t)j * b];
      Ri[i + (size_t)j * b] = -s / R[i + (size_t)i * b];
    }
  }
}

// Symmetric eigen-decomposition of a small dense matrix: Householder reduction to tridiagonal form with the
// transformations accumulated, then implicit-shift QL (the classic tred2 / tql2 pair).  ~4/3 n^3 + ~3 n^3 flops
// instead of the ~16 n^3 of eight cyclic Jacobi sweeps: the Rayleigh-Ritz matrices here are up to 64 x 64 and
// sit on the critical path between two launches.  H is column-major symmetric; eigenvalues come back in
// descending order with the eigenvectors as the columns of X (column-major).
//
// The working matrix is column-major and every inner loop runs down a column, so the host compiler vectorises them (an
// AVX2 clone is picked at load time where the CPU has it); the 2 x 2 rotations use sqrt(p^2 + e^2), the caller having
// scaled the matrix to entries of at most 1 (hypot() alone was 40 % of the time).
#define V(i, j) Vp[(size_t)(j) * n + (i)]
#if !defined(__HIP_DEVICE_COMPILE__) && !defined(TPG_HOST_NO_CLONES)
__attribute__((target_clones("arch=x86-64-v3", "default")))
#endif
static void sym_eig_core(double* __restrict Vp, double* __restrict d, double* __restrict e, int n, bool vectors) {
  // --- Householder tridiagonalisation, last row first
  for (int j = 0; j < n; j++) d[j] = V(n - 1, j);
  for (int i = n - 1; i > 0; i--) {
    double scale = 0, h = 0;
    for (int k = 0; k < i; k++) scale += fabs(d[k]);
    if (scale == 0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; j++) { d[j] = V(i - 1, j); V(i, j) = 0; V(j, i) = 0; }
    } else {
      for (int k = 0; k < i; k++) { d[k] /= scale; h += d[k] * d[k]; }
      double f = d[i - 1], g = sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; j++) e[j] = 0;
      for (int j = 0; j < i; j++) {  // e = A u (lower triangle only)
        f = d[j];
        V(j, i) = f;
        g = e[j] + V(j, j) * f;
        for (int k = j + 1; k < i; k++) { g += V(k, j) * d[k]; e[k] += V(k, j) * f; }
        e[j] = g;
      }
      f = 0;
      for (int j = 0; j < i; j++) { e[j] /= h; f += e[j] * d[j]; }
      const double hh = f / (h + h);
      for (int j = 0; j < i; j++) e[j] -= hh * d[j];
      for (int j = 0; j < i; j++) {  // rank-2 update of the leading block
        f = d[j]; g = e[j];
        for (int k = j; k < i; k++) V(k, j) -= f * e[k] + g * d[k];
        d[j] = V(i - 1, j);
        V(i, j) = 0;
      }
    }
    d[i] = h;
  }
  if (!vectors) {  // eigenvalues only: the diagonal of the tridiagonal matrix sits on the diagonal of V; no accumulation
    for (int i = 0; i < n; i++) d[i] = V(i, i);
  } else {
  for (int i = 0; i < n - 1; i++) {  // accumulate the reflectors
    V(n - 1, i) = V(i, i);
    V(i, i) = 1;
    const double h = d[i + 1];
    if (h != 0) {
      for (int k = 0; k <= i; k++) d[k] = V(k, i + 1) / h;
      for (int j = 0; j <= i; j++) {
        double g = 0;
        for (int k = 0; k <= i; k++) g += V(k, i + 1) * V(k, j);
        for (int k = 0; k <= i; k++) V(k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; k++) V(k, i + 1) = 0;
  }
  for (int j = 0; j < n; j++) { d[j] = V(n - 1, j); V(n - 1, j) = 0; }
  V(n - 1, n - 1) = 1;
  }
  e[0] = 0;
  // --- implicit QL on the tridiagonal (d, e)
  for (int i = 1; i < n; i++) e[i - 1] = e[i];
  e[n - 1] = 0;
  double f = 0, tst1 = 0;
  const double eps = 2.220446049250313e-16;
  for (int l = 0; l < n; l++) {
    tst1 = fmax(tst1, fabs(d[l]) + fabs(e[l]));
    int m = l;
    while (m < n - 1 && fabs(e[m]) > eps * tst1) m++;
    if (m > l) {
      int iter = 0;
      do {
        double g = d[l];
        double p = (d[l + 1] - g) / (2 * e[l]);
        double r = sqrt(p * p + 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; i++) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1, c2 = 1, c3 = 1, s = 0, s2 = 0;
        const double el1 = e[l + 1];
        for (int i = m - 1; i >= l; i--) {
          c3 = c2; c2 = c; s2 = s;
          g = c * e[i];
          h = c * p;
          r = sqrt(p * p + e[i] * e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          if (vectors)
            for (int k = 0; k < n; k++) {
              h = V(k, i + 1);
              V(k, i + 1) = s * V(k, i) + c * h;
              V(k, i) = c * V(k, i) - s * h;
            }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (fabs(e[l]) > eps * tst1 && ++iter < 200);
    }
    d[l] += f;
    e[l] = 0;
  }
}
#undef V

// vectors = false: the eigenvalues only (X comes back empty) -- the same reduction and the same QL iteration without the
// accumulation of the reflectors and the rotations of the vectors, i.e. the same values bit for bit at a third of the time
static void host_sym_eig(const std::vector<double>& H, int n, std::vector<double>& theta, std::vector<double>& X,
                         bool vectors = true) {
  std::vector<double> Vv((size_t)n * n), d((size_t)n), e((size_t)n);
  double big = 0;
  for (size_t t = 0; t < (size_t)n * n; t++) big = std::max(big, fabs(H[t]));
  const double inv = big > 0 ? 1.0 / big : 1.0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) Vv[(size_t)j * n + i] = 0.5 * (H[i + (size_t)j * n] + H[j + (size_t)i * n]) * inv;
  sym_eig_core(Vv.data(), d.data(), e.data(), n, vectors);
  std::vector<int> ord((size_t)n);
  for (int i = 0; i < n; i++) ord[(size_t)i] = i;
  std::sort(ord.begin(), ord.end(), [&](int x, int y) { return d[x] > d[y]; });
  theta.resize((size_t)n);
  X.assign(vectors ? (size_t)n * n : 0, 0.0);
  for (int j = 0; j < n; j++) {
    theta[(size_t)j] = d[(size_t)ord[(size_t)j]] * (big > 0 ? big : 1.0);
    if (vectors)
      for (int i = 0; i < n; i++) X[i + (size_t)j * n] = Vv[(size_t)ord[(size_t)j] * n + i];
  }
}
