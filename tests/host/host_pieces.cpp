// tests/host/host_pieces.cpp -- self-checking driver for the host-only pieces of libtpg_hip.so (tidypopgen_amd/csrc/host/*.h),
// built WITHOUT HIP by tests/test_host_sanitizers.py with -fsanitize=address,undefined and (the transport) -fsanitize=thread:
// the CPU-side equivalent of the reference's valgrind job (.github/workflows/R-CMD-check-valgrind.yaml:50-51).
//   host_pieces eig | bands | relfilter | nibpack | bedpack | addcounts | fsttiles | bits2 | inproc [threads] | inproc_mismatch
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#define TPG_HOST_NO_CLONES 1  // one plain build of the QL (function multiversioning and sanitizers do not mix everywhere)
#include "host/host_bands.h"
#include "host/host_eig.h"
#include "host/host_fsttiles.h"
#include "host/host_inproc.h"
#include "host/host_addcounts.h"
#include "host/host_bedpack.h"
#include "host/host_nibpack.h"
#include "host/host_relfilter.h"
#include "host/host_bits2.h"

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static double urand() {  // splitmix64 -> [0, 1)
  uint64_t x = (g_rng += 0x9E3779B97F4A7C15ull);
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

#define CHECK(cond, ...)                                    \
  do {                                                      \
    if (!(cond)) {                                          \
      fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
      fprintf(stderr, __VA_ARGS__);                         \
      fprintf(stderr, "\n");                                \
      return 1;                                             \
    }                                                       \
  } while (0)

// symmetric eigen-decomposition, Cholesky and triangular inverse at the sizes the eigen solver uses (b = 2 k + 12 <= 64)
static int test_eig() {
  for (int n : {1, 2, 3, 7, 32, 52, 64}) {
    for (int kind = 0; kind < 3; kind++) {
      std::vector<double> H((size_t)n * n);
      for (int j = 0; j < n; j++)
        for (int i = 0; i <= j; i++) {
          double v = kind == 0 ? urand() - 0.5 : kind == 1 ? (i == j ? (double)(n - i) * 1e3 : 1e-3 * (urand() - 0.5)) : (i == j ? 1.0 : 0.0);
          H[i + (size_t)j * n] = H[j + (size_t)i * n] = v;
        }
      std::vector<double> th, X;
      host_sym_eig(H, n, th, X);
      CHECK((int)th.size() == n && (int)X.size() == n * n, "sizes");
      {  // eigenvalues only: the same values bit for bit
        std::vector<double> th2, X2;
        host_sym_eig(H, n, th2, X2, false);
        CHECK(X2.empty() && th2.size() == th.size(), "values-only sizes");
        for (int j = 0; j < n; j++) CHECK(th2[j] == th[j], "values-only eigenvalue %d differs (n = %d)", j, n);
      }
      double hmax = 0;
      for (double v : H) hmax = std::max(hmax, fabs(v));
      for (int j = 0; j < n; j++) {
        if (j) CHECK(th[j - 1] >= th[j], "eigenvalues not descending at %d (n = %d)", j, n);
        double res = 0;
        for (int i = 0; i < n; i++) {
          double s = 0;
          for (int k = 0; k < n; k++) s += H[i + (size_t)k * n] * X[k + (size_t)j * n];
          res = std::max(res, fabs(s - th[j] * X[i + (size_t)j * n]));
        }
        CHECK(res <= 1e-12 * hmax * n, "residual %g (n = %d, kind %d, pair %d)", res, n, kind, j);
        for (int l = 0; l <= j; l++) {
          double dot = 0;
          for (int k = 0; k < n; k++) dot += X[k + (size_t)j * n] * X[k + (size_t)l * n];
          CHECK(fabs(dot - (l == j ? 1.0 : 0.0)) <= 1e-12 * n, "orthonormality %g (n = %d)", dot, n);
        }
      }
    }
    // G = B'B + I is positive definite: R'R = G, R Ri = I
    std::vector<double> B((size_t)n * n), G((size_t)n * n, 0.0);
    for (double& v : B) v = urand() - 0.5;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        double s = i == j ? 1.0 : 0.0;
        for (int k = 0; k < n; k++) s += B[k + (size_t)i * n] * B[k + (size_t)j * n];
        G[i + (size_t)j * n] = s;
      }
    std::vector<double> R = G, Ri;
    CHECK(host_cholesky_upper(R, n), "cholesky refused a positive definite matrix (n = %d)", n);
    host_upper_inverse(R, n, Ri);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        double rr = 0, ri = 0;
        for (int k = 0; k < n; k++) { rr += R[k + (size_t)i * n] * R[k + (size_t)j * n]; ri += R[i + (size_t)k * n] * Ri[k + (size_t)j * n]; }
        CHECK(fabs(rr - G[i + (size_t)j * n]) <= 1e-12 * n, "R'R != G (n = %d)", n);
        CHECK(fabs(ri - (i == j ? 1.0 : 0.0)) <= 1e-10 * n, "R Ri != I (n = %d): %g", n, ri);
      }
    std::vector<double> bad((size_t)n * n, 0.0);  // not positive definite: refused, no out-of-bounds access on the way
    bad[0] = -1.0;
    CHECK(!host_cholesky_upper(bad, n), "cholesky accepted an indefinite matrix");
  }
  return 0;
}

// bands: contiguous, cover every super-tile row once, every band fits the padded chunk
static int test_bands() {
  for (int64_t nst = 1; nst <= 70; nst++)
    for (int nranks : {1, 2, 3, 4, 7, 8, 16, 64}) {
      std::vector<int32_t> band;
      int64_t chunk = -1;
      pw_bands(nst, nranks, band, chunk);
      CHECK((int)band.size() == nranks + 1 && band[0] == 0 && band[(size_t)nranks] == nst, "band ends (nst %lld, %d ranks)", (long long)nst, nranks);
      auto off = [&](int64_t I) { return TPG_PW_TA * (I * nst - (I * (I - 1)) / 2); };
      for (int r = 0; r < nranks; r++) {
        CHECK(band[(size_t)r] <= band[(size_t)r + 1], "band %d runs backwards", r);
        CHECK(off(band[(size_t)r + 1]) - off(band[(size_t)r]) <= chunk, "band %d does not fit its chunk", r);
      }
      CHECK(chunk >= 1 && chunk * nranks >= off(nst), "chunks do not hold all units");
    }
  return 0;
}

// the literal loop of R/filter_high_relatedness.R:55-137 (both means recomputed from scratch for every comparison, in
// long double like R's mean()), against the incremental implementation
static int literal_filter(const std::vector<double>& Ain, size_t N, double thr, std::vector<uint8_t>& keep) {
  std::vector<double> A(Ain);
  for (double& x : A) x = fabs(x);
  std::vector<double> cm(N), col;
  for (size_t j = 0; j < N; j++) {
    col.clear();
    for (size_t i = 0; i < N; i++)
      if (i != j && A[i + j * N] == A[i + j * N]) col.push_back(A[i + j * N]);
    cm[j] = r_mean_ld(col);
  }
  std::vector<int32_t> ord(N);
  for (size_t j = 0; j < N; j++) ord[j] = (int32_t)j;
  std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
    const double x = cm[(size_t)a], y = cm[(size_t)b];
    const bool xn = x != x, yn = y != y;
    if (xn || yn) return !xn && yn;
    return x > y;
  });
  std::vector<double> M(N * N), M2;
  for (size_t b = 0; b < N; b++)
    for (size_t a = 0; a < N; a++) M[a + b * N] = A[(size_t)ord[a] + (size_t)ord[b] * N];
  M2 = M;
  for (size_t k = 0; k < N; k++) M2[k + k * N] = NAN;
  std::vector<uint8_t> del(N, 0);
  std::vector<double> tmp;
  for (size_t i = 0; i + 1 < N; i++) {
    bool any = false;
    for (double x : M2) any |= (x == x && x > thr);
    if (!any) break;
    if (del[i]) continue;
    for (size_t j = i + 1; j < N; j++) {
      if (del[i] || del[j]) continue;
      if (!(M[i + j * N] > thr)) continue;
      tmp.clear();
      for (size_t l = 0; l < N; l++) if (M2[i + l * N] == M2[i + l * N]) tmp.push_back(M2[i + l * N]);
      const double mn1 = r_mean_ld(tmp);
      tmp.clear();
      for (size_t l = 0; l < N; l++)
        for (size_t k = 0; k < N; k++) if (k != j && M2[k + l * N] == M2[k + l * N]) tmp.push_back(M2[k + l * N]);
      const double mn2 = r_mean_ld(tmp);
      const size_t d = mn1 > mn2 ? i : j;
      del[d] = 1;
      for (size_t k = 0; k < N; k++) { M2[d + k * N] = NAN; M2[k + d * N] = NAN; }
    }
  }
  keep.assign(N, 0);
  for (size_t k = 0; k < N; k++) keep[(size_t)ord[k]] = !del[k];
  return 0;
}

static int test_relfilter() {
  for (int rep = 0; rep < 40; rep++) {
    const size_t N = 1 + (size_t)(urand() * 45);
    std::vector<double> A(N * N);
    for (size_t j = 0; j < N; j++)
      for (size_t i = 0; i <= j; i++) {
        double v = i == j ? 0.5 : (urand() < 0.15 ? 0.1 + 0.4 * urand() : 0.05 * (urand() - 0.5));
        if (rep % 5 == 4 && i != j && urand() < 0.02) v = 0.25;  // ties
        A[i + j * N] = A[j + i * N] = v;
      }
    const double thr = rep % 3 == 0 ? 0.2 : 0.0884;
    std::vector<uint8_t> want, got(N, 7);
    std::vector<int32_t> order(N, -1);
    literal_filter(A, N, thr, want);
    std::vector<double> Ac(A);
    std::string err;
    CHECK(tpg_host_filter_high_relatedness(Ac, (int64_t)N, thr, got.data(), order.data(), err) == 0, "filter failed: %s", err.c_str());
    for (size_t k = 0; k < N; k++) CHECK(got[k] == want[k], "decision for individual %zu differs (rep %d, N = %zu)", k, rep, N);
    std::vector<int> seen(N, 0);
    for (size_t k = 0; k < N; k++) { CHECK(order[k] >= 0 && (size_t)order[k] < N, "order out of range"); seen[(size_t)order[k]]++; }
    for (size_t k = 0; k < N; k++) CHECK(seen[k] == 1, "order is not a permutation");
  }
  {  // an NA among the compared relatednesses is R's error, not a crash
    const size_t N = 5;
    std::vector<double> A(N * N, 0.3);
    A[1 + 0 * N] = A[0 + 1 * N] = NAN;
    std::vector<uint8_t> keep(N);
    std::string err;
    const int rc = tpg_host_filter_high_relatedness(A, (int64_t)N, 0.2, keep.data(), nullptr, err);
    CHECK(rc != 0 && err.find("missing value") != std::string::npos, "NA not reported: rc %d '%s'", rc, err.c_str());
  }
  return 0;
}

// R threads, many rounds of all-reduces of both types and changing sizes; every rank must see the exact sums every round
static int test_inproc(int R) {
  InprocGroup g;
  g.n = R;
  g.slot.assign((size_t)R, nullptr);
  g.slot_count.assign((size_t)R, 0);
  g.slot_dtype.assign((size_t)R, 0);
  std::vector<InprocRank> ranks((size_t)R);
  std::vector<int> fails((size_t)R, 0);
  auto body = [&](int r) {
    for (int round = 0; round < 300; round++) {
      const int64_t count = 1 + (round * 37) % 513;
      if (round & 1) {
        std::vector<int32_t> b((size_t)count);
        for (int64_t i = 0; i < count; i++) b[(size_t)i] = (int32_t)(r * 1000 + i + round);
        if (inproc_allreduce(&ranks[(size_t)r], b.data(), count, 0) != 0) fails[(size_t)r]++;
        for (int64_t i = 0; i < count; i++)
          if (b[(size_t)i] != (int32_t)(1000 * (R * (R - 1) / 2) + R * (i + round))) fails[(size_t)r]++;
      } else {
        std::vector<double> b((size_t)count);
        for (int64_t i = 0; i < count; i++) b[(size_t)i] = 0.5 * r + (double)i;
        if (inproc_allreduce(&ranks[(size_t)r], b.data(), count, 1) != 0) fails[(size_t)r]++;
        for (int64_t i = 0; i < count; i++)
          if (b[(size_t)i] != 0.5 * (R * (R - 1) / 2) + (double)R * (double)i) fails[(size_t)r]++;
      }
    }
  };
  std::vector<std::thread> th;
  for (int r = 0; r < R; r++) { ranks[(size_t)r] = InprocRank{&g, r}; }
  for (int r = 0; r < R; r++) th.emplace_back(body, r);
  for (auto& t : th) t.join();
  for (int r = 0; r < R; r++) CHECK(fails[(size_t)r] == 0, "rank %d saw %d wrong sums", r, fails[(size_t)r]);
  return 0;
}

// ranks that disagree on the count: every one of them gets an error back, nobody reads past a shorter buffer, and the
// group is usable afterwards
static int test_inproc_mismatch() {
  const int R = 3;
  InprocGroup g;
  g.n = R;
  g.slot.assign((size_t)R, nullptr);
  g.slot_count.assign((size_t)R, 0);
  g.slot_dtype.assign((size_t)R, 0);
  std::vector<InprocRank> ranks((size_t)R);
  for (int r = 0; r < R; r++) ranks[(size_t)r] = InprocRank{&g, r};
  std::vector<int> rc1((size_t)R, -1), rc2((size_t)R, -1);
  std::vector<double> sums((size_t)R, 0);
  auto body = [&](int r) {
    std::vector<int32_t> a((size_t)(r == 1 ? 4 : 64), 1);
    rc1[(size_t)r] = inproc_allreduce(&ranks[(size_t)r], a.data(), (int64_t)a.size(), 0);
    std::vector<double> b(8, 1.0 + r);
    rc2[(size_t)r] = inproc_allreduce(&ranks[(size_t)r], b.data(), 8, 1);
    sums[(size_t)r] = b[7];
  };
  std::vector<std::thread> th;
  for (int r = 0; r < R; r++) th.emplace_back(body, r);
  for (auto& t : th) t.join();
  for (int r = 0; r < R; r++) {
    CHECK(rc1[(size_t)r] != 0, "rank %d did not see the mismatch", r);
    CHECK(rc2[(size_t)r] == 0 && sums[(size_t)r] == 6.0, "group unusable after a mismatch (rank %d: rc %d, sum %g)", r, rc2[(size_t)r], sums[(size_t)r]);
  }
  return 0;
}

// nibble pack (the packed FBM upload): every length and alignment around the 64-byte vector step, the OR of the input bytes
static int test_nibpack() {
  for (size_t n : {0, 2, 30, 62, 64, 66, 126, 128, 130, 4096, 4098, 100000}) {
    for (int shift = 0; shift < 3; shift++) {
      std::vector<uint8_t> in(n + 8), out(n / 2 + 8, 0xAA);
      for (size_t i = 0; i < n; i++) in[shift + i] = (uint8_t)(urand() * 7);
      const uint8_t seen = tpg_nibpack(in.data() + shift, out.data() + shift, n);
      uint8_t want_seen = 0;
      for (size_t i = 0; i < n; i++) want_seen |= in[shift + i];
      CHECK(seen == want_seen && seen < 16, "OR of the input bytes: %u, want %u (n = %zu)", seen, want_seen, n);
      for (size_t i = 0; i < n / 2; i++)
        CHECK(out[shift + i] == (uint8_t)(in[shift + 2 * i] | (in[shift + 2 * i + 1] << 4)), "byte %zu of %zu", i, n / 2);
      CHECK(out[shift + n / 2] == 0xAA, "wrote past the end (n = %zu)", n);
      if (n >= 2) {  // one byte that does not fit a nibble, anywhere: reported
        const size_t at = (size_t)(urand() * n);
        in[shift + at] = (uint8_t)(16 + urand() * 200);
        CHECK(tpg_nibpack(in.data() + shift, out.data() + shift, n) >= 16, "a byte >= 16 at %zu of %zu went unnoticed", at, n);
      }
    }
  }
  return 0;
}

// 2-bit pack into the .bed layout (a streamed run with one code table): lengths and alignments around the 128-byte vector step,
// the last byte's unused bit pairs, the OR of the input bytes, every byte through the table
static int test_bedpack() {
  uint8_t lut[16];
  for (int k = 0; k < 16; k++) lut[k] = (uint8_t)((k * 7 + 3) & 3);
  for (size_t n : {0, 1, 3, 4, 5, 127, 128, 129, 131, 255, 256, 260, 4096, 4099, 100001}) {
    for (int shift = 0; shift < 3; shift++) {
      std::vector<uint8_t> in(n + 8), out((n + 3) / 4 + 8, 0xAA);
      for (size_t i = 0; i < n; i++) in[shift + i] = (uint8_t)(urand() * 16);
      const uint8_t seen = tpg_bedpack(in.data() + shift, out.data() + shift, n, lut);
      uint8_t want_seen = 0;
      for (size_t i = 0; i < n; i++) want_seen |= in[shift + i];
      CHECK(seen == want_seen && seen < 16, "OR of the input bytes: %u, want %u (n = %zu)", seen, want_seen, n);
      for (size_t i = 0; i < (n + 3) / 4; i++) {
        uint8_t want = 0;
        for (size_t r = 0; r < 4 && 4 * i + r < n; r++) want |= (uint8_t)(lut[in[shift + 4 * i + r]] << (2 * r));
        CHECK(out[shift + i] == want, "byte %zu of %zu: %02x, want %02x (n = %zu)", i, (n + 3) / 4, out[shift + i], want, n);
      }
      CHECK(out[shift + (n + 3) / 4] == 0xAA, "wrote past the end (n = %zu)", n);
      if (n >= 1) {  // one byte that no 16-entry table holds, anywhere: reported
        const size_t at = (size_t)(urand() * n);
        in[shift + at] = (uint8_t)(16 + urand() * 239);
        CHECK(tpg_bedpack(in.data() + shift, out.data() + shift, n, lut) >= 16, "a byte >= 16 at %zu of %zu went unnoticed", at, n);
      }
    }
  }
  return 0;
}

// K += counts (the literal increment_* mirrors): lengths and alignments around the vector step, the bias of a signed 16-bit
// count, large sums (exact: integers below 2^53), nothing written past the end
static int test_addcounts() {
  for (size_t n : {0, 1, 7, 8, 15, 16, 17, 31, 33, 4096, 4099, 100001}) {
    for (int shift = 0; shift < 3; shift++) {
      for (int bias : {0, 32768}) {
        std::vector<double> dst(n + 8), want(n + 8);
        std::vector<uint16_t> q(n + 8);
        std::vector<int32_t> w(n + 8);
        for (size_t i = 0; i < n + 8; i++) {
          dst[i] = want[i] = (double)(int64_t)(urand() * 9.0e15) - 4.5e15;
          q[i] = (uint16_t)(urand() * 65536);
          w[i] = (int32_t)((urand() - 0.5) * 4.0e9);
        }
        for (size_t i = 0; i < n; i++) want[shift + i] += (double)((int)q[shift + i] - bias);
        tpg_add_counts_u16(dst.data() + shift, q.data() + shift, n, bias);
        for (size_t i = 0; i < n + 8; i++) CHECK(dst[i] == want[i], "u16, bias %d: element %zu of %zu (shift %d)", bias, i, n, shift);
        for (size_t i = 0; i < n; i++) want[shift + i] += (double)w[shift + i];
        tpg_add_counts_i32(dst.data() + shift, w.data() + shift, n);
        for (size_t i = 0; i < n + 8; i++) CHECK(dst[i] == want[i], "i32: element %zu of %zu (shift %d)", i, n, shift);
      }
    }
  }
  return 0;
}

// the pair list of pairwise_pop_fst cut into tiles of populations: every listed pair in exactly one slot of the right tile,
// whatever the order, the orientation and the repeats of the list
static int test_fsttiles() {
  for (int G : {2, 3, 7, 51, 64}) {
    for (int kind = 0; kind < 3; kind++) {
      std::vector<int32_t> p0;
      for (int a = 0; a < G; a++)
        for (int b = a + 1; b < G; b++) {
          if (kind == 1 && urand() < 0.5) continue;                     // a subset
          const bool flip = kind == 2 && urand() < 0.4;                 // (g2, g1)
          p0.push_back(flip ? b : a);
          p0.push_back(flip ? a : b);
          if (kind == 2 && urand() < 0.05) { p0.push_back(a); p0.push_back(b); }  // listed twice
        }
      const int P = (int)(p0.size() / 2);
      std::vector<int32_t> tasks;
      fst_wc84_tiles(p0, P, tasks);
      CHECK(tasks.size() % FSTW_TASK_INTS == 0, "task record size");
      std::vector<int> seen((size_t)P, 0);
      for (size_t t = 0; t < tasks.size() / FSTW_TASK_INTS; t++) {
        const int32_t* T = &tasks[t * FSTW_TASK_INTS];
        CHECK(T[0] % FSTW_TR == 0 && T[1] % FSTW_TC == 0 && T[0] >= 0 && T[1] >= 0, "tile origin");
        int used = 0;
        for (int k = 0; k < FSTW_TR * FSTW_TC; k++) {
          const int pi = T[2 + k];
          if (pi < 0) continue;
          CHECK(pi < P, "pair index %d out of %d", pi, P);
          CHECK(p0[(size_t)2 * pi] == T[0] + k / FSTW_TC && p0[(size_t)2 * pi + 1] == T[1] + k % FSTW_TC, "pair %d in the wrong slot", pi);
          seen[(size_t)pi]++;
          used++;
        }
        CHECK(used > 0, "empty tile");
      }
      for (int pi = 0; pi < P; pi++) CHECK(seen[(size_t)pi] == 1, "pair %d placed %d times (G = %d, kind %d)", pi, seen[(size_t)pi], G, kind);
    }
  }
  return 0;
}

// the 16 x 16 transposition of 2-bit fields against its definition, and the byte permute against v_perm_b32's table
static int test_bits2() {
  CHECK(tpg_byte_perm(0x77665544u, 0x33221100u, 0x07040300u) == 0x77443300u, "byte permute");
  uint64_t st = 0x9E3779B97F4A7C15ull;
  for (int rep = 0; rep < 2000; rep++) {
    uint32_t W[16], T[16];
    for (int k = 0; k < 16; k++) {
      st = st * 6364136223846793005ull + 1442695040888963407ull;
      W[k] = T[k] = (uint32_t)(st >> 32);
    }
    tpg_transpose16_2bit(T);
    for (int k = 0; k < 16; k++)
      for (int p = 0; p < 16; p++)
        CHECK(((T[p] >> (2 * k)) & 3u) == ((W[k] >> (2 * p)) & 3u), "transposition: word %d field %d (rep %d)", k, p, rep);
  }
  return 0;
}

int main(int argc, char** argv) {
  const std::string what = argc > 1 ? argv[1] : "";
  int rc = 2;
  if (what == "eig") rc = test_eig();
  else if (what == "bands") rc = test_bands();
  else if (what == "relfilter") rc = test_relfilter();
  else if (what == "inproc") rc = test_inproc(argc > 2 ? atoi(argv[2]) : 4);
  else if (what == "inproc_mismatch") rc = test_inproc_mismatch();
  else if (what == "nibpack") rc = test_nibpack();
  else if (what == "bedpack") rc = test_bedpack();
  else if (what == "addcounts") rc = test_addcounts();
  else if (what == "fsttiles") rc = test_fsttiles();
  else if (what == "bits2") rc = test_bits2();
  else fprintf(stderr, "usage: host_pieces eig | bands | relfilter | nibpack | bedpack | addcounts | fsttiles | bits2 | inproc [threads] | inproc_mismatch\n");
  if (rc == 0) printf("ok %s\n", what.c_str());
  return rc;
}
