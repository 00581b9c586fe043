// What the host can do for `K[k] += counts[k]` (the per-block cost of the literal increment_* mirrors, pairwise.hip
// add_counts_to_caller): N x N doubles read-modify-written from 16-bit counts, by thread count and loop form, on anonymous
// memory and on a MAP_SHARED file mapping (what bigstatsr gives the R drivers).
//   g++ -O3 -std=c++17 -pthread tools/host_rmw_probe.cpp -o /tmp/host_rmw_probe && /tmp/host_rmw_probe [n] [dir]
#include <errno.h>
#include <fcntl.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void add_scalar(double* dst, const uint16_t* q, size_t k0, size_t k1) {
  for (size_t k = k0; k < k1; k++) dst[k] += (double)((int)q[k]);
}

__attribute__((target("avx2"))) static void add_avx2(double* dst, const uint16_t* q, size_t k0, size_t k1) {
  size_t k = k0;
  for (; k < k1 && (((uintptr_t)(dst + k)) & 31); k++) dst[k] += (double)q[k];
  for (; k + 16 <= k1; k += 16) {
    const __m256i w = _mm256_loadu_si256((const __m256i*)(q + k));
    const __m256i lo = _mm256_cvtepu16_epi32(_mm256_castsi256_si128(w)), hi = _mm256_cvtepu16_epi32(_mm256_extracti128_si256(w, 1));
    _mm256_store_pd(dst + k, _mm256_add_pd(_mm256_load_pd(dst + k), _mm256_cvtepi32_pd(_mm256_castsi256_si128(lo))));
    _mm256_store_pd(dst + k + 4, _mm256_add_pd(_mm256_load_pd(dst + k + 4), _mm256_cvtepi32_pd(_mm256_extracti128_si256(lo, 1))));
    _mm256_store_pd(dst + k + 8, _mm256_add_pd(_mm256_load_pd(dst + k + 8), _mm256_cvtepi32_pd(_mm256_castsi256_si128(hi))));
    _mm256_store_pd(dst + k + 12, _mm256_add_pd(_mm256_load_pd(dst + k + 12), _mm256_cvtepi32_pd(_mm256_extracti128_si256(hi, 1))));
  }
  for (; k < k1; k++) dst[k] += (double)q[k];
}

__attribute__((target("avx2"))) static void add_avx2_nt(double* dst, const uint16_t* q, size_t k0, size_t k1) {
  size_t k = k0;
  for (; k < k1 && (((uintptr_t)(dst + k)) & 31); k++) dst[k] += (double)q[k];
  for (; k + 16 <= k1; k += 16) {
    const __m256i w = _mm256_loadu_si256((const __m256i*)(q + k));
    const __m256i lo = _mm256_cvtepu16_epi32(_mm256_castsi256_si128(w)), hi = _mm256_cvtepu16_epi32(_mm256_extracti128_si256(w, 1));
    _mm256_stream_pd(dst + k, _mm256_add_pd(_mm256_load_pd(dst + k), _mm256_cvtepi32_pd(_mm256_castsi256_si128(lo))));
    _mm256_stream_pd(dst + k + 4, _mm256_add_pd(_mm256_load_pd(dst + k + 4), _mm256_cvtepi32_pd(_mm256_extracti128_si256(lo, 1))));
    _mm256_stream_pd(dst + k + 8, _mm256_add_pd(_mm256_load_pd(dst + k + 8), _mm256_cvtepi32_pd(_mm256_castsi256_si128(hi))));
    _mm256_stream_pd(dst + k + 12, _mm256_add_pd(_mm256_load_pd(dst + k + 12), _mm256_cvtepi32_pd(_mm256_extracti128_si256(hi, 1))));
  }
  _mm_sfence();
  for (; k < k1; k++) dst[k] += (double)q[k];
}

// both matrices in one sweep (IBS: K += a, K2 += b)
__attribute__((target("avx2"))) static void add2_avx2(double* A, double* B, const uint16_t* qa, const uint16_t* qb, size_t k0, size_t k1) {
  const size_t step = 4096;  // elements per matrix and turn: 32 KB of doubles
  for (size_t k = k0; k < k1; k += step) {
    const size_t e = k + step < k1 ? k + step : k1;
    add_avx2(A, qa, k, e);
    add_avx2(B, qb, k, e);
  }
}

typedef void (*AddFn)(double*, const uint16_t*, size_t, size_t);

static double run(AddFn fn, double* dst, const uint16_t* q, size_t nn, int nt) {
  std::vector<std::thread> th;
  const double t0 = now();
  for (int t = 1; t < nt; t++) th.emplace_back([=] { fn(dst, q, nn * t / nt, nn * (t + 1) / nt); });
  fn(dst, q, 0, nn / nt);
  for (auto& t : th) t.join();
  return now() - t0;
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? atol(argv[1]) : 5000, nn = n * n;
  const std::string dir = argc > 2 ? argv[2] : "/tmp";
  printf("hardware_concurrency %u, n = %zu (%.0f MB of doubles per matrix)\n", std::thread::hardware_concurrency(), n, nn * 8 / 1e6);
  std::vector<uint16_t> q(nn);
  for (size_t k = 0; k < nn; k++) q[k] = (uint16_t)(k * 2654435761u >> 17);
  // anonymous memory
  double* anon = (double*)mmap(nullptr, nn * 8, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  memset(anon, 0, nn * 8);
  // a file mapping as bigstatsr makes it
  const std::string path = dir + "/host_rmw_probe.bk";
  int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0600);
  {
    std::vector<double> z(1 << 20, 0.0);
    for (size_t done = 0; done < nn; done += z.size()) (void)!write(fd, z.data(), 8 * (nn - done < z.size() ? nn - done : z.size()));
  }
  double* file = (double*)mmap(nullptr, nn * 8, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  struct { const char* name; AddFn fn; } forms[] = {{"scalar (today)", add_scalar}, {"avx2", add_avx2}, {"avx2 + stream stores", add_avx2_nt}};
  for (int where = 0; where < 2; where++) {
    double* dst = where ? file : anon;
    for (auto& f : forms)
      for (int nt : {4, 8, 16, 24, 32}) {
        double best = 1e9, first = 0;
        for (int rep = 0; rep < 5; rep++) {
          const double dt = run(f.fn, dst, q.data(), nn, nt);
          if (rep == 0) first = dt;
          if (dt < best) best = dt;
        }
        printf("%-9s %-22s %2d threads: best %6.2f ms (first %6.2f)  %6.1f GB/s of read+write+counts\n", where ? "file map" : "anonymous",
               f.name, nt, best * 1e3, first * 1e3, nn * 18.0 / best / 1e9);
        fflush(stdout);
      }
  }
  // first touch of a fresh accumulator file (bigstatsr::FBM(n, n, init = 0): written, then mapped): what the FIRST block of a
  // driver loop pays on top.  Forms: write faults from the adding threads; MADV_POPULATE_WRITE; MAP_POPULATE; both on 16 threads' slices
  {
    auto fresh = [&](const char* tag) {
      const std::string p2 = dir + "/host_rmw_probe_" + tag + ".bk";
      int f2 = open(p2.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0600);
      std::vector<double> z(1 << 20, 0.0);
      for (size_t done = 0; done < nn; done += z.size()) (void)!write(f2, z.data(), 8 * (nn - done < z.size() ? nn - done : z.size()));
      unlink(p2.c_str());
      return f2;
    };
    for (int nt : {1, 4, 16, 32}) {
      int f2 = fresh("touch");
      double* m2 = (double*)mmap(nullptr, nn * 8, PROT_READ | PROT_WRITE, MAP_SHARED, f2, 0);
      const double dt = run(add_avx2, m2, q.data(), nn, nt);
      printf("first touch by the adding threads, %2d threads: %7.2f ms\n", nt, dt * 1e3);
      munmap(m2, nn * 8);
      close(f2);
    }
    {
      int f2 = fresh("madv");
      double* m2 = (double*)mmap(nullptr, nn * 8, PROT_READ | PROT_WRITE, MAP_SHARED, f2, 0);
      double t0 = now();
#ifdef MADV_POPULATE_WRITE
      const int rc = madvise(m2, nn * 8, MADV_POPULATE_WRITE);
      printf("MADV_POPULATE_WRITE: rc %d (%s) %7.2f ms", rc, rc ? strerror(errno) : "ok", (now() - t0) * 1e3);
#else
      printf("MADV_POPULATE_WRITE not in the headers");
#endif
      const double dt = run(add_avx2, m2, q.data(), nn, 16);
      printf(", then the add on 16 threads %7.2f ms\n", dt * 1e3);
      munmap(m2, nn * 8);
      close(f2);
    }
#ifdef MADV_POPULATE_WRITE
    for (int nt : {4, 16}) {
      int f2 = fresh("madvpar");
      double* m2 = (double*)mmap(nullptr, nn * 8, PROT_READ | PROT_WRITE, MAP_SHARED, f2, 0);
      double t0 = now();
      std::vector<std::thread> th;
      const size_t pages = (nn * 8 + 4095) / 4096;
      for (int t = 0; t < nt; t++)
        th.emplace_back([=] {
          const size_t a = pages * t / nt * 4096, b = pages * (t + 1) / nt * 4096;
          (void)madvise((char*)m2 + a, (b < nn * 8 ? b : nn * 8) - a, MADV_POPULATE_WRITE);
        });
      for (auto& t : th) t.join();
      printf("MADV_POPULATE_WRITE on %2d threads' slices: %7.2f ms", nt, (now() - t0) * 1e3);
      const double dt = run(add_avx2, m2, q.data(), nn, 16);
      printf(", then the add %7.2f ms\n", dt * 1e3);
      munmap(m2, nn * 8);
      close(f2);
    }
#endif
    {
      int f2 = fresh("mappop");
      double t0 = now();
      double* m2 = (double*)mmap(nullptr, nn * 8, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, f2, 0);
      printf("MAP_POPULATE: %7.2f ms", (now() - t0) * 1e3);
      const double dt = run(add_avx2, m2, q.data(), nn, 16);
      printf(", then the add on 16 threads %7.2f ms\n", dt * 1e3);
      munmap(m2, nn * 8);
      close(f2);
    }
    for (int nt : {4, 8, 16}) {  // K and K2 at once: two files, nt threads each
      int fa = fresh("two_a"), fb = fresh("two_b");
      double* ma = (double*)mmap(nullptr, nn * 8, PROT_READ | PROT_WRITE, MAP_SHARED, fa, 0);
      double* mb = (double*)mmap(nullptr, nn * 8, PROT_READ | PROT_WRITE, MAP_SHARED, fb, 0);
      const double t0 = now();
      std::thread other([&] { (void)run(add_avx2, mb, q.data(), nn, nt); });
      (void)run(add_avx2, ma, q.data(), nn, nt);
      other.join();
      printf("first touch of TWO files at once, %2d threads each: %7.2f ms\n", nt, (now() - t0) * 1e3);
      munmap(ma, nn * 8);
      munmap(mb, nn * 8);
      close(fa);
      close(fb);
    }
    {  // no mapping at all: pread the matrix, add, pwrite it back (16 threads, 4-MB pieces)
      int f2 = fresh("prw");
      double t0 = now();
      std::vector<std::thread> th;
      for (int t = 0; t < 16; t++)
        th.emplace_back([=, &q] {
          std::vector<double> buf(1 << 19);
          for (size_t k = nn * t / 16; k < nn * (t + 1) / 16; k += buf.size()) {
            const size_t e = std::min(nn * (t + 1) / 16, k + buf.size());
            (void)!pread(f2, buf.data(), (e - k) * 8, k * 8);
            add_avx2(buf.data() - k, q.data(), k, e);
            (void)!pwrite(f2, buf.data(), (e - k) * 8, k * 8);
          }
        });
      for (auto& t : th) t.join();
      printf("pread + add + pwrite, 16 threads: %7.2f ms\n", (now() - t0) * 1e3);
      close(f2);
    }
  }
  // checksum so that nothing is optimised away
  double s = 0;
  for (size_t k = 0; k < nn; k += 4097) s += anon[k] + file[k];
  printf("checksum %.0f\n", s);
  munmap(file, nn * 8);
  close(fd);
  unlink(path.c_str());
  return 0;
}
