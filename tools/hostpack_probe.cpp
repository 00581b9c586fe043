// hostpack_probe.cpp -- how fast can the HOST turn the 1-byte-per-genotype bytes of a bigstatsr .bk (out of the page cache)
// into 4 bits (any byte < 16: what a raw + imputed pair of views needs) or 2 bits (one view, through a 16-entry code table)
// per genotype, with T threads and AVX2?  The question behind "pack the .bk bytes on the host inside the upload team"
// (VERDICT round 3, item 5): the packing team's time has to stay well below the PCIe time it saves.
//   g++ -O3 -mavx2 -pthread tools/hostpack_probe.cpp -o tools/hostpack_probe.bin && tools/hostpack_probe.bin [GiB] [dir]
#include <fcntl.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static void pack4(const uint8_t* in, uint8_t* out, size_t n) {  // n multiple of 64: out[i] = in[2i] | in[2i+1] << 4
  const __m256i mul = _mm256_set1_epi16(0x1001);
  for (size_t i = 0; i < n; i += 64) {
    const __m256i a = _mm256_loadu_si256((const __m256i*)(in + i)), b = _mm256_loadu_si256((const __m256i*)(in + i + 32));
    const __m256i pa = _mm256_maddubs_epi16(a, mul), pb = _mm256_maddubs_epi16(b, mul);
    _mm256_storeu_si256((__m256i*)(out + i / 2), _mm256_permute4x64_epi64(_mm256_packus_epi16(pa, pb), 0xD8));
  }
}
static void pack2(const uint8_t* in, uint8_t* out, size_t n) {  // n multiple of 128: four 2-bit codes per byte through a table
  const __m256i lut = _mm256_setr_epi8(0, 1, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 0, 1, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3);
  const __m256i m4 = _mm256_set1_epi16(0x0401), m16 = _mm256_set1_epi16(0x1001);
  for (size_t i = 0; i < n; i += 128) {
    __m256i v[4];
    for (int k = 0; k < 4; k++) v[k] = _mm256_shuffle_epi8(lut, _mm256_loadu_si256((const __m256i*)(in + i + 32 * k)));
    const __m256i q0 = _mm256_permute4x64_epi64(_mm256_packus_epi16(_mm256_maddubs_epi16(v[0], m4), _mm256_maddubs_epi16(v[1], m4)), 0xD8);
    const __m256i q1 = _mm256_permute4x64_epi64(_mm256_packus_epi16(_mm256_maddubs_epi16(v[2], m4), _mm256_maddubs_epi16(v[3], m4)), 0xD8);
    _mm256_storeu_si256((__m256i*)(out + i / 4),
                        _mm256_permute4x64_epi64(_mm256_packus_epi16(_mm256_maddubs_epi16(q0, m16), _mm256_maddubs_epi16(q1, m16)), 0xD8));
  }
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 5.0;
  const std::string dir = argc > 2 ? argv[2] : "/tmp";
  const size_t bytes = (size_t)(gib * (1u << 30)) / 4096 * 4096;
  const std::string path = dir + "/hostpack_probe.bk";
  {
    int fd = open(path.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644);
    std::vector<uint8_t> blk(1 << 24);
    for (size_t i = 0; i < blk.size(); i++) blk[i] = (uint8_t)((i * 2654435761u >> 13) % 7);
    for (size_t o = 0; o < bytes; o += blk.size()) if (write(fd, blk.data(), std::min(blk.size(), bytes - o)) < 0) return 1;
    close(fd);
  }
  int fd = open(path.c_str(), O_RDONLY);
  const uint8_t* in = (const uint8_t*)mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
  uint8_t* out = (uint8_t*)aligned_alloc(4096, bytes / 2);
  memset(out, 0, bytes / 2);
  for (int T : {4, 8, 16, 32, 64}) {
    for (int mode = 0; mode < 3; mode++) {
      double best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
          th.emplace_back([&, t]() {
            const size_t lo = bytes / 4096 * t / T * 4096, hi = bytes / 4096 * (t + 1) / T * 4096;
            if (mode == 0) { uint8_t acc = 0; for (size_t o = lo; o < hi; o += 64) acc ^= in[o]; volatile uint8_t sink = acc; (void)sink; }
            else if (mode == 1) pack4(in + lo, out + lo / 2, hi - lo);
            else pack2(in + lo, out + lo / 4, hi - lo);
          });
        for (auto& x : th) x.join();
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      }
      printf("%2d threads  %-28s %7.1f ms  %6.1f GB/s of input\n", T, mode == 0 ? "touch (one byte per line)" : mode == 1 ? "4 bits per genotype" : "2 bits per genotype (table)", best, bytes / best / 1e6);
    }
  }
  unlink(path.c_str());
  return 0;
}
