// Ways to get a 5 GB file (page cache warm) into HBM, timed on this box.
// build: hipcc --offload-arch=gfx950 -O2 tools/upload_pipeline_probe.hip -o tools/upload_pipeline_probe.bin -lpthread
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename F> static void par(int nth, size_t bytes, F f) {
  std::vector<std::thread> th;
  for (int t = 0; t < nth; t++) { size_t lo = bytes * t / nth, hi = bytes * (t + 1) / nth; th.emplace_back([=]() { f(lo, hi); }); }
  for (auto& t : th) t.join();
}
static void pread_all(int fd, uint8_t* b, size_t off, size_t lo, size_t hi) {
  while (lo < hi) { ssize_t g = pread(fd, b + lo, hi - lo, off + lo); if (g <= 0) abort(); lo += g; }
}

int main(int argc, char** argv) {
  const char* path = argv[1];
  size_t bytes = strtoull(argv[2], 0, 10);
  int NT = argc > 3 ? atoi(argv[3]) : 16;
  int fd = open(path, O_RDONLY);
  uint8_t* d; hipMalloc(&d, bytes);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  // A: mmap + touch + one copy
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    uint8_t* p = (uint8_t*)mmap(0, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    par(NT, bytes, [=](size_t lo, size_t hi) { volatile uint8_t a = 0; for (size_t o = lo & ~4095ul; o < hi; o += 4096) a ^= p[o]; });
    double t1 = now();
    hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
    double t2 = now(); munmap(p, bytes);
    printf("A mmap+touch+copy: touch %.1f ms copy %.1f ms -> %.1f GB/s\n", 1e3 * (t1 - t0), 1e3 * (t2 - t1), bytes / (t2 - t0) / 1e9);
  }
  // B: pread into malloc staging, then copy, chunks, NOT overlapped / overlapped
  for (int ovl = 0; ovl < 2; ovl++) for (size_t CH : {64ul << 20, 256ul << 20, 1024ul << 20}) {
    uint8_t* b[2] = {(uint8_t*)malloc(CH), (uint8_t*)malloc(CH)};
    double t0 = now(), tr = 0, tc = 0;
    if (!ovl) {
      for (size_t a = 0; a < bytes; a += CH) { size_t len = std::min(CH, bytes - a); double x = now(); par(NT, len, [=](size_t lo, size_t hi) { pread_all(fd, b[0], a, lo, hi); }); double y = now();
        hipMemcpyAsync(d + a, b[0], len, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double z = now(); tr += y - x; tc += z - y; }
    } else {
      par(NT, std::min(CH, bytes), [=](size_t lo, size_t hi) { pread_all(fd, b[0], 0, lo, hi); });
      int cur = 0;
      for (size_t a = 0; a < bytes; a += CH, cur ^= 1) { size_t len = std::min(CH, bytes - a), nx = a + len; std::thread ah;
        if (nx < bytes) ah = std::thread([=]() { par(NT, std::min(CH, bytes - nx), [=](size_t lo, size_t hi) { pread_all(fd, b[cur ^ 1], nx, lo, hi); }); });
        hipMemcpyAsync(d + a, b[cur], len, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); if (ah.joinable()) ah.join(); }
    }
    double t1 = now();
    printf("B pread->malloc %s chunk %zu MiB: %.1f GB/s (read %.1f ms copy %.1f ms)\n", ovl ? "overlapped" : "serial", CH >> 20, bytes / (t1 - t0) / 1e9, 1e3 * tr, 1e3 * tc);
    free(b[0]); free(b[1]);
  }
  // D: pread into a pinned ring, async DMA
  for (size_t CH : {16ul << 20, 64ul << 20}) for (int slots : {2, 4}) {
    double ta = now();
    uint8_t* pin; hipHostMalloc(&pin, CH * slots, hipHostMallocDefault);
    std::vector<hipEvent_t> ev(slots); for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    double t0 = now();
    size_t k = 0;
    for (size_t a = 0; a < bytes; a += CH, k++) {
      size_t len = std::min(CH, bytes - a); int sl = k % slots;
      if (k >= (size_t)slots) hipEventSynchronize(ev[sl]);
      uint8_t* b = pin + (size_t)sl * CH;
      par(NT, len, [=](size_t lo, size_t hi) { pread_all(fd, b, a, lo, hi); });
      hipMemcpyAsync(d + a, b, len, hipMemcpyHostToDevice, s); hipEventRecord(ev[sl], s);
    }
    hipStreamSynchronize(s);
    double t1 = now();
    printf("D pread->pinned ring %d x %zu MiB: %.1f GB/s (+ %.1f ms to pin)\n", slots, CH >> 20, bytes / (t1 - t0) / 1e9, 1e3 * (t0 - ta));
    hipHostFree(pin);
  }
  return 0;
}
