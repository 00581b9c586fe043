// Host -> HBM strategies for a file in the page cache, timed phase by phase (MI355X box).
// build: hipcc --offload-arch=gfx950 -O2 -w tools/xfer_probe.hip -o tools/xfer_probe.bin -lpthread ; run: tools/xfer_probe.bin /tmp/x.bk 2000000000
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename F> static void par(size_t bytes, int T, F f) {
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) { size_t lo = bytes * t / T, hi = bytes * (t + 1) / T; th.emplace_back([=]() { f(lo, hi); }); }
  for (auto& t : th) t.join();
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
  const char* path = argv[1];
  size_t bytes = strtoull(argv[2], 0, 10);
  { FILE* f = fopen(path, "wb"); std::vector<char> z(1 << 24, 1); for (size_t o = 0; o < bytes; o += z.size()) fwrite(z.data(), 1, std::min(z.size(), bytes - o), f); fclose(f); }
  void* dev; CK(hipMalloc(&dev, bytes));
  hipStream_t st; CK(hipStreamCreate(&st));
  int fd = open(path, O_RDONLY);
  const int T = 16;
  auto pread_all = [&](char* dst) { par(bytes, T, [=](size_t lo, size_t hi) { size_t g = lo; while (g < hi) { ssize_t r = pread(fd, dst + g, hi - g, g); if (r <= 0) break; g += r; } }); };
  for (int rep = 0; rep < 2; rep++) {
    // S1/S3: anonymous buffer, threads pread into it, then one hipMemcpy
    double t0 = now();
    char* buf = (char*)mmap(0, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    double t1 = now();
    pread_all(buf);
    double t2 = now();
    CK(hipMemcpy(dev, buf, bytes, hipMemcpyHostToDevice));
    double t3 = now();
    CK(hipMemcpy(dev, buf, bytes, hipMemcpyHostToDevice));
    double t4 = now();
    printf("rep%d anon: pread(16 thr, fresh pages) %.1f GB/s, hipMemcpy first %.1f GB/s, again %.1f GB/s, total first-use %.1f GB/s\n", rep,
           bytes / (t2 - t1) / 1e9, bytes / (t3 - t2) / 1e9, bytes / (t4 - t3) / 1e9, bytes / (t3 - t0) / 1e9);
    // chunked: 256 MB pieces of the same buffer
    double t5 = now();
    for (size_t o = 0; o < bytes; o += (256u << 20)) CK(hipMemcpy((char*)dev + o, buf + o, std::min<size_t>(256u << 20, bytes - o), hipMemcpyHostToDevice));
    double t6 = now();
    printf("rep%d anon chunked 256MB: %.1f GB/s\n", rep, bytes / (t6 - t5) / 1e9);
    // pipelined: two 256-MB anonymous buffers, pread of chunk c+1 beside hipMemcpy of chunk c
    {
      size_t CH = 256u << 20;
      char* b2[2]; for (int k = 0; k < 2; k++) { b2[k] = (char*)mmap(0, CH, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); memset(b2[k], 0, CH); }
      double t7 = now();
      std::thread cp; int b = 0;
      for (size_t o = 0; o < bytes; o += CH, b ^= 1) {
        size_t len = std::min(CH, bytes - o);
        char* hb = b2[b];
        par(len, T, [=](size_t lo, size_t hi) { size_t g = lo; while (g < hi) { ssize_t r = pread(fd, hb + g, hi - g, o + g); if (r <= 0) break; g += r; } });
        if (cp.joinable()) cp.join();
        cp = std::thread([=]() { hipSetDevice(0); hipMemcpy((char*)dev + o, hb, len, hipMemcpyHostToDevice); });
      }
      cp.join();
      double t8 = now();
      printf("rep%d pipelined pread+hipMemcpy (2 x 256MB touched buffers): %.1f GB/s\n", rep, bytes / (t8 - t7) / 1e9);
      for (int k = 0; k < 2; k++) munmap(b2[k], CH);
    }
    munmap(buf, bytes);
    // S4: file mapping, parallel touch, one hipMemcpy
    double t9 = now();
    char* fm = (char*)mmap(0, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    par(bytes, T, [=](size_t lo, size_t hi) { volatile char x = 0; for (size_t o = lo; o < hi; o += 4096) x ^= fm[o]; });
    double t10 = now();
    CK(hipMemcpy(dev, fm, bytes, hipMemcpyHostToDevice));
    double t11 = now();
    printf("rep%d file mapping: touch %.1f GB/s, hipMemcpy %.1f GB/s, total %.1f GB/s\n", rep, bytes / (t10 - t9) / 1e9, bytes / (t11 - t10) / 1e9, bytes / (t11 - t9) / 1e9);
    munmap(fm, bytes);
    // S5: pinned ring: 8 x 32 MB hipHostMalloc slots, threads pread, hipMemcpyAsync
    {
      const int NS = 8; size_t SL = 32u << 20; char* sl[NS]; hipEvent_t ev[NS]; bool used[NS] = {0};
      double ta = now();
      for (int k = 0; k < NS; k++) { CK(hipHostMalloc((void**)&sl[k], SL, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); }
      double tb = now();
      int b = 0;
      for (size_t o = 0; o < bytes; o += SL, b = (b + 1) % NS) {
        size_t len = std::min(SL, bytes - o);
        if (used[b]) CK(hipEventSynchronize(ev[b]));
        char* hb = sl[b];
        par(len, 8, [=](size_t lo, size_t hi) { size_t g = lo; while (g < hi) { ssize_t r = pread(fd, hb + g, hi - g, o + g); if (r <= 0) break; g += r; } });
        CK(hipMemcpyAsync((char*)dev + o, hb, len, hipMemcpyHostToDevice, st)); CK(hipEventRecord(ev[b], st)); used[b] = true;
      }
      CK(hipStreamSynchronize(st));
      double tc = now();
      printf("rep%d pinned ring 8x32MB: alloc %.3f s, transfer %.1f GB/s\n", rep, tb - ta, bytes / (tc - tb) / 1e9);
      for (int k = 0; k < NS; k++) { hipHostFree(sl[k]); hipEventDestroy(ev[k]); }
    }
    // S6: hipHostRegister the anonymous buffer (after pread), then async copy
    {
      char* buf2 = (char*)mmap(0, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      double ta = now();
      pread_all(buf2);
      double tb = now();
      CK(hipHostRegister(buf2, bytes, hipHostRegisterDefault));
      double tc = now();
      CK(hipMemcpyAsync(dev, buf2, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
      double td = now();
      CK(hipHostUnregister(buf2));
      double te = now();
      printf("rep%d register: pread %.1f GB/s, register %.3f s, copy %.1f GB/s, unregister %.3f s, total %.1f GB/s\n", rep, bytes / (tb - ta) / 1e9, tc - tb, bytes / (td - tc) / 1e9, te - td, bytes / (te - ta) / 1e9);
      munmap(buf2, bytes);
    }
  }
  close(fd); unlink(path);
  return 0;
}
