// First touch of a fresh anonymous buffer by a team of copying threads (what a download into a result matrix the caller has
// just allocated pays), with and without MADV_HUGEPAGE.   g++ -O3 -pthread tools/thp_probe.cpp -o /tmp/thp_probe && /tmp/thp_probe [MB]
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 600) << 20;
  std::vector<char> src(bytes, 1);
  FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
  char line[128] = "?";
  if (f) { (void)!fgets(line, sizeof(line), f); fclose(f); }
  printf("THP: %s", line);
  for (int mode = 0; mode < 3; mode++)
    for (int nt : {8, 16}) {
      char* dst = (char*)mmap(nullptr, bytes + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      char* al = (char*)(((uintptr_t)dst + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
      int rc = 0;
      if (mode == 1) rc = madvise(al, bytes, MADV_HUGEPAGE);
      if (mode == 2) rc = madvise(al, bytes, MADV_NOHUGEPAGE);
      const double t0 = now();
      std::vector<std::thread> th;
      for (int t = 0; t < nt; t++) th.emplace_back([=, &src] { memcpy(al + bytes * t / nt, src.data() + bytes * t / nt, bytes * (t + 1) / nt - bytes * t / nt); });
      for (auto& t : th) t.join();
      const double dt = now() - t0;
      printf("%-16s %2d threads: %7.2f ms  %6.1f GB/s (madvise rc %d)\n", mode == 0 ? "default" : mode == 1 ? "MADV_HUGEPAGE" : "MADV_NOHUGEPAGE", nt, dt * 1e3, bytes / dt / 1e9, rc);
      munmap(dst, bytes + (2 << 20));
    }
  return 0;
}
