// Which XCD does workgroup b run on, and on which CU?  (s_getreg_b32 XCC_ID / HW_ID on gfx950)
// build: hipcc --offload-arch=gfx950 -O2 tools/xcc_probe.hip -o tools/xcc_probe.bin ; run: tools/xcc_probe.bin <grid> <block> <lds_bytes>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(int* out, int spin, int lds) {
  extern __shared__ int sh[];
  if (lds > 0) sh[threadIdx.x] = threadIdx.x;
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = (int)(xcc & 0xf); out[2 * blockIdx.x + 1] = (int)hwid; }
}

int main(int argc, char** argv) {
  int grid = argc > 1 ? atoi(argv[1]) : 512, block = argc > 2 ? atoi(argv[2]) : 256, lds = argc > 3 ? atoi(argv[3]) : 0;
  int* d;
  hipMalloc(&d, sizeof(int) * 2 * grid);
  hipLaunchKernelGGL(probe, dim3(grid), dim3(block), lds, 0, d, 2000000, lds);
  std::vector<int> h(2 * grid);
  hipMemcpy(h.data(), d, sizeof(int) * 2 * grid, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < grid; b++) bad += h[2 * b] != b % 8;
  printf("grid %d block %d: %d workgroups NOT on XCD b %% 8\n", grid, block, bad);
  for (int b = 0; b < grid && b < 40; b++) printf("wg %3d xcc %d cu %2d se %d\n", b, h[2 * b], (h[2 * b + 1] >> 8) & 0xf, (h[2 * b + 1] >> 13) & 0x7);
  for (int b = 256; b < grid && b < 280; b++) printf("wg %3d xcc %d cu %2d se %d\n", b, h[2 * b], (h[2 * b + 1] >> 8) & 0xf, (h[2 * b + 1] >> 13) & 0x7);
  return 0;
}
