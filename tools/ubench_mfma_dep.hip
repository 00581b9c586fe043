// Microbenchmark 2: MFMA operands are PRODUCED by v_perm_b32 (as in the real kernels), consumed DIST MFMAs later.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NVX, int DIST, int BFRESH>
__global__ __launch_bounds__(256) void k(const uint32_t* in, int* out, int iters, long long* clk) {
  uint32_t x0 = in[threadIdx.x], x1 = x0 * 3 + 1, x2 = x0 ^ 0x5555, x3 = x0 + 77;
  v4i fa[4], fbv[4];
  for (int i = 0; i < 4; i++) { fa[i] = (v4i){(int)x0 + i, (int)x1, (int)x2, (int)x3}; fbv[i] = (v4i){(int)x1, (int)x2 + i, (int)x3, (int)x0}; }
  v16i acc[10];
  for (int t = 0; t < 10; t++) for (int i = 0; i < 16; i++) acc[t][i] = 0;
  uint32_t y[8] = {x0, x1, x2, x3, x0 + 1, x1 + 1, x2 + 1, x3 + 1};
  uint32_t c0 = x0 & 0x03030303, c1 = x1 & 0x03030303;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < 10; t++) {
      // produce the A operand of the MFMA that runs DIST steps later
      const int slot = (t + DIST) & 3;
#pragma unroll
      for (int q = 0; q < 4; q++) fa[slot][q] = (int)__builtin_amdgcn_perm(0u, 0x00010101u + q + t, c0 + it);
      if (BFRESH) {
#pragma unroll
        for (int q = 0; q < 4; q++) fbv[slot][q] = (int)__builtin_amdgcn_perm(0u, 0x000100FFu + q + t, c1 + it);
      }
#pragma unroll
      for (int v = 0; v < NVX; v++) asm volatile("v_and_b32 %0, 0x3030303, %0" : "+v"(y[v & 7]));
      acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[t & 3], fbv[t & 3], acc[t], 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int t = 0; t < 10; t++) for (int i = 0; i < 16; i++) s += acc[t][i];
  for (int v = 0; v < 8; v++) s += y[v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int NVX, int DIST, int BFRESH>
void run(uint32_t* din, int* dout, long long* dclk) {
  int iters = 20000;
  k<NVX, DIST, BFRESH><<<256, 256>>>(din, dout, 100, dclk);
  k<NVX, DIST, BFRESH><<<256, 256>>>(din, dout, iters, dclk);
  (void)hipDeviceSynchronize();
  long long clk; (void)hipMemcpy(&clk, dclk, 8, hipMemcpyDeviceToHost);
  printf("perms A=4%s extraVALU=%d dist=%d : %6.2f cycles per MFMA (total VALU/MFMA = %d)\n", BFRESH ? " B=4" : "    ", NVX, DIST,
         (double)clk / (iters * 10.0), 4 + 4 * BFRESH + NVX);
}

int main() {
  uint32_t* din; int* dout; long long* dclk;
  (void)hipMalloc(&din, 1024 * 4); (void)hipMalloc(&dout, 2048 * 256 * 4); (void)hipMalloc(&dclk, 8);
  (void)hipMemset(din, 1, 1024 * 4);
  run<0, 0, 0>(din, dout, dclk); run<0, 1, 0>(din, dout, dclk); run<0, 2, 0>(din, dout, dclk);
  run<2, 0, 0>(din, dout, dclk); run<2, 1, 0>(din, dout, dclk); run<2, 2, 0>(din, dout, dclk);
  run<0, 0, 1>(din, dout, dclk); run<0, 1, 1>(din, dout, dclk); run<0, 2, 1>(din, dout, dclk);
  return 0;
}
