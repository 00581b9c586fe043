// Microbenchmark: cycles per v_mfma_i32_32x32x32_i8 when NV VALU instructions (v_perm_b32 or v_and_b32)
// are issued per MFMA by the same wave, at 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NV, int KIND, int NACC>
__global__ __launch_bounds__(256) void k(const uint32_t* in, int* out, int iters, long long* clk) {
  uint32_t x0 = in[threadIdx.x], x1 = x0 * 3 + 1, x2 = x0 ^ 0x5555, x3 = x0 + 77;
  v4i a = {(int)x0, (int)x1, (int)x2, (int)x3}, b = {(int)x1, (int)x2, (int)x3, (int)x0};
  v16i acc[NACC];
  for (int t = 0; t < NACC; t++) for (int i = 0; i < 16; i++) acc[t][i] = 0;
  uint32_t y[8] = {x0, x1, x2, x3, x0 + 1, x1 + 1, x2 + 1, x3 + 1};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < NACC; t++) {
      acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; v++) {
        if (KIND == 0) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(y[v & 7]) : "v"(x1), "v"(x2));
        else if (KIND == 1) asm volatile("v_and_b32 %0, 0x3030303, %0" : "+v"(y[v & 7]));
        else asm volatile("v_lshrrev_b32 %0, 2, %0" : "+v"(y[v & 7]));
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int t = 0; t < NACC; t++) for (int i = 0; i < 16; i++) s += acc[t][i];
  for (int v = 0; v < 8; v++) s += y[v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int NV, int KIND, int NACC>
void run(const char* name, int blocks_per_cu, uint32_t* din, int* dout, long long* dclk) {
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int grid = 256 * blocks_per_cu;
  k<NV, KIND, NACC><<<grid, 256>>>(din, dout, 100, dclk);
  hipEventRecord(e0);
  k<NV, KIND, NACC><<<grid, 256>>>(din, dout, iters, dclk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long clk; hipMemcpy(&clk, dclk, 8, hipMemcpyDeviceToHost);
  double mfma_per_wave = (double)iters * NACC;
  // s_memtime ticks at 100 MHz on gfx9xx? report both
  double ns_per_mfma_wave = ms * 1e6 / mfma_per_wave;
  printf("%-8s NV=%2d nacc=%d waves/SIMD=%d : %8.3f ms  %7.2f ns per MFMA per wave  -> %6.2f ns per MFMA per SIMD  (memtime %lld)\n",
         name, NV, NACC, blocks_per_cu, ms, ns_per_mfma_wave, ns_per_mfma_wave / blocks_per_cu, clk);
}

int main() {
  uint32_t* din; int* dout; long long* dclk;
  hipMalloc(&din, 1024 * 4); hipMalloc(&dout, 2048 * 256 * 4); hipMalloc(&dclk, 8);
  hipMemset(din, 1, 1024 * 4);
  for (int w = 1; w <= 2; w++) {
    run<0, 0, 10>("perm", w, din, dout, dclk);
    run<2, 0, 10>("perm", w, din, dout, dclk);
    run<4, 0, 10>("perm", w, din, dout, dclk);
    run<5, 0, 10>("perm", w, din, dout, dclk);
    run<6, 0, 10>("perm", w, din, dout, dclk);
    run<7, 0, 10>("perm", w, din, dout, dclk);
    run<8, 0, 10>("perm", w, din, dout, dclk);
    run<12, 0, 10>("perm", w, din, dout, dclk);
    run<6, 1, 10>("and", w, din, dout, dclk);
    run<12, 1, 10>("and", w, din, dout, dclk);
    run<6, 2, 10>("shr", w, din, dout, dclk);
    run<12, 2, 10>("shr", w, din, dout, dclk);
  }
  return 0;
}
