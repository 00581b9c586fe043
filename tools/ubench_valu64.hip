// Issue cost of the 64-bit VALU instructions a register fold could use, one wave per SIMD.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int KIND>
__global__ __launch_bounds__(256, 1) void k(const double* in, double* out, int iters, long long* clk) {
  double a[8], w = in[threadIdx.x];
  float f[8];
  unsigned long long q[8];
  for (int i = 0; i < 8; i++) { a[i] = in[threadIdx.x + i]; f[i] = (float)i + threadIdx.x; q[i] = i + threadIdx.x; }
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (KIND == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(a[(i + 1) & 7]));
        else if (KIND == 1) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
        else if (KIND == 2) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
        else if (KIND == 3) asm volatile("v_add_f64 %0, %1, %0" : "+v"(a[i]) : "v"(w));
        else if (KIND == 4) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(f[i]), "v"(f[(i + 1) & 7]) : "vcc");
        else if (KIND == 5) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(a[i]) : "v"(w));
        else if (KIND == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
        else if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(a[(i + 1) & 7]));
        else if (KIND == 8) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 7]));
        else if (KIND == 9) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
        else if (KIND == 10) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
        else if (KIND == 11) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
      }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + f[i] + (double)q[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int KIND>
void run(const char* name, double* din, double* dout, long long* dclk, int blocks_per_cu) {
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<256 * blocks_per_cu, 256>>>(din, dout, 100, dclk);
  hipEventRecord(e0);
  k<KIND><<<256 * blocks_per_cu, 256>>>(din, dout, iters, dclk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-16s waves/SIMD=%d: %7.3f ns per instruction per SIMD\n", name, blocks_per_cu, ms * 1e6 / ((double)iters * 32) / blocks_per_cu);
}

int main() {
  double *din, *dout; long long* dclk;
  hipMalloc(&din, 4096 * 8); hipMalloc(&dout, 2 * 256 * 256 * 8); hipMalloc(&dclk, 8);
  hipMemset(din, 0, 4096 * 8);
  for (int w = 1; w <= 2; w++) {
    run<0>("v_fma_f64", din, dout, dclk, w);
    run<1>("v_cvt_f64_f32", din, dout, dclk, w);
    run<2>("v_cvt_f64_i32", din, dout, dclk, w);
    run<3>("v_add_f64", din, dout, dclk, w);
    run<5>("v_mul_f64", din, dout, dclk, w);
    run<4>("v_mad_u64_u32", din, dout, dclk, w);
    run<6>("v_fma_f32", din, dout, dclk, w);
    run<7>("v_pk_fma_f32", din, dout, dclk, w);
    run<8>("v_cvt_i32_f32", din, dout, dclk, w);
    run<9>("v_mad_u32_u24", din, dout, dclk, w);
    run<10>("v_mul_lo_u32", din, dout, dclk, w);
    run<11>("v_lshl_add_u64", din, dout, dclk, w);
  }
  return 0;
}
