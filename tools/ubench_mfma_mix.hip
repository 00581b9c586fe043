// Proxy for kernel designs: per 10 MFMAs, NVALU v_perm_b32 + NRD ds_read_b128 + NWR ds_write_b128, an optional workgroup
// barrier every 40 MFMAs; 4 waves per workgroup, one wave per SIMD, 256 workgroups.  ns per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench_mfma_mix.hip -o tools/ubench_mfma_mix.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NVALU, int NRD, int NWR, int BAR>
__global__ __launch_bounds__(256, 1) void k(const uint32_t* in, int* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 sh[2][48][64];  // 96 KB
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t x0 = in[threadIdx.x], x1 = x0 * 3 + 1, x2 = x0 ^ 0x5555, x3 = x0 + 77;
  v4i a = {(int)x0, (int)x1, (int)x2, (int)x3}, b = {(int)x1, (int)x2, (int)x3, (int)x0};
  v16i acc[10];
  for (int t = 0; t < 10; t++) for (int i = 0; i < 16; i++) acc[t][i] = 0;
  uint32_t y[8] = {x0, x1, x2, x3, x0 + 1, x1 + 1, x2 + 1, x3 + 1};
  uint4 d[8];
  for (int t = 0; t < 8; t++) d[t] = make_uint4(x0, x1, x2, x3);
  const uint32_t sl = __builtin_amdgcn_readfirstlane(in[300]);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
#pragma unroll
      for (int t = 0; t < 10; t++) {
        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[t], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < (NVALU + 9 - t) / 10; v++) asm volatile("v_perm_b32 %0, %1, %0, %0" : "+v"(y[(t + v) & 7]) : "s"(sl));
        if (t < NRD) d[t & 7] = sh[it & 1][(s * 10 + t + 13 * wv) % 48][lane];
        if (t >= 10 - NWR) sh[(it & 1) ^ 1][(s * 4 + wv * 24 + t) % 48][lane] = make_uint4(y[0], y[1], y[2], y[3]);
      }
#pragma unroll
      for (int t = 0; t < 8; t++) y[t] ^= d[t].x;
    }
    if (BAR) __syncthreads();
  }
  int s = 0;
  for (int t = 0; t < 10; t++) for (int i = 0; i < 16; i++) s += acc[t][i];
  for (int v = 0; v < 8; v++) s += y[v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NVALU, int NRD, int NWR, int BAR>
void run(const char* name, uint32_t* din, int* dout) {
  int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NVALU, NRD, NWR, BAR><<<256, 256>>>(din, dout, 50);
  hipEventRecord(e0);
  k<NVALU, NRD, NWR, BAR><<<256, 256>>>(din, dout, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %8.3f ms  %6.2f ns per MFMA\n", name, ms, ms * 1e6 / ((double)iters * 40));
}

int main() {
  uint32_t* din; int* dout;
  hipMalloc(&din, 4096); hipMalloc(&dout, 256 * 256 * 4);
  hipMemset(din, 1, 4096);
  run<0, 0, 0, 0>("MFMA only", din, dout);
  run<60, 0, 0, 0>("60 VALU / 10 MFMA (pairwise now)", din, dout);
  run<45, 0, 0, 0>("45 VALU", din, dout);
  run<30, 0, 0, 0>("30 VALU", din, dout);
  run<30, 6, 2, 0>("30 VALU + 6 rd + 2 wr b128", din, dout);
  run<30, 6, 2, 1>("30 VALU + 6 rd + 2 wr b128 + barrier/40", din, dout);
  run<60, 0, 0, 1>("60 VALU + barrier/40", din, dout);
  return 0;
}
