// Microbenchmark: cycles per v_mfma_i32_32x32x32_i8 with 5 VALU per MFMA plus LDS reads of different widths
// (one read instruction every RD_EVERY MFMAs), 4 waves per CU sharing the LDS, 1 wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 ubench_mfma_lds.hip -o ubench_mfma_lds.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// WIDTH: 0 none, 1 b32, 2 b64, 4 b128.  SRC3: 1 -> one of the five perms has three distinct VGPR sources
template <int WIDTH, int RD_EVERY, int SRC3, int NV>
__global__ __launch_bounds__(256) void k(const uint32_t* in, int* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint32_t sh[4][1024];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = lane; i < 1024; i += 64) sh[wv][i] = in[i & 255] + i;
  __syncthreads();
  uint32_t x0 = in[threadIdx.x], x1 = x0 * 3 + 1, x2 = x0 ^ 0x5555, x3 = x0 + 77;
  v4i a = {(int)x0, (int)x1, (int)x2, (int)x3}, b = {(int)x1, (int)x2, (int)x3, (int)x0};
  constexpr int NACC = 16;
  v16i acc[NACC];
  for (int t = 0; t < NACC; t++) for (int i = 0; i < 16; i++) acc[t][i] = 0;
  uint32_t y[8] = {x0, x1, x2, x3, x0 + 1, x1 + 1, x2 + 1, x3 + 1};
  const uint32_t addr = (uint32_t)(uintptr_t)&sh[wv][(lane >> 5) * 64];  // 2 distinct addresses per wave (broadcast)
  v4i d[NACC / 2];
  for (int t = 0; t < NACC / 2; t++) d[t] = a;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < NACC; t++) {
      acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[t], 0, 0, 0);
      if (WIDTH && (t % RD_EVERY) == 0) {
        const int q = (t / RD_EVERY) % (NACC / 2);
        if (WIDTH == 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[q]) : "v"(addr), "n"(16 * (t % 8)));
        else if (WIDTH == 2) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*(long long*)&d[q]) : "v"(addr), "n"(16 * (t % 8)));
        else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d[q][0]) : "v"(addr), "n"(16 * (t % 8)));
      }
#pragma unroll
      for (int v = 0; v < NV; v++) {
        if (SRC3 && v == 0) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(y[t & 7]) : "v"(x1), "v"(x2), "v"(y[(t + 3) & 7]));
        else asm volatile("v_perm_b32 %0, %1, %0, %0" : "+v"(y[v & 7]) : "s"(0x10203u));
      }
    }
    if (WIDTH) {
      asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
      for (int t = 0; t < NACC / 2; t++) y[t & 7] ^= (uint32_t)d[t][0];
    }
  }
  int s = 0;
  for (int t = 0; t < NACC; t++) for (int i = 0; i < 16; i++) s += acc[t][i];
  for (int v = 0; v < 8; v++) s += y[v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int WIDTH, int RD_EVERY, int SRC3, int NV>
void run(const char* name, uint32_t* din, int* dout) {
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<WIDTH, RD_EVERY, SRC3, NV><<<256, 256>>>(din, dout, 100);
  hipEventRecord(e0);
  k<WIDTH, RD_EVERY, SRC3, NV><<<256, 256>>>(din, dout, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s NV=%d: %8.3f ms  %7.2f ns per MFMA\n", name, NV, ms, ms * 1e6 / ((double)iters * 16));
}

int main() {
  uint32_t* din; int* dout;
  hipMalloc(&din, 1024 * 4); hipMalloc(&dout, 2048 * 256 * 4);
  hipMemset(din, 1, 1024 * 4);
  run<0, 1, 0, 5>("no LDS", din, dout);
  run<0, 1, 0, 4>("no LDS", din, dout);
  run<0, 1, 1, 5>("no LDS, one 3-VGPR perm", din, dout);
  run<4, 2, 0, 5>("b128 every 2 MFMA", din, dout);
  run<4, 1, 0, 5>("b128 every MFMA", din, dout);
  run<2, 1, 0, 5>("b64 every MFMA", din, dout);
  run<1, 1, 0, 5>("b32 every MFMA", din, dout);
  run<4, 2, 1, 5>("b128/2 + 3-VGPR perm", din, dout);
  run<4, 2, 1, 4>("b128/2 + 3-VGPR perm", din, dout);
  return 0;
}
