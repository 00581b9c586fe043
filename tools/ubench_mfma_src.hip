// Microbenchmark: cost of VALU operand kinds next to v_mfma_i32_32x32x32_i8 (NV VALU per MFMA, 1 wave/SIMD):
// v_perm_b32 with the LUT in an SGPR vs in a VGPR, v_and_b32 with a literal / SGPR / VGPR mask, v_lshrrev inline.
//   hipcc --offload-arch=gfx950 -O3 -w ubench_mfma_src.hip -o ubench_mfma_src.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int KIND, int NV>
__global__ __launch_bounds__(256) void k(const uint32_t* in, int* out, int iters) {
  uint32_t x0 = in[threadIdx.x], x1 = x0 * 3 + 1, x2 = x0 ^ 0x5555, x3 = x0 + 77;
  v4i a = {(int)x0, (int)x1, (int)x2, (int)x3}, b = {(int)x1, (int)x2, (int)x3, (int)x0};
  constexpr int NACC = 16;
  v16i acc[NACC];
  for (int t = 0; t < NACC; t++) for (int i = 0; i < 16; i++) acc[t][i] = 0;
  uint32_t y[8] = {x0, x1, x2, x3, x0 + 1, x1 + 1, x2 + 1, x3 + 1};
  uint32_t vl = in[threadIdx.x + 256];  // "LUT" in a VGPR
  uint32_t sl = __builtin_amdgcn_readfirstlane(vl);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < NACC; t++) {
      acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; v++) {
        uint32_t& r = y[(t * NV + v) & 7];
        if (KIND == 0) asm volatile("v_perm_b32 %0, %1, %0, %0" : "+v"(r) : "s"(sl));
        else if (KIND == 1) asm volatile("v_perm_b32 %0, %1, %0, %0" : "+v"(r) : "v"(vl));
        else if (KIND == 2) asm volatile("v_and_b32 %0, 0x3030303, %0" : "+v"(r));
        else if (KIND == 3) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r) : "s"(sl));
        else if (KIND == 4) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r) : "v"(vl));
        else if (KIND == 5) asm volatile("v_lshrrev_b32 %0, 2, %0" : "+v"(r));
        else if (KIND == 6) asm volatile("v_perm_b32 %0, %1, %1, %0" : "+v"(r) : "s"(sl));  // SGPR twice, selector VGPR
        else if (KIND == 7) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(r) : "v"(vl), "v"(x2));  // 3 distinct VGPRs
      }
    }
  }
  int s = 0;
  for (int t = 0; t < NACC; t++) for (int i = 0; i < 16; i++) s += acc[t][i];
  for (int v = 0; v < 8; v++) s += y[v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int NV>
void run(const char* name, uint32_t* din, int* dout) {
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND, NV><<<256, 256>>>(din, dout, 100);
  hipEventRecord(e0);
  k<KIND, NV><<<256, 256>>>(din, dout, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s NV=%d: %8.3f ms  %7.2f ns per MFMA\n", name, NV, ms, ms * 1e6 / ((double)iters * 16));
}

#define ALL(KIND, NAME) run<KIND, 0>(NAME, din, dout); run<KIND, 3>(NAME, din, dout); run<KIND, 5>(NAME, din, dout); run<KIND, 6>(NAME, din, dout); run<KIND, 8>(NAME, din, dout);
int main() {
  uint32_t* din; int* dout;
  hipMalloc(&din, 1024 * 4); hipMalloc(&dout, 2048 * 256 * 4);
  hipMemset(din, 1, 1024 * 4);
  ALL(0, "perm sgpr-lut")
  ALL(1, "perm vgpr-lut")
  ALL(7, "perm 3 distinct vgpr")
  ALL(6, "perm sgpr,sgpr,vsel")
  ALL(2, "and literal")
  ALL(3, "and sgpr")
  ALL(4, "and vgpr")
  ALL(5, "lshrrev inline")
  return 0;
}
