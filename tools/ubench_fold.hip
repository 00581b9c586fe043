// Proxy for the weight-class PCA Gram: P accumulator tiles, per class `steps` K steps of P FP4 MFMAs (the first with
// C = 0), then a fold  out64 += w * (double)acc  of all P tiles (16 v_cvt_f64_f32 + 16 v_fma_f64 per tile).
// One wave per SIMD, no memory traffic: the compute ceiling of the design.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef double v16d __attribute__((ext_vector_type(16)));

template <int P>
__global__ __launch_bounds__(256, 1) void k(const uint32_t* in, const double* wtab, double* out, int classes, int steps) {
  uint32_t x0 = in[threadIdx.x], x1 = in[threadIdx.x + 256], x2 = in[threadIdx.x + 512], x3 = in[threadIdx.x + 768];
  v8i a = {(int)x0, (int)x1, (int)x2, (int)x3, 0, 0, 0, 0}, b = {(int)x1, (int)x2, (int)x3, (int)x0, 0, 0, 0, 0};
  v16d o[P];
  for (int t = 0; t < P; t++) for (int i = 0; i < 16; i++) o[t][i] = 0;
  const int sc = 0x7f7f7f7f;
  for (int c = 0; c < classes; c++) {
    const double w = wtab[c & 1023];
    v16f acc[P];
    const v16f z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < P; t++) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, z, 4, 4, 0, sc, 0, sc);
    for (int s = 1; s < steps; s++) {
#pragma unroll
      for (int t = 0; t < P; t++) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[t], 4, 4, 0, sc, 0, sc);
    }
#pragma unroll
    for (int t = 0; t < P; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) o[t][i] = __builtin_fma((double)acc[t][i], w, o[t][i]);
  }
  double s = 0;
  for (int t = 0; t < P; t++) for (int i = 0; i < 16; i++) s += o[t][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int P>
void run(uint32_t* din, double* dw, double* dout, int steps) {
  int classes = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<P><<<256, 256>>>(din, dw, dout, 10, steps);
  hipEventRecord(e0);
  k<P><<<256, 256>>>(din, dw, dout, classes, steps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ns_class = ms * 1e6 / classes;
  printf("P=%d steps=%2d: %8.3f ms  %8.1f ns per class per wave = %6.1f ns per (pair tile, class) ; MFMA-only would be %6.1f ns\n", P,
         steps, ms, ns_class, ns_class / P, steps * 17.4);
}

int main() {
  uint32_t* din; double* dw; double* dout;
  hipMalloc(&din, 4096); hipMalloc(&dw, 8192); hipMalloc(&dout, 256 * 256 * 8);
  uint32_t h[1024]; double w[1024];
  for (int i = 0; i < 1024; i++) { h[i] = 0x24200242u * (i % 3); w[i] = 1.0 + i * 1e-3; }
  hipMemcpy(din, h, 4096, hipMemcpyHostToDevice); hipMemcpy(dw, w, 8192, hipMemcpyHostToDevice);
  for (int steps : {1, 2, 4, 8, 16}) run<8>(din, dw, dout, steps);
  for (int steps : {1, 4, 16}) run<6>(din, dw, dout, steps);
  return 0;
}
