// Microbenchmark + exactness probe for v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands used as an
// exact small-integer contraction (genotype planes: 0.5, 1.0, +-2.0).   hipcc --offload-arch=gfx950 -O3
//  1. exactness: random nibble planes, accumulators started near 2^24, against an integer host sum;
//  2. rate: NACC independent accumulator tiles, NV VALU (v_and_b32 with a literal) per MFMA, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define FMT_FP4 4

__device__ __forceinline__ v16f mfma_fp4(v8i a, v8i b, v16f c, int sa, int sb) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FMT_FP4, FMT_FP4, 0, sa, 0, sb);
}

// one wave: D[32][32] = C + scale * A(32x64) * B(64x32); operand words as the wave holds them (4 dwords per lane)
__global__ void exact_kernel(const uint32_t* A, const uint32_t* B, const float* C, float* D, int sa, int sb) {
  int l = threadIdx.x;
  v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
  for (int i = 0; i < 4; i++) { a[i] = (int)A[l * 4 + i]; b[i] = (int)B[l * 4 + i]; }
  v16f c;
  for (int i = 0; i < 16; i++) c[i] = C[i * 64 + l];
  c = mfma_fp4(a, b, c, sa, sb);
  for (int i = 0; i < 16; i++) D[i * 64 + l] = c[i];
}

static double fp4_value(int nib) {
  static const double mag[8] = {0, 0.5, 1, 1.5, 2, 3, 4, 6};
  return (nib & 8) ? -mag[nib & 7] : mag[nib & 7];
}

template <int NV, int NACC>
__global__ __launch_bounds__(256) void rate_kernel(const uint32_t* in, float* out, int iters) {
  uint32_t x0 = in[threadIdx.x], x1 = in[threadIdx.x + 256], x2 = in[threadIdx.x + 512], x3 = in[threadIdx.x + 768];
  v8i a = {(int)x0, (int)x1, (int)x2, (int)x3, 0, 0, 0, 0}, b = {(int)x1, (int)x2, (int)x3, (int)x0, 0, 0, 0, 0};
  v16f acc[NACC];
  for (int t = 0; t < NACC; t++) for (int i = 0; i < 16; i++) acc[t][i] = 0;
  uint32_t y[8] = {x0, x1, x2, x3, x0 + 1, x1 + 1, x2 + 1, x3 + 1};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < NACC; t++) {
      acc[t] = mfma_fp4(a, b, acc[t], 0x7f7f7f7f, 0x7f7f7f7f);
#pragma unroll
      for (int v = 0; v < NV; v++) asm volatile("v_and_b32 %0, 0xf3f3f3f3, %0" : "+v"(y[v & 7]));
    }
  }
  float s = 0;
  for (int t = 0; t < NACC; t++) for (int i = 0; i < 16; i++) s += acc[t][i];
  for (int v = 0; v < 8; v++) s += (float)y[v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int NACC>
void run(uint32_t* din, float* dout, const char* what) {
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  rate_kernel<NV, NACC><<<256, 256>>>(din, dout, 100);
  hipEventRecord(e0);
  rate_kernel<NV, NACC><<<256, 256>>>(din, dout, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n = (double)iters * NACC;
  double ops = n * 1024 * 2.0 * 32 * 32 * 64;
  printf("%s NV=%2d nacc=%2d: %8.3f ms  %6.2f ns per MFMA per SIMD  %6.2f POP/s\n", what, NV, NACC, ms, ms * 1e6 / n,
         ops / (ms * 1e-3) / 1e15);
}

int main() {
  // ---- exactness ----
  uint32_t hA[256], hB[256];
  float hC[1024], hD[1024];
  uint32_t *dA, *dB; float *dC, *dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
  // genotype nibbles: dosage 0 -> 0xE, 1 -> 0x3, 2 -> 0x6, missing -> 0; planes v = &2, h = &1, d = &0xC
  const uint32_t gen[4] = {0xE, 0x3, 0x6, 0x0};
  const uint32_t masks[3] = {0x22222222u, 0x11111111u, 0xCCCCCCCCu};
  const char* pname[3] = {"v", "h", "d"};
  srand(1);
  int bad_total = 0;
  for (int trial = 0; trial < 12; trial++) {
    int pa = trial % 3, pb = (trial / 3) % 3;
    uint32_t GA[256], GB[256];
    for (int i = 0; i < 256; i++) {
      uint32_t wa = 0, wb = 0;
      for (int e = 0; e < 8; e++) { wa |= gen[rand() & 3] << (4 * e); wb |= gen[rand() & 3] << (4 * e); }
      GA[i] = wa; GB[i] = wb;
      hA[i] = wa & masks[pa]; hB[i] = wb & masks[pb];
    }
    // accumulators: a spread of magnitudes up to just under 2^24 units (unit = value of one product)
    double unit = fabs(fp4_value(masks[pa] & 7 ? masks[pa] & 7 : 4)) * fabs(fp4_value(masks[pb] & 7 ? masks[pb] & 7 : 4));
    for (int i = 0; i < 1024; i++) {
      long c = (i % 5 == 0) ? 0 : (i % 5 == 1) ? 16777216 - 100 - (rand() % 1000) : (i % 5 == 2) ? -(16777216 - 100 - (rand() % 1000))
               : (rand() % 16000000);
      hC[i] = (float)(c * unit);
    }
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, 4096, hipMemcpyHostToDevice);
    int sa = 0x7f7f7f7f, sb = 0x7f7f7f7f;
    exact_kernel<<<1, 64>>>(dA, dB, dC, dD, sa, sb);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    // host: lane l = (r, h) holds k = 32h + e, nibble e of its 4 dwords; C/D: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    int bad = 0;
    for (int reg = 0; reg < 16; reg++)
      for (int l = 0; l < 64; l++) {
        int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
        double s = hC[reg * 64 + l];
        for (int h = 0; h < 2; h++)
          for (int e = 0; e < 32; e++) {
            int na = (hA[(row + 32 * h) * 4 + e / 8] >> (4 * (e % 8))) & 15;
            int nb = (hB[(col + 32 * h) * 4 + e / 8] >> (4 * (e % 8))) & 15;
            s += fp4_value(na) * fp4_value(nb);
          }
        if ((double)hD[reg * 64 + l] != s) {
          if (bad < 4) printf("  mismatch plane %s x %s reg %d lane %d: device %.3f host %.3f (c %.3f)\n", pname[pa], pname[pb], reg, l,
                              hD[reg * 64 + l], s, hC[reg * 64 + l]);
          bad++;
        }
      }
    printf("exactness %s x %s: %d of 1024 differ\n", pname[pa], pname[pb], bad);
    bad_total += bad;
  }
  // scale bytes: 0x80 = 2.0, 0x7e = 0.5
  {
    for (int i = 0; i < 256; i++) { hA[i] = 0x11111111u; hB[i] = 0x22222222u; }   // 0.5 x 1.0 over 64 k = 32
    for (int i = 0; i < 1024; i++) hC[i] = 0;
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, 4096, hipMemcpyHostToDevice);
    exact_kernel<<<1, 64>>>(dA, dB, dC, dD, 0x80808080, 0x7f7f7f7f);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    printf("scale_a = 2.0: 64 x (0.5 x 1.0) = %.3f (expect 64)\n", hD[0]);
    exact_kernel<<<1, 64>>>(dA, dB, dC, dD, 0x80808080, 0x7e7e7e7e);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    printf("scale_a = 2.0, scale_b = 0.5: = %.3f (expect 32)\n", hD[0]);
    exact_kernel<<<1, 64>>>(dA, dB, dC, dD, 0, 0);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    printf("scale operands 0, 0: = %.6g (32 if the unscaled form is selected)\n", hD[0]);
  }
  printf("exactness: %s\n", bad_total ? "FAILED" : "ok");

  // ---- rate ----
  uint32_t* din; float* dout;
  hipMalloc(&din, 1024 * 4); hipMalloc(&dout, 256 * 256 * 4);
  uint32_t hin[1024];
  for (int i = 0; i < 1024; i++) {
    uint32_t w = 0;
    for (int e = 0; e < 8; e++) w |= gen[rand() & 3] << (4 * e);
    hin[i] = w & 0x22222222u;
  }
  hipMemcpy(din, hin, 4096, hipMemcpyHostToDevice);
  run<0, 15>(din, dout, "fp4");
  run<2, 15>(din, dout, "fp4");
  run<3, 15>(din, dout, "fp4");
  run<4, 15>(din, dout, "fp4");
  run<5, 15>(din, dout, "fp4");
  run<6, 15>(din, dout, "fp4");
  run<8, 15>(din, dout, "fp4");
  return bad_total != 0;
}
