// Peak-rate probe: back-to-back independent fp32 MFMAs, no memory traffic, one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
  float a = a0 + threadIdx.x, b = b0;
  if (KIND == 0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x4 acc[NACC * 4];
    for (int i = 0; i < NACC * 4; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < NACC * 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC * 4; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}
template <int KIND, int NACC>
void run(const char *name, int wgs_per_cu) {
  float *out; hipMalloc(&out, 256 * 256 * 8 * 4);
  const int iters = 20000, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND, NACC><<<grid, 256>>>(out, 100, 1.f, 2.f);
  hipEventRecord(e0);
  k<KIND, NACC><<<grid, 256>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // flops: 32x32x2: 4096 per MFMA, 4*NACC per iter; 16x16x4: 2048 per MFMA, 2*4*NACC per iter
  double flops = (double)grid * 4 * iters * (KIND == 0 ? 4.0 * NACC * 4096 : 8.0 * NACC * 2048);
  printf("%s NACC=%d wgs/cu=%d: %.2f ms  %.1f TFLOP/s\n", name, NACC, wgs_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<0, 1>("32x32x2", 1); run<0, 1>("32x32x2", 2); run<0, 2>("32x32x2", 1); run<0, 2>("32x32x2", 2);
  run<0, 4>("32x32x2", 1); run<0, 4>("32x32x2", 2); run<0, 16>("32x32x2", 1);
  run<1, 4>("16x16x4", 1); run<1, 4>("16x16x4", 2); run<1, 16>("16x16x4", 1);
  return 0;
}
