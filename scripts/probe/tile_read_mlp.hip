// How fast can ONE workgroup per CU (4 waves, one per SIMD) read a 256 x 256 fp32 tile of a matrix that misses every cache -- the
// read half of the tile kernel's last flush -- depending on how many bytes it keeps in flight?  A wave's vmcnt counter holds 64
// operations, so 4-byte loads cap a wave at 16 KB in flight and 16-byte loads at 64 KB.  Patterns (per wave, its 128 x 128 quarter):
//   A  4 phases of 64 global_load_dword    (lane -> 4 rows x 64 B, the 16x16x32 accumulator layout; the product's flush)
//   B  2 phases of 32 global_load_dwordx4  (lane -> 16 rows x 64 B, the transposed accumulator layout)
//   C  1 phase  of 64 global_load_dwordx4
//   D  4 phases of 16 global_load_dwordx4  (the transposed layout with one row of blocks in flight)
// Build: hipcc -O3 --offload-arch=gfx950 -o tile_read_mlp tile_read_mlp.hip ; run: ./tile_read_mlp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256, 1) void k(const float *__restrict__ C, long ldc, int tiles_per_wg, float *out) {
  extern __shared__ char lds[];   // 144 KB: one workgroup per CU like the tile kernel
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, kb = lane >> 4;
  float acc = 0.f;
  for (int t = 0; t < tiles_per_wg; ++t) {
    // tiles far apart: tile index -> (row block, column block) of a 40960-column matrix, a different one per workgroup and step
    const long tile = (long)t * gridDim.x + blockIdx.x;
    const long ti = tile / 160, tj = tile % 160;
    const float *base = C + (ti * 256 + wm * 128) * ldc + tj * 256 + wn * 128;
    if (PAT == 0) {
#pragma unroll 1
      for (int i = 0; i < 4; ++i) {
        float v[64];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            v[16 * j + e] = __builtin_nontemporal_load(base + (long)(32 * i + 16 * (e >> 3) + 4 * kb + (e & 3)) * ldc + 32 * j + 16 * ((e >> 2) & 1) + r16);
#pragma unroll
        for (int q = 0; q < 64; ++q) acc += v[q];
      }
    } else {
      constexpr int NPH = PAT == 1 ? 2 : PAT == 2 ? 1 : 4, PER = 64 / NPH;   // dwordx4 loads per phase
#pragma unroll 1
      for (int ph = 0; ph < NPH; ++ph) {
        f32x4 v[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          const int g = ph * PER + q;            // (i, j, tr, tc) of the 16 x 16 tile: 64 per wave
          const int i = g >> 4, j = (g >> 2) & 3, tr = (g >> 1) & 1, tc = g & 1;
          v[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + (long)(32 * i + 16 * tr + r16) * ldc + 32 * j + 16 * tc + 4 * kb));
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) acc += v[q][0] + v[q][1] + v[q][2] + v[q][3];
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  const long n = 40960, ldc = n;
  float *C, *out;
  hipMalloc(&C, n * n * 4); hipMalloc(&out, 256 * 256 * 4);
  hipMemset(C, 0, n * n * 4);
  const int tpw = 64;   // 64 tiles per workgroup x 256 workgroups = 16384 tiles of 256 KB = 4.3 GB per run: far beyond L2 + Infinity Cache
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[4] = {"A 4 x 64 dword", "B 2 x 32 dwordx4", "C 1 x 64 dwordx4", "D 4 x 16 dwordx4"};
  // (few workgroups: the regime of the product, where ~ 4 % of the CUs flush at any moment; 256: every CU at once = the HBM rate)
  for (int nwg : {8, 32, 256})
    for (int pat = 0; pat < 4; ++pat) {
      const int rep = nwg == 8 ? 0 : 1;
      auto launch = [&]() {
        if (pat == 0) k<0><<<nwg, 256, 144 * 1024>>>(C, ldc, tpw, out);
        else if (pat == 1) k<1><<<nwg, 256, 144 * 1024>>>(C, ldc, tpw, out);
        else if (pat == 2) k<2><<<nwg, 256, 144 * 1024>>>(C, ldc, tpw, out);
        else k<3><<<nwg, 256, 144 * 1024>>>(C, ldc, tpw, out);
      };
      if (rep == 0) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
      }
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%3d workgroups, %-18s: %7.2f us per 256 x 256 tile and CU = %6.1f GB/s per CU\n", nwg, names[pat], ms * 1e3 / tpw,
             tpw * 262144.0 / ms / 1e6);
    }
  return 0;
}
