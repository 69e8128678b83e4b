// Ceiling probe for the bf16-pipe tile loop of gemm_f32.hip:gemm256_bx_kernel: the SAME per-K-tile
// instruction mix -- 96 v_mfma_f32_32x32x16_bf16 on a 4 x 4 grid of 32 x 32 accumulator tiles (256 accumulator registers, one
// wave per SIMD), six partial products per tile in the product kernel's order, 24 conflict-free ds_read_b128 operand fragments
// -- with NOTHING else: the operand pieces sit in LDS once (no global -> LDS requests, no barrier, no chain flush, no C
// traffic).  What this loop sustains at the clock the chip holds under it is what the product kernel could reach if all of
// its data movement were free.  Operands are exact three-way bf16 splits of fp32 values drawn like the workloads: N(0,1), and
// "relu" (half of the values zero, as the masked factors of the benchmark).   Build: hipcc -O3 --offload-arch=gfx950
// -DSHAPE16: the product kernel's second form (round 6) -- 192 v_mfma_f32_16x16x32_bf16 on 8 x 8 accumulators of 16 x 16, two
// partial products fused per instruction ([a2|a0] x [b0|b2], [a1|a1] x [b1|b0], [a0|a0] x [b1|b0]), 40 fragment reads per K tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS: per wave 24 fragments of 1 KB: [A: 4 row tiles][3 pieces], [B: 4 col tiles][3 pieces]; lane l reads bytes 16 l .. 16 l + 15
__global__ __launch_bounds__(256) void bare_loop(const uint4 *__restrict__ img, float *__restrict__ out, long iters,
                                                 unsigned long long *__restrict__ clocks) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint4 *l4 = reinterpret_cast<uint4 *>(lds);
  for (int i = tid; i < 4 * 24 * 64; i += 256) l4[i] = img[i];
  __syncthreads();
#if defined(SHAPE16)
  // fragment of 16 rows (tile t16 of 8, halves of the 32-row images) and a combination of two pieces: lane (r16, kb) reads k half
  // kb & 1 of the piece kb >> 1 selects
  const int r16 = lane & 15, kb = lane >> 4;
  const unsigned char *base16 = lds + wave * 24 * 1024 + ((kb & 1) * 32 + r16) * 16;
  auto frag = [&](int oper, int t16, int p_first, int p_second) __attribute__((always_inline)) -> bf16x8 {
    const int piece = (kb >> 1) ? p_second : p_first;
    return *reinterpret_cast<const bf16x8 *>(base16 + ((oper * 12 + 3 * (t16 >> 1)) * 1024) + piece * 1024 + (t16 & 1) * 256);
  };
  f32x4 acc[64];
#pragma unroll
  for (int t = 0; t < 64; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (long it = 0; it < iters; ++it) {
    bf16x8 b[8][2];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      b[ct][0] = frag(1, ct, 0, 2);
      b[ct][1] = frag(1, ct, 1, 0);
    }
    // (software-pipelined like the product's loop: the A fragments of row tile rt + 1 are read before the MFMAs of row tile rt)
    bf16x8 an0 = frag(0, 0, 2, 0), an1 = frag(0, 0, 1, 1), an2 = frag(0, 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) {
      const bf16x8 a0 = an0, a1 = an1, a2 = an2;
      if (rt < 7) {
        an0 = frag(0, rt + 1, 2, 0); an1 = frag(0, rt + 1, 1, 1); an2 = frag(0, rt + 1, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) acc[8 * rt + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b[ct][0], acc[8 * rt + ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) acc[8 * rt + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b[ct][1], acc[8 * rt + ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) acc[8 * rt + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b[ct][1], acc[8 * rt + ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if ((it & 4095) == 4095) {
#pragma unroll
      for (int t = 0; t < 64; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] *= 1.f / 65536.f;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 64; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) s += acc[t][e];
#else
  const unsigned char *mine = lds + wave * 24 * 1024 + lane * 16;
  f32x16 acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (long it = 0; it < iters; ++it) {
    bf16x8 a[4][3], b[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a[i][p] = *reinterpret_cast<const bf16x8 *>(mine + (3 * i + p) * 1024);
        b[i][p] = *reinterpret_cast<const bf16x8 *>(mine + (12 + 3 * i + p) * 1024);
      }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x16 c = acc[4 * i + j];
        // smallest partial products first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi  (pieces 0 = hi, 1 = mid, 2 = lo)
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
        acc[4 * i + j] = c;
      }
    // keep the accumulators bounded (fp32 overflow would change the data the pipe sees): scale back every 4096 K tiles
    if ((it & 4095) == 4095) {
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] *= 1.f / 65536.f;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[t][e];
#endif
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

static unsigned short bf16_rne(float f) {
  unsigned u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
#if defined(SHAPE16)
  printf("v_mfma_f32_16x16x32_bf16 form (192 instructions, 40 fragment reads per K tile)\n");
#endif
  for (int kind = 0; kind < 3; ++kind) {   // 0: N(0,1)   1: relu-like (half zeros)   2: all zeros (the clock without data toggling)
    std::vector<unsigned short> img(4 * 24 * 64 * 8);
    srand(1 + kind);
    auto gauss = []() {
      double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
      return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
    };
    for (int w = 0; w < 4; ++w)
      for (int f = 0; f < 8; ++f)        // 4 A + 4 B fragments, three pieces each
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 8; ++e) {
            float v = kind == 2 ? 0.f : gauss();
            if (kind == 1 && (rand() & 1)) v = 0.f;
            const unsigned short h = bf16_rne(v);
            const float r = v - bf16_f(h);
            const unsigned short m = bf16_rne(r);
            const unsigned short lo = bf16_rne(r - bf16_f(m));
            const unsigned short p[3] = {h, m, lo};
            for (int q = 0; q < 3; ++q) img[(((size_t)(w * 24 + 3 * f + q) * 64) + l) * 8 + e] = p[q];
          }
    uint4 *dimg; float *dout; unsigned long long *dclk;
    hipMalloc(&dimg, img.size() * 2); hipMalloc(&dout, 256 * 256 * 4); hipMalloc(&dclk, 256 * 16);
    hipMemcpy(dimg, img.data(), img.size() * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(bare_loop), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    bare_loop<<<256, 256, 96 * 1024>>>(dimg, dout, 20000, dclk);   // warm-up + rate estimate
    hipDeviceSynchronize();
    hipEventRecord(e0);
    bare_loop<<<256, 256, 96 * 1024>>>(dimg, dout, 200000, dclk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const long iters = (long)(200000.0 * seconds * 1e3 / ms);
    hipEventRecord(e0);
    bare_loop<<<256, 256, 96 * 1024>>>(dimg, dout, iters, dclk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[512]; hipMemcpy(clk, dclk, sizeof(clk), hipMemcpyDeviceToHost);
    double ghz = 0; for (int i = 0; i < 256; ++i) ghz += (double)clk[2 * i] / ((double)clk[2 * i + 1] * 10.0); ghz /= 256;   // s_memrealtime: 100 MHz
    const double flops = (double)iters * 96.0 * 32768.0 * 4.0 * 256.0;   // per K tile and wave 96 MFMAs of 32 x 32 x 16 x 2
    const char *names[3] = {"N(0,1)", "relu-like (half zeros)", "zeros"};
    printf("%-24s %.2f s: %.0f TFLOP/s issued bf16 = %.1f TFLOP/s of fp32 work (6 partial products) = %.3f of 2516.6/6; "
           "%.2f cycles per MFMA at %.3f GHz (s_memtime / s_memrealtime)\n",
           names[kind], ms / 1e3, flops / ms / 1e9, flops / ms / 1e9 / 6.0, flops / ms / 1e9 / 2516.6,
           (ms * 1e-3) * ghz * 1e9 / ((double)iters * 96.0), ghz);   // (per 32768-flop MFMA; SHAPE16: per pair of its instructions)
    hipFree(dimg); hipFree(dout); hipFree(dclk);
  }
  return 0;
}
