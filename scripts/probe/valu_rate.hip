// Issue rate of the VALU instructions the in-register bf16 split uses (gfx950): cycles per instruction and wave, one wave
// per SIMD, long dependent-free sequences.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, long long *cyc, float seed) {
  float a0 = seed, a1 = seed * 2, a2 = seed * 3, a3 = seed * 4, a4 = seed * 5, a5 = seed * 6, a6 = seed * 7, a7 = seed * 8;
  unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 64; ++it) {
    if (OP == 0) { REP64(__asm__ volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %6\n v_cvt_pk_bf16_f32 %2, %6, %7\n v_cvt_pk_bf16_f32 %3, %7, %4" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
    if (OP == 1) { REP64(__asm__ volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));) }
    if (OP == 2) { REP64(__asm__ volatile("v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_sub_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
    if (OP == 3) { REP64(__asm__ volatile("v_and_b32 %0, 0xffff0000, %4\n v_and_b32 %1, 0xffff0000, %5\n v_lshlrev_b32 %2, 16, %6\n v_lshlrev_b32 %3, 16, %7" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
    if (OP == 4) { REP64(__asm__ volatile("v_perm_b32 %0, %4, %5, %8\n v_perm_b32 %1, %5, %6, %8\n v_perm_b32 %2, %6, %7, %8\n v_perm_b32 %3, %7, %4, %8" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(0x07060302u));) }
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ __float_as_uint(a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  unsigned *out; long long *cyc, h[4];
  hipMalloc(&out, 1024); hipMalloc(&cyc, 64);
  const char *names[5] = {"v_cvt_pk_bf16_f32", "v_pk_add_f32", "v_sub_f32", "v_and/v_lshlrev", "v_perm_b32"};
  for (int op = 0; op < 5; ++op)
    for (int threads = 64; threads <= 256; threads += 192) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&]() {
        if (op == 0) k<0><<<1, threads>>>(out, cyc, 1.5f);
        if (op == 1) k<1><<<1, threads>>>(out, cyc, 1.5f);
        if (op == 2) k<2><<<1, threads>>>(out, cyc, 1.5f);
        if (op == 3) k<3><<<1, threads>>>(out, cyc, 1.5f);
        if (op == 4) k<4><<<1, threads>>>(out, cyc, 1.5f);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
      const double n = 64.0 * 64 * 4;
      printf("%-20s waves/CU=%d: %.2f shader-clock ticks per instruction (s_memtime units), %.3f ns per instruction\n", names[op], threads / 64, (double)h[0] / n, ms * 1e6 / n);
    }
  return 0;
}
