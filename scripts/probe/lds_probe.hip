// Probe: where does global_load_lds_dwordx4 put lane i's 16 bytes?  (expected: M0 base + 16 * lane)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const float __attribute__((address_space(1))) *gcp;
typedef __attribute__((address_space(3))) void *lp;
__global__ void k(const float *__restrict__ g, float *out) {
  __shared__ __attribute__((aligned(16))) float s[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) s[i] = -1.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lane i fetches global chunk (63 - i): a permutation, to see that the LDS slot follows the lane id
  gcp gp = (gcp)(g + (wave * 64 + (63 - lane)) * 4);
  __builtin_amdgcn_global_load_lds(gp, (lp)(s + wave * 512 + 256), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = s[i];
}
int main() {
  float *g, *o, h[2048], hg[1024];
  for (int i = 0; i < 1024; ++i) hg[i] = (float)i;
  hipMalloc(&g, sizeof(hg)); hipMalloc(&o, sizeof(h));
  hipMemcpy(g, hg, sizeof(hg), hipMemcpyHostToDevice);
  k<<<1, 128>>>(g, o);
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  for (int w = 0; w < 2; ++w) {
    printf("wave %d region base %d:\n", w, w * 512 + 256);
    for (int i = 0; i < 2048; ++i) if (h[i] >= 0 && (i < w * 512 + 256 + 24 || i > w*512+256+256-8) && i >= w*512 && i < (w+1)*512+256) printf(" s[%d]=%g", i, h[i]);
    printf("\n");
  }
  int cnt = 0; for (int i = 0; i < 2048; ++i) cnt += h[i] >= 0;
  printf("written floats: %d\n", cnt);
  return 0;
}
