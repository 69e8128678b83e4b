// Probe: back-to-back dependent kernel launches in one stream - time per launch vs grid size / work
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void empty_k(float *p, int work) {
  if (work) {
    float a = p[blockIdx.x * 256 + threadIdx.x];
    for (int i = 0; i < work; ++i) a = a * 1.0001f + 0.5f;
    p[blockIdx.x * 256 + threadIdx.x] = a;
  }
}
int main() {
  float *p; hipMalloc(&p, 1 << 24); hipMemset(p, 0, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 20000;
  for (int grid : {1, 160, 320}) for (int work : {0, 1, 2000}) {
    for (int i = 0; i < 100; ++i) empty_k<<<grid, 256>>>(p, work);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < N; ++i) empty_k<<<grid, 256>>>(p, work);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("grid %d work %d: %.2f us per launch\n", grid, work, ms * 1e3 / N);
  }
  return 0;
}
