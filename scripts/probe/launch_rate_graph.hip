// Probe: a chain of dependent kernel launches replayed from a hipGraph vs launched one by one (us per kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void small_k(float *p, int work) {
  float a = p[blockIdx.x * 256 + threadIdx.x];
  for (int i = 0; i < work; ++i) a = a * 1.0001f + 0.5f;
  p[blockIdx.x * 256 + threadIdx.x] = a;
}
int main() {
  float *p; hipMalloc(&p, 1 << 24); hipMemset(p, 0, 1 << 24);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 2000, REP = 10;
  for (int grid : {1, 320}) for (int work : {0, 200}) {
    for (int i = 0; i < 100; ++i) small_k<<<grid, 256, 0, s>>>(p, work);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < REP; ++r) for (int i = 0; i < N; ++i) small_k<<<grid, 256, 0, s>>>(p, work);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const float plain = ms * 1e3f / (N * REP);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) small_k<<<grid, 256, 0, s>>>(p, work);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < REP; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("grid %d work %d: plain %.2f us, graph %.2f us per kernel\n", grid, work, plain, ms * 1e3f / (N * REP));
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
