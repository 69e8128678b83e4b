// Probe: is "plain stores inside an XCD, one sc1 (agent-scope) write-through at the hand-over, sc1 loads everywhere"
// coherent ACROSS XCDs -- in particular, does an sc1 load ever return a stale copy that the reader's own L2 kept from its
// earlier plain accesses?  Eight leader workgroups (one per XCD) pass a 64 KB buffer round robin.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(float *buf, int *flag, int *err, int n, int rounds) {
  const int x = blockIdx.x;
  if (x >= 8) return;
  const int tid = threadIdx.x;
  __shared__ int dead;
  if (tid == 0) dead = 0;
  __syncthreads();
  for (int i = x; i < rounds; i += 8) {
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < i)
        if (++spins > 3000000) { atomicAdd(err, 1); dead = 1; break; }
    }
    __syncthreads();
    if (dead) break;
    int bad = 0;
    for (int c = tid; c < n; c += 256) bad += __hip_atomic_load(buf + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (float)i;
    if (bad) atomicAdd(err + 1, bad);
    for (int c = tid; c < n; c += 256) buf[c] = (float)(i + 1000);          // plain store: dirty in this XCD's L2
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int c = tid; c < n; c += 256) s += buf[c];                            // plain load (L1 / L2)
    if (s == 12345.f) atomicAdd(err + 2, 1);
    for (int c = tid; c < n; c += 256) __hip_atomic_store(buf + c, (float)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
int main() {
  float *buf; int *ctl;
  const int n = 16384, rounds = 4000;
  hipMalloc(&buf, n * 4); hipMalloc(&ctl, 256);
  hipMemset(buf, 0, n * 4); hipMemset(ctl, 0, 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<<<256, 256>>>(buf, ctl, ctl + 8, n, rounds);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  int err[3]; hipMemcpy(err, ctl + 8, 12, hipMemcpyDeviceToHost);
  printf("%.2f us per hand-over; timeouts %d, stale elements %d\n", ms * 1e3 / rounds, err[0], err[1]);
  return 0;
}
