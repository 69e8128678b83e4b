// Probe: what does one in-kernel grid barrier + exchange of an n-vector cost, (a) over all 256 CUs, (b) over the 32
// workgroups of ONE XCD with the agent-scope protocol, (c) over one XCD with L2-local traffic only (no write-back /
// invalidate: plain stores, L2 atomics, sc1 loads)?  Decides the shape of a persistent tridiagonalisation for n <= 2048.
// Every spin is bounded: the grid always drains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Args {
  float *x;        // 2 buffers of n floats
  unsigned long long *xt;   // 2 buffers of n tagged elements
  int *counter;    // monotonic barrier counter
  int *err;        // err[0]: timeouts, err[1]: wrong sums, err[2..]: xcc ids seen (bitmask)
  int n, iters, mode, nwg;
};

__device__ __forceinline__ int xcc_id() {
  int v;
  __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15;
}

__global__ __launch_bounds__(256) void barrier_probe(Args a) {
  int wg = blockIdx.x;
  if (a.mode == 1 || a.mode == 2 || a.mode == 4 || a.mode == 6) {
    if ((blockIdx.x & 7) != 0) return;
    wg = blockIdx.x >> 3;
  }
  const int tid = threadIdx.x;
  if (tid == 0) atomicOr(a.err + 2, 1 << xcc_id());
  const int per = a.n / a.nwg;   // slice of this workgroup
  float acc = 0.f;
  __shared__ int dead;
  if (tid == 0) dead = 0;
  __syncthreads();
  for (int it = 0; it < a.iters; ++it) {
    float *buf = a.x + (it & 1) * a.n;
    if (a.mode >= 4) {   // tagged 64-bit elements: {value, iteration}; the readers poll the data itself
      unsigned long long *tb = a.xt + (size_t)(it & 1) * a.n;
      const float val = (float)((it & 1023) + 1);
      if (tid < per) {
        const unsigned long long w = ((unsigned long long)(unsigned)(it + 1) << 32) | __float_as_uint(val);
        __hip_atomic_store(tb + wg * per + tid, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      float s = 0.f;
      for (int i = tid; i < a.n; i += 256) {
        unsigned long long w;
        int spins = 0;
        do {
          w = __hip_atomic_load(tb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (++spins > 2000000) { atomicAdd(a.err, 1); dead = 1; break; }
        } while ((unsigned)(w >> 32) != (unsigned)(it + 1));
        s += __uint_as_float((unsigned)w);
      }
      acc += s;
      __syncthreads();
    } else if (a.mode == 3 || a.mode == 6) {   // all workgroups, no fences: agent-scope atomic stores / loads
      if (tid < per) __hip_atomic_store(buf + wg * per + tid, (float)((it & 1023) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(a.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int target = (it + 1) * a.nwg;
        int spins = 0;
        while (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (++spins > 2000000) { atomicAdd(a.err, 1); dead = 1; break; }
        }
      }
      __syncthreads();
      float s = 0.f;
      for (int i = tid; i < a.n; i += 256) s += __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += s;
    } else if (a.mode <= 1) {
      if (tid < per) buf[wg * per + tid] = (float)((it & 1023) + 1);
      __syncthreads();
      if (tid == 0) {
        __threadfence();
        __hip_atomic_fetch_add(a.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int target = (it + 1) * a.nwg;
        int spins = 0;
        while (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (++spins > 2000000) { atomicAdd(a.err, 1); dead = 1; break; }
        }
        __threadfence();
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      float s = 0.f;
      for (int i = tid; i < a.n; i += 256) s += buf[i];
      acc += s;
    } else {
      if (tid < per) buf[wg * per + tid] = (float)((it & 1023) + 1);
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(a.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int target = (it + 1) * a.nwg;
        int spins = 0;
        while (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (++spins > 2000000) { atomicAdd(a.err, 1); dead = 1; break; }
        }
      }
      __syncthreads();
      float s = 0.f;
      for (int i = tid; i < a.n; i += 256) s += __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += s;
    }
    if (dead) break;
    // check: every element read equals the iteration's value
    if (acc != (float)(a.n / 256) * (float)((it & 1023) + 1)) atomicAdd(a.err + 1, 1);
    acc = 0.f;
  }
}

int main() {
  Args a;
  const int n = 1280 * 2;
  hipMalloc(&a.x, 2 * n * sizeof(float));
  hipMalloc(&a.xt, 2 * n * 8);
  hipMemset(a.xt, 0, 2 * n * 8);
  hipMalloc(&a.counter, 256);
  hipMalloc(&a.err, 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 2; mode < 7; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(a.counter, 0, 256);
      hipMemset(a.err, 0, 256);
      hipMemset(a.x, 0, 2 * n * sizeof(float));
      a.n = n; a.iters = 4000; a.mode = mode; a.nwg = (mode == 0 || mode == 3 || mode == 5) ? 256 : 32;
      hipMemset(a.xt, 0, 2 * n * 8);
      hipEventRecord(e0);
      barrier_probe<<<256, 256>>>(a);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      int err[3];
      hipMemcpy(err, a.err, 12, hipMemcpyDeviceToHost);
      printf("mode %d: %.3f us per barrier+exchange (timeouts %d, wrong sums %d, xcc mask 0x%x)\n", mode, ms * 1e3 / a.iters, err[0], err[1], err[2]);
      fflush(stdout);
      if (err[0]) break;
    }
  }
  return 0;
}
