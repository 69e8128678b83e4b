// Householder tridiagonalisation of a matrix that FITS THE REGISTER FILE OF ONE XCD (n <= 1280) -- or, with a slower exchange,
// of the whole chip (n <= 2048, NWG = 256; <10, 2, 256> would spill 146 registers: a workgroup per CU holds n / 256 rows, exchanges by agent-scope stores): one persistent launch
// instead of the three dependent launches per column of sytrd.hip (the launch chain is 29 ms at n = 1280, 20x the
// arithmetic).  Same outputs, bit for bit the same conventions: d, e, tau, reflector j in A's dead upper-triangle row j.
//
// The 32 workgroups of ONE XCD hold the full symmetric matrix in registers, rows dealt round-robin (row r lives in
// workgroup r % 32): 512 threads = 8 waves, a wave owns up to RPW rows, a lane the columns 256 k + 4 l .. + 3 of them
// (n = 1280: 5 rows x 20 columns = 100 matrix registers per thread + 20 each for v and w).  Unblocked right-looking
// reduction, ONE exchange per column:
//   * every workgroup keeps a replica of the current row j (= column j: the matrix is stored in full), so the reflector
//     (norm, beta, tau, v) is computed redundantly by every wave -- same instructions, same data, bit-identical -- and
//     never broadcast;
//   * y = A v: a row's dot product ends inside its wave (DPP adds); the workgroup writes its <= 40 values of y, and
//     the owner of row j + 1 adds that row as it is BEFORE update j;
//   * exchange: plain stores, one L2 atomic per workgroup, readers poll the counter and fetch with sc1 loads (2.0 us per
//     exchange measured by scripts/probe/grid_barrier.hip; the textbook release/acquire fences cost 17 us because they
//     write back / invalidate the L2, and a barrier over all 8 XCDs 6.7 us);
//   * w = tau y - (tau^2 y.v / 2) v, the rank-2 update of the own rows and of the replica of row j + 1 -- which is then
//     the current row j + 1 in every workgroup: the next column starts without another exchange.
// Workgroups find themselves on one XCD because the dispatcher deals workgroup b to XCD b % 8: the launch has 256
// workgroups, 224 return at once.  The first exchange compares the XCC ids; should they ever differ the stores become
// agent-scope atomic stores (write-through), which is correct anywhere (4 us per exchange).  Every spin is bounded (2 s):
// the grid always drains.  Co-residency is decided once, atomically, by the arrival gate (device_utils.h:persist_arrive):
// an attempt that does not get all its workgroups resident within 2 s aborts WITHOUT having written anything, and the
// launcher has a second attempt queued behind it (returns at once when the first one ran).  When that aborts too, or an
// exchange stalls later, the kernel ORs PERSIST_TMO_SYTRD into the sticky failure word and the solve ends with
// info = VIVIT_INFO_PERSIST_TIMEOUT -- a status of its own, not the non-finite-input one (the host wrapper then repeats the
// solve on the launch chain when it still has the input).  One process per GPU: two processes that launch this kernel on
// the same card at the same moment can each hold part of XCD 0 and time out (VIVIT_SYTRD_PERSIST=0 / VIVIT_QR_PERSIST=0
// select the launch chains; tests/test_distributed_gpu.py does for its two ranks on one card).
#include <cstdlib>

#include "common.h"
#include "device_utils.h"
#include "eig_internal.h"

namespace vivit {

constexpr int TP_WG = 32;        // workgroups (= CUs of one XCD)
constexpr int TP_THREADS = 512;  // 8 waves; wave g owns the local rows g, g + 8, ..  (row r = w + 32 * local)

struct PersistWs {
  float *ybuf;     // [2][NP]
  float *rowbuf;   // [2][NP]
  int *counter;    // [32]: per attempt a (0 | 1) at 8 a: monotonic arrival counter, arrival-gate state
  int *xcc;        // [32] XCC id of each workgroup
  int *tmo;        // the sticky failure word (persist_timeout_word)
  int attempt;     // 0: first launch; 1: the retry behind it (runs only if attempt 0 aborted at its arrival gate)
  int fault;       // VIVIT_PERSIST_FAULT (tests)
};

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float ld_l2(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int KC, int RPW, int NWG>   // column chunks of 256: n <= 256 KC; rows per wave: n <= 8 NWG RPW; NWG = 32 (one XCD) or 256 (all)
__global__ __launch_bounds__(TP_THREADS) void trd_persist_kernel(float *__restrict__ A, int64_t lda, int n, SytrdWs ws, PersistWs pw) {
  if (NWG == TP_WG && (blockIdx.x & 7) != 0) return;
  // the retry runs only when the first attempt aborted at its gate (that verdict is final once attempt 0 has drained)
  if (pw.attempt == 1 && __hip_atomic_load(pw.counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != PERSIST_ABORT) return;
  int *const cnt = pw.counter + 8 * pw.attempt;
  constexpr int NP = 256 * KC;
  const int w = NWG == TP_WG ? blockIdx.x >> 3 : blockIdx.x;
  const int tid = threadIdx.x, g = tid >> 6, l = tid & 63, lane = l;
  __shared__ __attribute__((aligned(16))) float s_x[2][NP];   // replica of the current row, double-buffered
  __shared__ __attribute__((aligned(16))) float s_y[NP];      // gathered y
  __shared__ __attribute__((aligned(16))) float s_r[NP];      // gathered row j + 1 (before update j)
  __shared__ int s_flag[2];                                   // 0: slow (not one XCD), 1: dead (1: stalled exchange, 2: aborted at the gate)

  // ---- the own rows, full width, from the lower triangle
  f2 a[RPW][KC][2];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int r = w + NWG * (g + 8 * q);
#pragma unroll
    for (int k = 0; k < KC; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 256 * k + 4 * l + e;
        const bool ok = r < n && c < n;
        const int64_t idx = !ok ? 0 : (c <= r ? (int64_t)r * lda + c : (int64_t)c * lda + r);
        const float x = A[idx];
        a[q][k][e >> 1][e & 1] = ok ? x : 0.f;
      }
  }
  for (int c = tid; c < NP; c += TP_THREADS) s_x[0][c] = c < n ? A[(int64_t)c * lda] : 0.f;   // row 0 = column 0

  // ---- are we on one XCD?  (first exchange, the safe way)
  if (tid == 0) {
    int xcc;
    __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (NWG == TP_WG) __hip_atomic_store(pw.xcc + w, xcc & 15, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool go = persist_arrive(cnt, cnt + 1, NWG, ((pw.fault >> pw.attempt) & 1) ? 0ull : PERSIST_TIMEOUT_TICKS);
    const int dead = go ? 0 : 2;
    int slow = NWG != TP_WG;   // all XCDs: always the agent-scope stores
    const int x0 = __hip_atomic_load(pw.xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = 1; i < TP_WG && NWG == TP_WG; ++i) slow |= __hip_atomic_load(pw.xcc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != x0;
    s_flag[0] = slow;
    s_flag[1] = dead;
  }
  __syncthreads();
  const bool slow = s_flag[0] != 0;
  if (s_flag[1] == 2) {   // aborted at the gate: nothing has been written; the second abort fails the solve
    if (tid == 0 && w == 0 && pw.attempt == 1) {
      __hip_atomic_fetch_or(pw.tmo, PERSIST_TMO_SYTRD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ws.scal[2] = 1.f;
    }
    return;
  }

  for (int j = 0; j < n - 2 && !s_flag[1]; ++j) {
    const int par = j & 1;
    const float *xr = s_x[par];
    // ---- reflector of column j from the replica of row j (every wave for itself)
    float ss = 0.f;
    for (int c = j + 2 + lane; c < n; c += 64) ss = fmaf(xr[c], xr[c], ss);
    const float ssq = wave_sum_dpp(ss);
    const float alpha = xr[j + 1];
    float beta = alpha, tau = 0.f, sc = 0.f;
    if (ssq > 0.f) {
      beta = -copysignf(sqrt_nr(alpha * alpha + ssq), alpha);
      tau = (beta - alpha) * rcp_nr(beta);
      sc = rcp_nr(alpha - beta);
    }
    const int kmin = (j + 1) >> 8;   // column chunks below are dead (v = 0 there)
    f2 v[KC][2];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const float4 x4 = *reinterpret_cast<const float4 *>(xr + 256 * k + 4 * l);
      const int c = 256 * k + 4 * l;
      v[k][0][0] = c <= j ? 0.f : (c == j + 1 ? 1.f : x4.x * sc);
      v[k][0][1] = c + 1 <= j ? 0.f : (c + 1 == j + 1 ? 1.f : x4.y * sc);
      v[k][1][0] = c + 2 <= j ? 0.f : (c + 2 == j + 1 ? 1.f : x4.z * sc);
      v[k][1][1] = c + 3 <= j ? 0.f : (c + 3 == j + 1 ? 1.f : x4.w * sc);
    }
    if (w == (j & (NWG - 1))) {   // the owner of row j files the results
      if (g == 0) {
#pragma unroll
        for (int k = 0; k < KC; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = 256 * k + 4 * l + e;
            if (c > j && c < n) A[(int64_t)j * lda + c] = v[k][e >> 1][e & 1];
          }
      }
      if (tid == 0) { ws.d[j] = xr[j]; ws.e[j] = beta; ws.tau[j] = tau; }
    }
    // ---- y = A v on the own rows
    float *yb = pw.ybuf + (size_t)par * NP, *rb = pw.rowbuf + (size_t)par * NP;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int r = w + NWG * (g + 8 * q);
      f2 p2 = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (k >= kmin) {
          p2 = __builtin_elementwise_fma(a[q][k][0], v[k][0], p2);
          p2 = __builtin_elementwise_fma(a[q][k][1], v[k][1], p2);
        }
      const float p = wave_sum_dpp(p2[0] + p2[1]);
      if (l == 0 && r > j && r < n) {
        if (slow) __hip_atomic_store(yb + r, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else yb[r] = p;
      }
      // the owner of row j + 1 publishes it (state before update j)
      if (r == j + 1) {
#pragma unroll
        for (int k = 0; k < KC; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = 256 * k + 4 * l + e;
            if (slow) __hip_atomic_store(rb + c, a[q][k][e >> 1][e & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else rb[c] = a[q][k][e >> 1][e & 1];
          }
      }
    }
    // ---- exchange
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int target = (j + 2) * NWG;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { s_flag[1] = 1; break; }
    }
    __syncthreads();
    for (int c = tid; c < NP; c += TP_THREADS) {
      s_y[c] = (c > j && c < n) ? ld_l2(yb + c) : 0.f;
      s_r[c] = c < n ? ld_l2(rb + c) : 0.f;
    }
    __syncthreads();
    // ---- w = tau y - (tau^2 (y.v) / 2) v
    float yv = 0.f;
    for (int c = j + 1 + lane; c < n; c += 64) yv = fmaf(s_y[c], c == j + 1 ? 1.f : xr[c] * sc, yv);
    yv = wave_sum_dpp(yv);
    const float al = 0.5f * tau * tau * yv;
    f2 wv[KC][2];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const float4 y4 = *reinterpret_cast<const float4 *>(s_y + 256 * k + 4 * l);
      wv[k][0][0] = fmaf(tau, y4.x, -al * v[k][0][0]);
      wv[k][0][1] = fmaf(tau, y4.y, -al * v[k][0][1]);
      wv[k][1][0] = fmaf(tau, y4.z, -al * v[k][1][0]);
      wv[k][1][1] = fmaf(tau, y4.w, -al * v[k][1][1]);
    }
    // ---- rank-2 update of the own rows (dead rows: v_r = w_r = 0)
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int r = w + NWG * (g + 8 * q);
      const bool live = r > j && r < n;
      const int rc = live ? r : 0;
      const float vr = !live ? 0.f : (r == j + 1 ? 1.f : xr[rc] * sc);
      const float wr = !live ? 0.f : fmaf(tau, s_y[rc], -al * vr);
      const f2 nvr = {-vr, -vr}, nwr = {-wr, -wr};
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (k >= kmin) {
          a[q][k][0] = __builtin_elementwise_fma(nvr, wv[k][0], a[q][k][0]);
          a[q][k][0] = __builtin_elementwise_fma(nwr, v[k][0], a[q][k][0]);
          a[q][k][1] = __builtin_elementwise_fma(nvr, wv[k][1], a[q][k][1]);
          a[q][k][1] = __builtin_elementwise_fma(nwr, v[k][1], a[q][k][1]);
        }
    }
    // ---- the replica of row j + 1 receives the same update (v_{j+1} = 1): it is the next current row
    if (g == 0) {
      const float wj1 = fmaf(tau, s_y[j + 1], -al);
      float *xn = s_x[par ^ 1];
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        const float4 r4 = *reinterpret_cast<const float4 *>(s_r + 256 * k + 4 * l);
        float4 o;
        o.x = fmaf(-wj1, v[k][0][0], fmaf(-1.f, wv[k][0][0], r4.x));
        o.y = fmaf(-wj1, v[k][0][1], fmaf(-1.f, wv[k][0][1], r4.y));
        o.z = fmaf(-wj1, v[k][1][0], fmaf(-1.f, wv[k][1][0], r4.z));
        o.w = fmaf(-wj1, v[k][1][1], fmaf(-1.f, wv[k][1][1], r4.w));
        *reinterpret_cast<float4 *>(xn + 256 * k + 4 * l) = o;
      }
    }
    __syncthreads();
  }

  // ---- tail: the last 2 x 2 block.  Row n - 2 is the current replica; d[n-1] sits with the owner of row n - 1.
  const float *xr = s_x[(n - 2) & 1];
  if (w == ((n - 2) & (NWG - 1)) && tid == 0) {
    ws.d[n - 2] = xr[n - 2];
    ws.e[n - 2] = xr[n - 1];
    ws.e[n - 1] = 0.f;
    ws.tau[n - 2] = 0.f;
    ws.tau[n - 1] = 0.f;
  }
  if (tid == 0 && s_flag[1]) {   // a stalled exchange: fail the solve, with the status of its own
    ws.scal[2] = 1.f;
    __hip_atomic_fetch_or(pw.tmo, PERSIST_TMO_SYTRD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int r = w + NWG * (g + 8 * q);
    if (r == n - 1) {
#pragma unroll
      for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (256 * k + 4 * l + e == n - 1) ws.d[n - 1] = a[q][k][e >> 1][e & 1];
    }
  }
}

bool sytrd_persist_ok(int64_t n) {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("VIVIT_SYTRD_PERSIST");
    on = e ? atoi(e) : 1;   // 1: all sizes, 2: one-XCD sizes only (n <= 1280)
  }
  if (persist_override() == 0) return false;
  return on != 0 && n >= 64 && n <= (on == 2 ? 1280 : 2048) && device_cu_count() >= 256;
}

// workspace: 4 NP floats + 64 ints, carved from ws.vw (3 * 64 * n floats)
int sytrd_persist_launch(float *A, int64_t n, int64_t lda, const SytrdWs &ws, hipStream_t stream) {
  const int kc0 = (int)cdiv(n, 256);
  const int kc = kc0 <= 6 ? kc0 : (kc0 + 1) & ~1;   // the instantiated widths: 1..6, 8
  const int64_t NP = 256 * (int64_t)kc;
  PersistWs pw;
  pw.ybuf = ws.vw;
  pw.rowbuf = ws.vw + 2 * NP;
  pw.counter = reinterpret_cast<int *>(ws.vw + 4 * NP);
  pw.xcc = pw.counter + 32;
  pw.tmo = persist_timeout_word(stream);
  pw.fault = persist_fault();
  if (!pw.tmo) return VIVIT_E_LAUNCH;
  if (4 * NP + 64 > 3 * 64 * n) return VIVIT_E_WORKSPACE;
  if (hipMemsetAsync(pw.counter, 0, 64 * sizeof(int), stream) != hipSuccess) return VIVIT_E_LAUNCH;
  const dim3 grid(8 * TP_WG);
  const int ni = (int)n;
  // n <= 1280: the 32 workgroups of one XCD (2 us per exchange; <6, 6, 32> would spill 84 registers); above, all 256 CUs with
  // agent-scope exchanges (6.7 us: scripts/probe/grid_barrier.hip mode 3), a workgroup holds n / 256 rows
  for (int attempt = 0; attempt < 2; ++attempt) {
    pw.attempt = attempt;
    switch (kc) {
      case 1: trd_persist_kernel<1, 1, 32><<<grid, TP_THREADS, 0, stream>>>(A, lda, ni, ws, pw); break;
      case 2: trd_persist_kernel<2, 2, 32><<<grid, TP_THREADS, 0, stream>>>(A, lda, ni, ws, pw); break;
      case 3: trd_persist_kernel<3, 3, 32><<<grid, TP_THREADS, 0, stream>>>(A, lda, ni, ws, pw); break;
      case 4: trd_persist_kernel<4, 4, 32><<<grid, TP_THREADS, 0, stream>>>(A, lda, ni, ws, pw); break;
      case 5: trd_persist_kernel<5, 5, 32><<<grid, TP_THREADS, 0, stream>>>(A, lda, ni, ws, pw); break;
      case 6: trd_persist_kernel<6, 1, 256><<<grid, TP_THREADS, 0, stream>>>(A, lda, ni, ws, pw); break;
      case 8: trd_persist_kernel<8, 1, 256><<<grid, TP_THREADS, 0, stream>>>(A, lda, ni, ws, pw); break;
      default: return VIVIT_E_UNSUPPORTED;
    }
}
  return launch_status();
}

}  // namespace vivit
