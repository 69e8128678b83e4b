// Device helpers shared by the eigensolver kernels (gfx950).
#pragma once
#include "common.h"

namespace vivit {

constexpr int SMALL_N_MAX = 192;
constexpr float EPS32 = 5.9604645e-8f;  // 2^-24

// Total order on floats for rank sorting: -inf < ... < -0 < +0 < ... < +inf < NaN.  With NaNs
// mapped to the largest key the rank of n values is always a permutation of 0..n-1, so a
// non-finite input can never turn into an out-of-range row index downstream.
__device__ __forceinline__ unsigned sort_key(float f) {
  const unsigned b = __float_as_uint(f);
  if ((b & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// 1 / x and sqrt(x) from the hardware approximations (1 ulp) plus one Newton step: the IEEE division / square-root
// expansions are ~100 dependent cycles each on the critical path of every task.
__device__ __forceinline__ float rcp_nr(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(r, fmaf(-x, r, 1.f), r);
}
__device__ __forceinline__ float sqrt_nr(float x) {
  if (x < 1e-30f) return sqrtf(x);  // (v_sqrt_f32 does not take denormals; the library call rescales)
  const float y = __builtin_amdgcn_sqrtf(x);
  return y > 0.f ? fmaf(fmaf(-y, y, x), 0.5f * __builtin_amdgcn_rcpf(y), y) : y;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- arrival gate of the persistent kernels (sytrd_persist.hip, sy2sb.hip:qr_persist_kernel, sb2st.hip) --------------
// Their workgroups exchange data through memory, so all `nwg` of them must be resident at once.  Whether they are is
// decided ONCE, atomically, before anything is written: `count` collects arrivals, `state` goes 0 (open) -> 1 (go: the
// last arriver) or -> 2 (abort: the first workgroup that waited `limit` ticks of s_memrealtime, 100 MHz) by compare-and-
// swap, so every workgroup -- also one that is only dispatched after the others gave up -- reads the same verdict.  An
// aborted attempt has touched nothing; the launcher runs the kernel a second time behind it (attempt 1, which returns at
// once unless attempt 0 aborted), and only when that aborts too the solve fails with VIVIT_INFO_PERSIST_TIMEOUT.
constexpr unsigned long long PERSIST_TIMEOUT_TICKS = 200000000ull;   // 2 s
enum { PERSIST_OPEN = 0, PERSIST_GO = 1, PERSIST_ABORT = 2 };
// bits of the sticky device word persist_timeout_word(): which persistent kernel gave up
enum { PERSIST_TMO_SYTRD = 1, PERSIST_TMO_PANEL_QR = 2, PERSIST_TMO_SB2ST = 4 };

__device__ __forceinline__ bool persist_arrive(int *count, int *state, int nwg, unsigned long long limit) {
  const int mine = __hip_atomic_fetch_add(count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  int expect = PERSIST_OPEN;
  if (mine == nwg && limit != 0)   // (limit 0 = the injected fault of VIVIT_PERSIST_FAULT: the gate never opens, whoever is fastest)
    __hip_atomic_compare_exchange_strong(state, &expect, PERSIST_GO, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  int st;
  while ((st = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == PERSIST_OPEN) {
    if (__builtin_amdgcn_s_memrealtime() - t0 >= limit) {
      expect = PERSIST_OPEN;
      __hip_atomic_compare_exchange_strong(state, &expect, PERSIST_ABORT, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  return st == PERSIST_GO;
}

// Sum over the 64 lanes, identical in every lane: four DPP adds inside each row of 16 lanes (xor 1, xor 2, mirror in 8,
// mirror in 16) and one readlane per row - about a tenth of the latency of the ds_bpermute butterfly.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
  auto lane_of = [](float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); };
  return (lane_of(v, 0) + lane_of(v, 16)) + (lane_of(v, 32) + lane_of(v, 48));
}

// Sum over the 256 threads, fixed order; red needs 4 floats. Every thread must call.
__device__ __forceinline__ float block_sum(float v, float *red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float block_max(float v, float *red, int tid) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// Implicit-shift QL on a private (d, e) copy; all lanes of the calling wave run it in lockstep.
// z: this lane's row of Z (nullptr = no vectors).  Returns the number of unconverged values.
//
// Deflation test: |e[m]| <= eps * (|d[m]| + |d[m+1]|)  OR  |e[m]| <= eps/2 * ||T||.  The second
// (EISPACK tql2-style, norm-relative) clause is what makes rank-deficient Gram matrices converge:
// their null-space block is rounding noise of size eps * ||G|| on which a purely relative test
// can stagnate in fp32, while the Householder stage has already committed a backward error of
// that size, so nothing is lost.
// `norm_floor`: a lower bound for the norm the absolute deflation tolerance is taken from.  A LEAF of a divide &
// conquer tree must pass the norm of the WHOLE tridiagonal matrix: a leaf that lies in the numerically-zero part of a
// rank-deficient spectrum has entries of 1e-20 ||T|| and below, its own norm is meaningless as a yardstick (and the
// squares of such entries underflow in the rotations: 60 sweeps without progress, "did not converge").
__device__ inline int ql_implicit(float *d, float *e, int n, float *z, float norm_floor = 0.f) {
  int nfail = 0;
  float tn = norm_floor;
  for (int i = 0; i < n; ++i) tn = fmaxf(tn, fabsf(d[i]) + (i + 1 < n ? fabsf(e[i]) : 0.f));
  const float abs_tol = 0.5f * EPS32 * tn;
  for (int l = 0; l < n; ++l) {
    int iter = 0;
    while (true) {
      int m = l;
      for (; m < n - 1; ++m) {
        const float dd = fabsf(d[m]) + fabsf(d[m + 1]);
        const float ae = fabsf(e[m]);
        if (ae <= EPS32 * dd || ae <= abs_tol) break;
      }
      if (m == l) break;
      if (iter++ >= 60) { ++nfail; break; }
      float g = (d[l + 1] - d[l]) / (2.f * e[l]);
      float r = sqrtf(g * g + 1.f);
      g = d[m] - d[l] + e[l] / (g + copysignf(r, g));
      float s = 1.f, c = 1.f, p = 0.f;
      int i;
      for (i = m - 1; i >= l; --i) {
        const float f = s * e[i], b = c * e[i];
        r = sqrtf(f * f + g * g);
        e[i + 1] = r;
        if (r == 0.f) {
          d[i + 1] -= p;
          e[m] = 0.f;
          break;
        }
        s = f / r;
        c = g / r;
        g = d[i + 1] - p;
        r = (d[i] - g) * s + 2.f * c * b;
        p = s * r;
        d[i + 1] = g + p;
        g = c * r - b;
        if (z) {
          const float zf = z[i + 1], zi = z[i];
          z[i + 1] = s * zi + c * zf;
          z[i] = c * zi - s * zf;
        }
      }
      if (r == 0.f && i >= l) continue;
      d[l] -= p;
      e[l] = g;
      e[m] = 0.f;
    }
  }
  return nfail;
}



// Compact-WY T factor, column-parallel.  The forward (columnwise) larft recurrence
//   T[0:i, i] = -tau_i T[0:i, 0:i] S[0:i, i],  T[i][i] = tau_i          (S = V^T V)
// is the statement T (D^-1 + striu(S)) = I, D = diag(tau); T is therefore also the LEFT inverse and
// column j can be computed on its own by back substitution
//   T[j][j] = tau_j,   T[i][j] = -tau_i sum_{c=i+1..j} S[i][c] T[c][j]   (i = j-1 .. 0)
// (tau_i = 0 gives a zero row and column, as in the recurrence).  One thread per column: O(nb^2/2)
// multiply-adds without a single barrier instead of nb barrier-separated steps.
// S, T: LDS arrays with row stride ld (thread j walks column j: conflict-free for ld odd).
__device__ __forceinline__ void tfactor_column(const float *S, const float *taus, float *T, int ld, int nb, int j) {
  if (j >= nb) return;
  for (int i = j + 1; i < nb; ++i) T[i * ld + j] = 0.f;
  T[j * ld + j] = taus[j];
  for (int i = j - 1; i >= 0; --i) {
    float acc = 0.f;
    for (int c = i + 1; c <= j; ++c) acc += S[i * ld + c] * T[c * ld + j];
    T[i * ld + j] = -taus[i] * acc;
  }
}

} // namespace vivit
