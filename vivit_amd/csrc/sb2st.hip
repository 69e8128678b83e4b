// Band -> tridiagonal reduction by bulge chasing (stage 2 of the two-stage tridiagonalisation).
//
// The symmetric band matrix (half bandwidth NB = 64) lives in a row-band layout with room for the
// bulge:  AB[i][j - i + 2*NB] = A[i][j]  for  i - 2*NB <= j <= i   (n x (2*NB+1) floats, 21 MB at
// n = 40 960: L2 / Infinity-Cache resident).  Sweep s annihilates column s below the sub-diagonal
// with a Householder reflector on rows s+1 .. s+NB and chases the resulting bulge down the band:
// task (s, k) works on rows  R_k = [s+1+k*NB, s+1+(k+1)*NB):
//     (i)   k > 0: apply H(s,k-1) from the right to the block E = A[R_k, R_{k-1}]   (creates the bulge)
//     (ii)  new reflector H(s,k) from the first column of E (k = 0: from column s), E <- H E
//     (iii) D = A[R_k, R_k] <- H D H
// Task (s, k) only depends on (s, k-1) and (s-1, k+1), so all tasks with equal t = 2 s + k are
// independent: the host launches one kernel per wavefront step t (about 2 n launches of up to
// n / (2 NB - 1) workgroups; a dependent launch boundary costs less than an in-kernel grid barrier
// and cannot deadlock).  Each task is one 256-thread workgroup with E and D resident in LDS.
//
// The reflectors are kept for the back-transformation: v(s,k) at R2[s][R_k], tau at tau2[s][k].
#include "common.h"
#include "device_utils.h"
#include "eig_internal.h"

namespace vivit {

constexpr int NB = 64;             // half bandwidth
constexpr int LDAB = 2 * NB + 1;   // band row length

__device__ __forceinline__ float quad_sum(float v) {  // sum over the 4 lanes of a quad
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}

__global__ __launch_bounds__(256) void sb2st_task_kernel(float *__restrict__ AB, int n, int t, int s_lo,
                                                         float *__restrict__ R2, int64_t ldr, float *__restrict__ tau2,
                                                         int nk, int rmod) {
  __shared__ float E[NB][NB + 1];
  __shared__ float D[NB][NB + 1];
  __shared__ float v[NB], pv[NB], z[NB], pw[NB];
  __shared__ float sc[4];  // 0: tau  1: ptau
  const int tid = threadIdx.x;
  const int s = s_lo + blockIdx.x;
  const int k = t - 2 * s;
  const int c0 = s + 1 + k * NB;
  const int L = (n - c0) < NB ? (n - c0) : NB;
  if (k < 0 || L <= 0 || s > n - 3) return;
  const int r4 = tid >> 2, q4 = tid & 3;  // 4 threads per row, 16 columns each

  // ---- load D (lower part mirrored) and, for k > 0, E and the previous reflector
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx / NB, c = idx - r * NB;
    float dv = 0.f, ev = 0.f;
    if (r < L && c < L) {
      const int rr = r >= c ? r : c, cc = r >= c ? c : r;
      dv = AB[(int64_t)(c0 + rr) * LDAB + (cc - rr + 2 * NB)];
    }
    if (k > 0 && r < L) ev = AB[(int64_t)(c0 + r) * LDAB + (NB + c - r)];
    D[r][c] = dv;
    E[r][c] = ev;
  }
  if (tid < NB) {
    pv[tid] = (k > 0) ? R2[(int64_t)(s % rmod) * ldr + (c0 - NB + tid)] : 0.f;
    v[tid] = 0.f;
  }
  if (tid == 0) sc[1] = (k > 0) ? tau2[(int64_t)s * nk + (k - 1)] : 0.f;
  __syncthreads();

  // ---- (i) E <- E (I - ptau pv pv^T)
  if (k > 0) {
    const float ptau = sc[1];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) dot += E[r4][q4 * 16 + c] * pv[q4 * 16 + c];
    dot = quad_sum(dot) * ptau;
#pragma unroll
    for (int c = 0; c < 16; ++c) E[r4][q4 * 16 + c] -= dot * pv[q4 * 16 + c];
  }
  __syncthreads();

  // ---- (ii) Householder reflector from x = first column of E (k = 0: column s of the band)
  if (tid < 64) {  // one wavefront
    float x = 0.f;
    if (tid < L) x = (k > 0) ? E[tid][0] : AB[(int64_t)(c0 + tid) * LDAB + (s - c0 - tid + 2 * NB)];
    const float ssq = wave_sum(tid >= 1 ? x * x : 0.f);
    const float alpha = __shfl(x, 0, 64);
    float tau = 0.f, beta = alpha, scal = 0.f;
    if (ssq > 0.f) {
      beta = -copysignf(sqrtf(alpha * alpha + ssq), alpha);
      tau = (beta - alpha) / beta;
      scal = 1.f / (alpha - beta);
    }
    if (tid < L) v[tid] = (tid == 0) ? 1.f : x * scal;
    if (tid == 0) sc[0] = tau;
    if (k == 0 && tid < L) AB[(int64_t)(c0 + tid) * LDAB + (s - c0 - tid + 2 * NB)] = (tid == 0) ? beta : 0.f;
    if (k > 0 && tid == 0) sc[2] = beta;
  }
  __syncthreads();
  const float tau = sc[0];
  if (k > 0) {
    // z[c] = sum_r v[r] E[r][c]  (4 threads per column, 16 rows each), then E -= tau v z^T
    const int c = tid >> 2;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += v[q4 * 16 + r] * E[q4 * 16 + r][c];
    acc = quad_sum(acc) * tau;
#pragma unroll
    for (int r = 0; r < 16; ++r) E[q4 * 16 + r][c] -= v[q4 * 16 + r] * acc;
    __syncthreads();
    if (tid < L) E[tid][0] = (tid == 0) ? sc[2] : 0.f;  // exact zeros below the new sub-band entry
  }
  // ---- (iii) D <- H D H:  p = tau D v;  w = p - tau/2 (p.v) v;  D -= v w^T + w v^T
  {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) acc += D[r4][q4 * 16 + c] * v[q4 * 16 + c];
    acc = quad_sum(acc) * tau;
    if (q4 == 0) z[r4] = acc;  // p
  }
  __syncthreads();
  if (tid < 64) {
    const float p = z[tid], vv = v[tid];
    const float pdotv = wave_sum(p * vv);
    pw[tid] = p - 0.5f * tau * pdotv * vv;
  }
  __syncthreads();
  {
    const float vr = v[r4], wr = pw[r4];
#pragma unroll
    for (int c = 0; c < 16; ++c) D[r4][q4 * 16 + c] -= vr * pw[q4 * 16 + c] + wr * v[q4 * 16 + c];
  }
  __syncthreads();

  // ---- write back the lower part of D, E, and the reflector
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx / NB, c = idx - r * NB;
    if (r < L && c <= r) AB[(int64_t)(c0 + r) * LDAB + (c - r + 2 * NB)] = D[r][c];
    if (k > 0 && r < L) AB[(int64_t)(c0 + r) * LDAB + (NB + c - r)] = E[r][c];
  }
  if (tid < L) R2[(int64_t)(s % rmod) * ldr + c0 + tid] = v[tid];
  if (tid == 0) tau2[(int64_t)s * nk + k] = tau;
}

__global__ __launch_bounds__(256) void sb2st_extract_kernel(const float *__restrict__ AB, int n, float *__restrict__ d,
                                                            float *__restrict__ e) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  d[i] = AB[(int64_t)i * LDAB + 2 * NB];
  if (i + 1 < n) e[i] = AB[(int64_t)(i + 1) * LDAB + 2 * NB - 1];
  else e[i] = 0.f;
}

int sb2st_num_levels(int64_t n) { return (int)cdiv(n, NB) + 1; }
int64_t sb2st_ring_rows(int64_t n) { const int64_t need = n / NB + 64; return need < n ? need : n; }

// Reduce the band AB (destroyed) to tridiagonal (d, e).  R2: [r2rows][ldr] reflector storage, sweep s
// uses row s % r2rows (r2rows = n keeps every reflector for the back-transformation; a ring of
// SB2ST_RING rows is enough for the chase itself when only eigenvalues are wanted);
// tau2: [n][sb2st_num_levels(n)].
int sb2st_launch(float *AB, int64_t n, float *d, float *e, float *R2, int64_t ldr, int64_t r2rows, float *tau2,
                 hipStream_t stream) {
  const int ni = (int)n;
  const int nk = sb2st_num_levels(n);
  if (n >= 3) {
    const int64_t tmax = 2 * (n - 3) + nk;
    for (int64_t t = 0; t <= tmax; ++t) {
      int64_t s_hi = t / 2;
      if (s_hi > n - 3) s_hi = n - 3;
      int64_t s_lo = 0;
      const int64_t num = t * NB + 1 - n;
      if (num >= 0) s_lo = num / (2 * NB - 1) + 1;
      if (s_lo > s_hi) continue;
      sb2st_task_kernel<<<(unsigned)(s_hi - s_lo + 1), 256, 0, stream>>>(AB, ni, (int)t, (int)s_lo, R2, ldr, tau2, nk, (int)r2rows);
    }
  }
  sb2st_extract_kernel<<<(unsigned)cdiv(n, 256), 256, 0, stream>>>(AB, ni, d, e);
  return launch_status();
}

} // namespace vivit

using namespace vivit;

extern "C" {

int vivit_sb2st_half_bandwidth(void) { return NB; }

size_t vivit_sb2st_f32_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return sizeof(float) * (size_t)n * (size_t)sb2st_num_levels(n) + 256;
}

// AB: [n][2*NB+1] row-band layout (see top of file), destroyed.  d: [n], e: [n-1 (n allocated)].
// R2: [n][n] reflector rows (required).  workspace: tau2.
int vivit_sb2st_f32(float *AB, int64_t n, float *d, float *e, float *R2, void *workspace, size_t workspace_bytes,
                    void *stream) {
  if (n < 1 || !AB || !d || !e || !R2) return VIVIT_E_BADARG;
  if (!workspace || workspace_bytes < vivit_sb2st_f32_workspace_bytes(n)) return VIVIT_E_WORKSPACE;
  float *tau2 = reinterpret_cast<float *>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  return sb2st_launch(AB, n, d, e, R2, n, n, tau2, static_cast<hipStream_t>(stream));
}

} // extern "C"
