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
// and cannot deadlock).  Each task is one 256-thread workgroup (four waves, each a quarter of the columns).
//
// The reflectors are kept for the back-transformation: v(s,k) at R2[s][R_k], tau at tau2[s][k].
#include "common.h"
#include "device_utils.h"
#include "eig_internal.h"

namespace vivit {

#ifndef SB2ST_VARIANT
#define SB2ST_VARIANT 0
#endif
constexpr int NB = 64;             // half bandwidth
constexpr int LDAB = 2 * NB + 1;   // band row length

// value of lane `l` (compile-time constant) as a wave-uniform scalar
__device__ __forceinline__ float rl(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

constexpr int LDT = NB + 4;  // LDS row stride: 16-byte aligned rows, conflict-free row-per-lane ds_read_b128

// value of lane `l` (wave-uniform, run-time) as a wave-uniform scalar
__device__ __forceinline__ float rlu(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// One task per 256-thread workgroup.  The four waves load the two 64 x 64 blocks from the band (one
// coalesced 256-byte segment per band row: row r of the band holds [E row r | lower D row r]
// contiguously; all 32 loads of a thread are in flight together) into LDS.  Thread (wave q, lane r) then
// owns columns 16 q .. 16 q + 15 of row r of E and of D in registers.  Every matrix-vector product is a
// 16-term partial sum per thread, completed across the four waves through LDS (two barriers: one for
// g = E pv, one for E^T v and D v together); the small vector work (reflector, tau, w) is done redundantly
// and bit-identically by every wave, so no broadcast barrier is needed.  (History: ten barrier-separated
// phases on LDS-resident blocks: 12 us per wavefront step; one wave with the blocks in registers and
// v_readlane broadcasts, no barrier: 11.5 us of which 6.7 us were that single wave's 512 FMAs + 450
// readlanes and 4.7 us the serialised loads - scripts/probe/run_variant_sb2st.py.)
__global__ __launch_bounds__(256) void sb2st_task_kernel(float *__restrict__ AB, int n, int t, int s_lo,
                                                         float *__restrict__ R2, int64_t ldr, float *__restrict__ tau2,
                                                         int nk, int rmod) {
  __shared__ __attribute__((aligned(16))) float sE[NB * LDT];
  __shared__ __attribute__((aligned(16))) float sD[NB * LDT];
  __shared__ float red[3][4][NB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = s_lo + blockIdx.x;
  const int k = t - 2 * s;
  const int c0 = s + 1 + k * NB;
  const int L = (n - c0) < NB ? (n - c0) : NB;
  if (k < 0 || L <= 0 || s > n - 3) return;
  const int q0 = 16 * wave;

  // ---- load: wave w takes band rows r = w, w + 4, ...; every address is inside the band array whatever
  // r, lane and k are, so the loads are unconditional (all in flight at once) and masked afterwards
  float pv = 0.f, ptau = 0.f, x = 0.f;
  if (k > 0) {
    pv = R2[(int64_t)(s % rmod) * ldr + (c0 - NB + lane)];
    ptau = tau2[(int64_t)s * nk + (k - 1)];
  } else {
    x = lane < L ? AB[(int64_t)(c0 + lane) * LDAB + (2 * NB - 1 - lane)] : 0.f;  // column s of the band
  }
#if SB2ST_VARIANT != 2
  float ev[NB / 4], dv[NB / 4];
#pragma unroll
  for (int rr = 0; rr < NB / 4; ++rr) {
    const int r = 4 * rr + wave;
    const float *row = AB + (int64_t)(c0 + (r < L ? r : 0)) * LDAB;
    ev[rr] = row[NB - r + lane];                                   // E[r][lane]
    dv[rr] = row[lane <= r ? 2 * NB - r + lane : 2 * NB];          // D[r][lane], lane <= r
  }
  __builtin_amdgcn_sched_barrier(0);   // keep the consumers behind ALL loads (hipcc hoists the first one otherwise)
#endif
#pragma unroll
  for (int rr = 0; rr < NB / 4; ++rr) {
    const int r = 4 * rr + wave;
    const bool in = r < L;
#if SB2ST_VARIANT == 2   // timing attribution only (results invalid): no block loads
    const float e1 = in ? 0.001f * (r + lane) : 0.f, d1 = in ? 0.002f * (r - lane) : 0.f;
#else
    const float e1 = (in && k > 0) ? ev[rr] : 0.f, d1 = in ? dv[rr] : 0.f;
#endif
    sE[r * LDT + lane] = e1;
    if (lane <= r) {
      sD[r * LDT + lane] = d1;
      sD[lane * LDT + r] = d1;
    }
  }
  __syncthreads();

#if SB2ST_VARIANT != 1     // (variant 1: timing attribution only, results invalid: no compute)
  float er[16], d[16];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    const float4 a = *reinterpret_cast<const float4 *>(sE + lane * LDT + q0 + 4 * c4);
    er[4 * c4] = a.x; er[4 * c4 + 1] = a.y; er[4 * c4 + 2] = a.z; er[4 * c4 + 3] = a.w;
    const float4 b = *reinterpret_cast<const float4 *>(sD + lane * LDT + q0 + 4 * c4);
    d[4 * c4] = b.x; d[4 * c4 + 1] = b.y; d[4 * c4 + 2] = b.z; d[4 * c4 + 3] = b.w;
  }
  float g = 0.f;
  if (k > 0) {
    // (i) g = ptau E pv;  E <- E - g pv^T
    float pvq[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) pvq[j] = rlu(pv, q0 + j);
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) part += er[j] * pvq[j];
    red[0][wave][lane] = part;
    __syncthreads();
    g = ptau * ((red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]));
#pragma unroll
    for (int j = 0; j < 16; ++j) er[j] -= g * pvq[j];
    x = sE[lane * LDT] - g * rl(pv, 0);   // first column of the updated E
  }
  // (ii) Householder reflector from x (every wave, identically)
  const float ssq = wave_sum_dpp(lane >= 1 ? x * x : 0.f);
  const float alpha = rl(x, 0);
  float tau = 0.f, beta = alpha, scal = 0.f;
  if (ssq > 0.f) {
    beta = -copysignf(sqrt_nr(alpha * alpha + ssq), alpha);
    tau = (beta - alpha) * rcp_nr(beta);
    scal = rcp_nr(alpha - beta);
  }
  const float v = lane < L ? (lane == 0 ? 1.f : x * scal) : 0.f;
  float vq[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) vq[j] = rlu(v, q0 + j);
  if (k > 0) {
    // z = tau E'^T v with E' = E - g pv^T (E still unchanged in LDS): this wave's 16 rows, lane = column
    float zp = 0.f, gv = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      zp += vq[j] * sE[(q0 + j) * LDT + lane];
      gv += vq[j] * rlu(g, q0 + j);
    }
    red[1][wave][lane] = zp - pv * gv;
  }
  {
    // (iii) p = tau D v: this wave's 16 columns, lane = row
    float pp = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) pp += d[j] * vq[j];
    red[2][wave][lane] = pp;
  }
  __syncthreads();
  if (k > 0) {
    const float z = tau * ((red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]));
#pragma unroll
    for (int j = 0; j < 16; ++j) er[j] -= v * rlu(z, q0 + j);
    if (wave == 0 && lane < L) er[0] = (lane == 0) ? beta : 0.f;  // exact zeros below the new sub-band entry
  } else if (wave == 0 && lane < L) {
    AB[(int64_t)(c0 + lane) * LDAB + (2 * NB - 1 - lane)] = (lane == 0) ? beta : 0.f;
  }
  // w = p - tau/2 (p.v) v;  D -= v w^T + w v^T
  const float p = tau * ((red[2][0][lane] + red[2][1][lane]) + (red[2][2][lane] + red[2][3][lane]));
  const float pdotv = wave_sum_dpp(p * v);
  const float w = p - 0.5f * tau * pdotv * v;
#pragma unroll
  for (int j = 0; j < 16; ++j) d[j] -= v * rlu(w, q0 + j) + w * vq[j];
  // results back to LDS (rows), reflector to global memory
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    if (k > 0)
      *reinterpret_cast<float4 *>(sE + lane * LDT + q0 + 4 * c4) = make_float4(er[4 * c4], er[4 * c4 + 1], er[4 * c4 + 2], er[4 * c4 + 3]);
    *reinterpret_cast<float4 *>(sD + lane * LDT + q0 + 4 * c4) = make_float4(d[4 * c4], d[4 * c4 + 1], d[4 * c4 + 2], d[4 * c4 + 3]);
  }
  if (wave == 0) {
    if (lane < L) R2[(int64_t)(s % rmod) * ldr + c0 + lane] = v;
    if (lane == 0) tau2[(int64_t)s * nk + k] = tau;
  }
#endif
  __syncthreads();

  // ---- store: lower part of D and E, same segments as loaded
#pragma unroll
  for (int rr = 0; rr < NB / 4; ++rr) {
    const int r = 4 * rr + wave;
#if SB2ST_VARIANT == 2
    if (r < L && sE[r * LDT + lane] == 12345.678f) {
#else
    if (r < L) {
#endif
      float *row = AB + (int64_t)(c0 + r) * LDAB;
      if (k > 0) row[NB - r + lane] = sE[r * LDT + lane];
      if (lane <= r) row[2 * NB - r + lane] = sD[r * LDT + lane];
    }
  }
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
