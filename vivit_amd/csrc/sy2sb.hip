// Full symmetric -> symmetric band (half bandwidth NB = 64): stage 1 of the two-stage
// tridiagonalisation.  All O(n^3) work is MFMA GEMM; the matrix is streamed O(n / NB) times instead
// of the O(n) times of the one-stage reduction (sytrd.hip), which is what lifts the eigensolver off
// the HBM roofline.
//
// Panel p (columns j0 = p*NB .. j0+NB-1): QR-factorise the block below the band,
//   B = A[j0+NB:, j0:j0+NB] = Q R,  Q = I - V T V^T  (Householder, compact WY),
// then update the trailing matrix two-sidedly,  A22 <- Q^T A22 Q = A22 - V W^T - W V^T  with
//   P = A22 V,  X = P T,  W = X - 1/2 V (T^T (V^T X)).
// Everything after the panel QR is GEMM-shaped and runs k-major (Vt = V^T rows, Pt, Xt, Wt) so that
// the final update is one rank-2*NB GEMM over the lower tiles with a mirrored store (A22 is kept
// fully symmetric in memory: the P = A22 V product is then a plain GEMM).
//
// The panel QR is column by column with two launches per column (reflector + partial dots, then
// rank-1 update + next column's norm), 128-row tiles staged through LDS.
//
// Outputs: band entries stay in A (A[i][j], 0 <= i-j <= NB), reflector c of panel p goes to the
// dead upper-triangle row j0+c: A[j0+c][j0+NB+c ..] (v[j0+NB+c] = 1), tau1[j0+c].
#include <cstdlib>

#include "common.h"
#include "device_utils.h"
#include "eig_internal.h"

namespace vivit {

constexpr int SNB = 64;     // half bandwidth = panel width (must equal sb2st.hip's NB)
constexpr int QT = 128;     // rows per workgroup tile in the panel QR
#ifndef VIVIT_SGRP
#define VIVIT_SGRP 8
#endif
constexpr int SGRP = VIVIT_SGRP;  // panels per delayed trailing-matrix update (measured at n = 40 960 with the fp32 MFMA
                            // kernels: 1 -> 2.00 s, 2 -> 1.52 s, 4 -> 1.52 s; with the rank-512 update of 4 panels on
                            // the 256-tile kernels (K >= 512): 1.43 s; round 3, with the 64-row MFMA kernel for the
                            // P^T corrections: 4 -> 1.127 s, 6 -> 1.122, 8 -> 1.094 (other box: 8 -> 1.113, 12 -> 1.117,
                            // 16 -> 1.134): the rank-1024 update spends a fifth instead of a third of a tile on C)

struct QrPart {
  float *u;     // [2][nwg][SNB]
  float *diag;  // [2][SNB]
};

struct Sy2sbWs {
  float *pan;      // [n][SNB]   compact copy of the current panel block
  float *stackA;   // [2*SGRP*SNB][n]  V1 | W1 | V2 | W2 ...  (k-major, ld = n)
  float *stackB;   // [2*SGRP*SNB][n]  W1 | V1 | W2 | V2 ...
  float *xt;       // [SNB][n]    scratch (X^T)
  float *G12;      // [SNB][2*SNB*(SGRP-1)]
  float *S, *T, *Y3, *S2;  // [SNB*SNB] each
  float *tau1;     // [n]
  float *betas;    // [SNB] diagonal of R of the current panel
  QrPart qp;       // partials of the fused panel QR
  void *qpw;       // exchange buffers of the persistent panel QR
  void *gws;       // split-K workspace
  size_t gws_bytes;
};

__global__ __launch_bounds__(256) void sb_panel_load_kernel(const float *__restrict__ A, int64_t lda, int64_t j0, int64_t mp,
                                                            float *__restrict__ pan) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= mp * SNB) return;
  const int64_t r = idx / SNB, c = idx - r * SNB;
  pan[idx] = A[(j0 + SNB + r) * lda + j0 + c];
}

// The same panel read from the block ROW j0 (the matrix is stored in full: A[j0 + c][j0 + SNB + r] = A[j0 + SNB + r][j0 + c]):
// 64 x 64 tiles transposed through LDS, reads and writes both contiguous.  (Row mode: the pending updates of a group are
// applied to the block row with the 64-row kernel, see sy2sb_launch.)
__global__ __launch_bounds__(256) void sb_panel_load_row_kernel(const float *__restrict__ A, int64_t lda, int64_t j0, int64_t mp,
                                                                float *__restrict__ pan) {
  __shared__ float t[SNB][SNB + 1];
  const int64_t r0 = (int64_t)blockIdx.x * SNB;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < SNB * SNB; idx += 256) {
    const int c = idx / SNB, rr = idx % SNB;
    t[c][rr] = r0 + rr < mp ? A[(j0 + c) * lda + j0 + SNB + r0 + rr] : 0.f;
  }
  __syncthreads();
  for (int idx = tid; idx < SNB * SNB; idx += 256) {
    const int rr = idx / SNB, c = idx % SNB;
    if (r0 + rr < mp) pan[(r0 + rr) * SNB + c] = t[c][rr];
  }
}

// Coef[m][k] = S[k][g + m]: the 64 x kp coefficient block of a block-row update from the k-major stack
__global__ __launch_bounds__(256) void coef_gather_kernel(const float *__restrict__ S, int64_t lds_, int64_t g, int kp,
                                                          float *__restrict__ Coef) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= SNB * kp) return;
  const int k = idx / SNB, m = idx % SNB;
  Coef[(int64_t)m * kp + k] = S[(int64_t)k * lds_ + g + m];
}

// ---- fused panel QR step: ONE launch per column (the dependent kernel boundary is what a column costs).
// Launch c (c = -1: prologue) does, per 128-row workgroup tile held in LDS:
//   1. sums the per-workgroup partials of column c left by launch c-1:  u[cc] = sum_{r>c} x_r pan[r][cc]
//      (x = column c; u[c] is ||x||^2);  with the diagonal row (handed over in `diag`)
//      that gives the reflector scalars and  z = tau v^T P = tau (pan[c][:] + scal u)  without a second pass
//   2. writes v for its rows (k-major stack copies, reflector row of A) and updates its rows of the panel
//   3. produces the partials (and, if it owns row c+1, the diagonal row) for column c+1 from the updated tile.
// Partials and diagonal row are double-buffered by column parity (a fast workgroup may already write the
// partials of column c+1 while a slow one still reads those of column c).

__global__ __launch_bounds__(256) void qr_step_kernel(float *__restrict__ pan, int64_t mp, int c, int ncol, int nwg, QrPart pt,
                                                      float *__restrict__ vdst1, float *__restrict__ vdst2, int64_t lds_, int64_t gi0,
                                                      float *__restrict__ A, int64_t lda, int64_t j0,
                                                      float *__restrict__ tau1, float *__restrict__ betas) {
  __shared__ float tile[QT][SNB + 1];
  __shared__ float vs[QT], xs2[QT];
  __shared__ float zs[SNB];
  __shared__ float zq[4][SNB];
  __shared__ __attribute__((aligned(16))) float zq16[16][SNB];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * QT;
  // Every global load of the launch is issued before anything waits (the panel was written by the previous launch on
  // other XCDs, so each load is a trip to the memory side: serialised, the 8 tile loads + the partials + din were
  // ~5 of the launch's 8.3 us): tile quads and din into registers first, the partials right behind them.
  constexpr int NTL = QT * (SNB / 4) / 256;
  float4 tl[NTL];
#pragma unroll
  for (int i = 0; i < NTL; ++i) {
    const int idx = tid + 256 * i;
    const int rl = idx / (SNB / 4), c4 = (idx - rl * (SNB / 4)) * 4;
    const int64_t r = r0 + rl;
    tl[i] = *reinterpret_cast<const float4 *>(pan + (r < mp ? r : 0) * SNB + c4);
  }
  // first batch of the partial vectors of column c (prologue launch c = -1: loaded and ignored, so that the tile is
  // written to LDS at ONE place behind all loads - with several call sites hipcc hoists the masking selects, and
  // with them the wait for the tile loads, in front of the partial loads).  Thread (group q of 16, column quad c4)
  // takes partials q, q + 16, ... as float4, 20 independent loads in flight per batch (with 4 scalar loads in
  // flight this reduction was 6.6 of the launch's 12.4 us: pure L2 latency)
  const int par = c & 1;
  const float *uin = pt.u + (int64_t)par * nwg * SNB;
  const float *din = pt.diag + par * SNB;
  const int cq4 = tid & 15, q16 = tid >> 4;
  float4 pv[20];
#pragma unroll
  for (int i = 0; i < 20; ++i) {
    const int w = q16 + 16 * i;
    pv[i] = *reinterpret_cast<const float4 *>(uin + (int64_t)(w < nwg ? w : 0) * SNB + 4 * cq4);
  }
  const float din_c = din[c >= 0 ? c : 0], din_t = din[tid & (SNB - 1)];
  __builtin_amdgcn_sched_barrier(0);  // every load in flight before the first one is consumed
#pragma unroll
  for (int i = 0; i < NTL; ++i) {
    const int idx = tid + 256 * i;
    const int rl = idx / (SNB / 4), c4 = (idx - rl * (SNB / 4)) * 4;
    const bool ok = r0 + rl < mp;
    tile[rl][c4 + 0] = ok ? tl[i].x : 0.f;
    tile[rl][c4 + 1] = ok ? tl[i].y : 0.f;
    tile[rl][c4 + 2] = ok ? tl[i].z : 0.f;
    tile[rl][c4 + 3] = ok ? tl[i].w : 0.f;
  }
  if (c >= 0) {
    {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 20; ++i)
        if (q16 + 16 * i < nwg) { acc.x += pv[i].x; acc.y += pv[i].y; acc.z += pv[i].z; acc.w += pv[i].w; }
      for (int w0 = q16 + 16 * 20; w0 < nwg; w0 += 16 * 20) {   // (more than 320 workgroups: n > 40 960)
        float4 v[20];
#pragma unroll
        for (int i = 0; i < 20; ++i) {
          const int w = w0 + 16 * i;
          v[i] = *reinterpret_cast<const float4 *>(uin + (int64_t)(w < nwg ? w : 0) * SNB + 4 * cq4);
        }
#pragma unroll
        for (int i = 0; i < 20; ++i)
          if (w0 + 16 * i < nwg) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
      }
      *reinterpret_cast<float4 *>(&zq16[q16][4 * cq4]) = acc;
    }
    __syncthreads();  // tile and zq16 complete
    // ||x||^2 is the partial sum of column c itself: u[c] = sum_{r>c} x_r pan[r][c] = sum x_r^2
    float ssq = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 16; ++g2) ssq += zq16[g2][c];
    const float alpha = din_c;
    float tau = 0.f, beta = alpha, scal = 0.f;
    if (ssq > 0.f) {
      beta = -copysignf(sqrt_nr(alpha * alpha + ssq), alpha);
      tau = (beta - alpha) * rcp_nr(beta);
      scal = rcp_nr(alpha - beta);
    }
    if (tid < SNB) {
      float u = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < 16; ++g2) u += zq16[g2][tid];
      zs[tid] = tid > c ? tau * (din_t + scal * u) : 0.f;
    }
    if (tid < QT) {
      const int64_t r = r0 + tid;
      float v = 0.f;
      if (r < mp) {
        if (r == c) v = 1.f;
        else if (r > c) v = tile[tid][c] * scal;
        // k-major copies for the GEMMs and the reflector row for the back-transformation
        vdst1[(int64_t)c * lds_ + gi0 + r] = v;
        vdst2[(int64_t)c * lds_ + gi0 + r] = v;
        if (r >= c) A[(j0 + c) * lda + gi0 + r] = v;
      }
      vs[tid] = v;
    }
    if (blockIdx.x == 0 && tid == 0) { tau1[j0 + c] = tau; betas[c] = beta; }
  } else {
    if (tid < QT) vs[tid] = 0.f;   // prologue launch: nothing to apply
    if (tid < SNB) zs[tid] = 0.f;
  }
  __syncthreads();  // vs, zs (and the tile) complete
  // ---- update  pan[r][cc] -= v_r z[cc]  (cc > c; rows above c have v = 0)  fused with the partials of column
  // cn = c + 1:  u[cc] = sum_{r > cn} x_r pan[r][cc]  with x = the UPDATED column cn (snapshot in xs2 first: the
  // loop below overwrites that column of the tile while other threads would still read it)
  const int cn = c + 1;
  const bool has_next = cn < ncol;
  if (tid < QT) {
    const int64_t r = r0 + tid;
    xs2[tid] = (has_next && r > cn && r < mp) ? tile[tid][cn] - vs[tid] * zs[cn] : 0.f;
  }
  __syncthreads();
  {
    const int cc = tid & 63, q = tid >> 6;
    const float zc = zs[cc];
    float acc = 0.f;
#pragma unroll 8
    for (int i = 0; i < QT / 4; ++i) {
      const int rl = q + 4 * i;
      const float v = vs[rl];
      float x = tile[rl][cc];
      if (cc > c && v != 0.f) {
        x -= v * zc;
        tile[rl][cc] = x;
        if (r0 + rl < mp) pan[(r0 + rl) * SNB + cc] = x;
      }
      acc += xs2[rl] * x;
    }
    zq[q][cc] = acc;
  }
  if (!has_next) return;
  __syncthreads();
  {
    const int par = cn & 1;
    if (tid < SNB) pt.u[((int64_t)par * nwg + blockIdx.x) * SNB + tid] = (zq[0][tid] + zq[1][tid]) + (zq[2][tid] + zq[3][tid]);
    if (cn >= r0 && cn < r0 + QT && tid < SNB) pt.diag[par * SNB + tid] = tile[cn - r0][tid];
  }
}

// ---- the same panel QR as ONE persistent launch on ONE XCD (the chain is 65 dependent launches of ~7.3 us per panel,
// 0.31 s of the band reduction at n = 40 960).  The 32 workgroups of XCD 0 hold the whole panel in registers (10.5 MB at
// mp = 40 960: workgroup w rows [w RW, (w + 1) RW), RW = 64 RI; thread (row slot rs = tid / 8, column group cg = tid % 8)
// keeps rows rs + 64 i, columns 8 cg .. 8 cg + 7) and run the same fused column step with ONE exchange per column through
// the shared L2: a workgroup's partial sums u_w[64] (and the diagonal row, always workgroup 0's) go out by plain stores, one
// L2 atomic per workgroup, wave 0 polls the counter and sums the 32 partials with sc1 loads (2 us per exchange:
// scripts/probe/grid_barrier.hip; see sytrd_persist.hip for the protocol and its fallback).  pan receives rows 0..63 only
// (R; all that is read afterwards).
constexpr int QP_WG = 32, QP_THREADS = 512;
struct QrPersistWs {
  float *ubuf;    // [2][QP_WG][SNB]
  float *dbuf;    // [2][SNB]
  int *counter;   // per attempt a (0 | 1) at 8 a: monotonic arrival counter, arrival-gate state; [32 ..]: XCC ids
  int *tmo;       // the sticky failure word (persist_timeout_word)
  int attempt;    // 0: first launch; 1: the retry behind it (runs only if attempt 0 aborted at its arrival gate)
  int fault;      // VIVIT_PERSIST_FAULT (tests)
};

__device__ __forceinline__ float sel8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, int j) {
  const float t0 = (j & 1) ? a1 : a0, t1 = (j & 1) ? a3 : a2, t2 = (j & 1) ? a5 : a4, t3 = (j & 1) ? a7 : a6;
  const float u0 = (j & 2) ? t1 : t0, u1 = (j & 2) ? t3 : t2;
  return (j & 4) ? u1 : u0;
}

template <int RI>
__global__ __launch_bounds__(QP_THREADS) void qr_persist_kernel(float *__restrict__ pan, int64_t mp, int ncol, QrPersistWs pw,
                                                                float *__restrict__ vdst1, float *__restrict__ vdst2, int64_t lds_,
                                                                int64_t gi0, float *__restrict__ A, int64_t lda, int64_t j0,
                                                                float *__restrict__ tau1, float *__restrict__ betas) {
  if ((blockIdx.x & 7) != 0) return;
  // the retry runs only when the first attempt aborted at its gate (device_utils.h:persist_arrive; sytrd_persist.hip)
  if (pw.attempt == 1 && __hip_atomic_load(pw.counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != PERSIST_ABORT) return;
  int *const cnt = pw.counter + 8 * pw.attempt;
  const int w = blockIdx.x >> 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = tid & 7, rs = tid >> 3;
  constexpr int RW = 64 * RI;
  const int64_t rbase = (int64_t)w * RW;
  __shared__ float s_u[SNB], s_d[SNB], s_dn[SNB];
  __shared__ float s_red[QP_THREADS / 64][SNB];
  __shared__ float s_v[RW];
  __shared__ int s_flag[2];   // 0: slow (not one XCD), 1: dead (1: stalled exchange, 2: aborted at the gate)

  float x[RI][8];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int64_t r = rbase + rs + 64 * i;
    const float *src = pan + (r < mp ? r : 0) * SNB + 8 * cg;
    const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
    const bool ok = r < mp;
    x[i][0] = ok ? a.x : 0.f; x[i][1] = ok ? a.y : 0.f; x[i][2] = ok ? a.z : 0.f; x[i][3] = ok ? a.w : 0.f;
    x[i][4] = ok ? b.x : 0.f; x[i][5] = ok ? b.y : 0.f; x[i][6] = ok ? b.z : 0.f; x[i][7] = ok ? b.w : 0.f;
  }
  if (tid == 0) {   // are the 32 workgroups on one XCD?  (first exchange, the safe way)
    int xcc;
    __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    __hip_atomic_store(pw.counter + 32 + w, xcc & 15, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool go = persist_arrive(cnt, cnt + 1, QP_WG, ((pw.fault >> pw.attempt) & 1) ? 0ull : PERSIST_TIMEOUT_TICKS);
    const int dead = go ? 0 : 2;
    int slow = 0;
    const int x0 = __hip_atomic_load(pw.counter + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = 1; i < QP_WG; ++i) slow |= __hip_atomic_load(pw.counter + 32 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != x0;
    s_flag[0] = slow;
    s_flag[1] = dead;
  }
  __syncthreads();
  const bool slow = s_flag[0] != 0;
  if (s_flag[1] == 2) {   // aborted at the gate: nothing has been written; the second abort fails the solve
    if (tid == 0 && w == 0 && pw.attempt == 1) {
      __hip_atomic_fetch_or(pw.tmo, PERSIST_TMO_PANEL_QR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      betas[0] = __builtin_nanf("");
    }
    return;
  }
  const int grp = lane & ~7;   // first lane of this row slot's eight threads

  // partial sums of column cn over this workgroup's rows (x = the UPDATED column cn), diagonal row cn, out to the exchange
  auto publish = [&](int cn) __attribute__((always_inline)) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int64_t r = rbase + rs + 64 * i;
      float xn = __shfl(sel8(x[i][0], x[i][1], x[i][2], x[i][3], x[i][4], x[i][5], x[i][6], x[i][7], cn & 7), grp | (cn >> 3), 64);
      xn = (r > cn && r < mp) ? xn : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(xn, x[i][j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[j] += __shfl_xor(acc[j], 8, 64);
      acc[j] += __shfl_xor(acc[j], 16, 64);
      acc[j] += __shfl_xor(acc[j], 32, 64);
    }
    if (lane < 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s_red[wave][8 * lane + j] = acc[j];
    }
    if (w == 0 && rs == cn) {   // the diagonal row of column cn (rows 0..63 live in workgroup 0, i = 0)
#pragma unroll
      for (int j = 0; j < 8; ++j) s_dn[8 * cg + j] = x[0][j];
    }
  };
  // wave 0: the workgroup's partials (and diagonal row) to the exchange buffers of parity par, then the arrival counter
  auto send = [&](int par) __attribute__((always_inline)) {
    if (wave == 0) {
      float u = 0.f;
#pragma unroll
      for (int q = 0; q < QP_THREADS / 64; ++q) u += s_red[q][lane];
      float *ub = pw.ubuf + ((size_t)par * QP_WG + w) * SNB + lane, *db = pw.dbuf + (size_t)par * SNB + lane;
      if (slow) {
        __hip_atomic_store(ub, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == 0) __hip_atomic_store(db, s_dn[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        *ub = u;
        if (w == 0) *db = s_dn[lane];
      }
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  publish(0);
  __syncthreads();
  send(0);
  for (int c = 0; c < ncol && !s_flag[1]; ++c) {
    const int par = c & 1;
    if (wave == 0) {   // exchange: all 32 workgroups have delivered column c's partials
      const int target = (c + 2) * QP_WG;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
        if (__builtin_amdgcn_s_memrealtime() - t0 > PERSIST_TIMEOUT_TICKS) { s_flag[1] = 1; break; }
      const float *ub = pw.ubuf + (size_t)par * QP_WG * SNB + lane;
      float pv[QP_WG];
#pragma unroll
      for (int q = 0; q < QP_WG; ++q) pv[q] = __hip_atomic_load(ub + q * SNB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float dn = __hip_atomic_load(pw.dbuf + (size_t)par * SNB + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float u = 0.f;
#pragma unroll
      for (int q = 0; q < QP_WG; ++q) u += pv[q];
      s_u[lane] = u;
      s_d[lane] = dn;
    }
    __syncthreads();   // A
    if (s_flag[1]) break;
    const float ssq = s_u[c], alpha = s_d[c];
    float tau = 0.f, beta = alpha, scal = 0.f;
    if (ssq > 0.f) {
      beta = -copysignf(sqrt_nr(alpha * alpha + ssq), alpha);
      tau = (beta - alpha) * rcp_nr(beta);
      scal = rcp_nr(alpha - beta);
    }
    float z[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cc = 8 * cg + j;
      z[j] = cc > c ? tau * (s_d[cc] + scal * s_u[cc]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int64_t r = rbase + rs + 64 * i;
      const float xc = __shfl(sel8(x[i][0], x[i][1], x[i][2], x[i][3], x[i][4], x[i][5], x[i][6], x[i][7], c & 7), grp | (c >> 3), 64);
      const float v = r == c ? 1.f : ((r > c && r < mp) ? xc * scal : 0.f);
      if (cg == 0) s_v[rs + 64 * i] = v;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[i][j] = fmaf(-v, z[j], x[i][j]);
    }
    if (w == 0 && tid == 0) { tau1[j0 + c] = tau; betas[c] = beta; }
    const bool has_next = c + 1 < ncol;
    if (has_next) publish(c + 1);
    __syncthreads();   // B
    if (has_next) send(par ^ 1);
    // v for this workgroup's rows: k-major copies for the GEMMs and the reflector row for the back-transformation
    for (int idx = tid; idx < RW; idx += QP_THREADS) {
      const int64_t r = rbase + idx;
      if (r < mp) {
        const float v = s_v[idx];
        vdst1[(int64_t)c * lds_ + gi0 + r] = v;
        vdst2[(int64_t)c * lds_ + gi0 + r] = v;
        if (r >= c) A[(j0 + c) * lda + gi0 + r] = v;
      }
    }
  }
  if (w == 0) {   // R: rows 0..63 of the factored panel
    const int64_t r = rs;
    if (r < mp) {
      float *dst = pan + r * SNB + 8 * cg;
      *reinterpret_cast<float4 *>(dst) = make_float4(x[0][0], x[0][1], x[0][2], x[0][3]);
      *reinterpret_cast<float4 *>(dst + 4) = make_float4(x[0][4], x[0][5], x[0][6], x[0][7]);
    }
    if (tid == 0 && s_flag[1]) {   // a stalled exchange: poison the band and raise the status of its own
      betas[0] = __builtin_nanf("");
      __hip_atomic_fetch_or(pw.tmo, PERSIST_TMO_PANEL_QR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static bool qr_persist_enabled() {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("VIVIT_QR_PERSIST");
    on = e ? atoi(e) : 1;
  }
  return on != 0 && persist_override() != 0;
}
constexpr size_t QR_PERSIST_WS_BYTES = sizeof(float) * (2 * QP_WG * SNB + 2 * SNB) + 64 * sizeof(int) + 256;

// launches the persistent panel QR if the panel fits (mp <= 40 960); false = use the launch chain
static bool qr_persist_launch(float *pan, int64_t mp, int ncol, void *wsp, float *v1, float *v2, int64_t ldn, int64_t gi0, float *A,
                              int64_t lda, int64_t j0, float *tau1, float *betas, hipStream_t stream) {
  if (!qr_persist_enabled() || mp > (int64_t)QP_WG * 64 * 20 || device_cu_count() < 256) return false;
  QrPersistWs pw;
  pw.ubuf = reinterpret_cast<float *>(align_up(reinterpret_cast<uintptr_t>(wsp), 256));
  pw.dbuf = pw.ubuf + 2 * QP_WG * SNB;
  pw.counter = reinterpret_cast<int *>(pw.dbuf + 2 * SNB);
  pw.tmo = persist_timeout_word(stream);
  pw.fault = persist_fault();
  if (!pw.tmo) return false;
  if (hipMemsetAsync(pw.counter, 0, 64 * sizeof(int), stream) != hipSuccess) return false;
  const int64_t ri = cdiv(mp, (int64_t)QP_WG * 64);
  const dim3 grid(8 * QP_WG);
#define QP_LAUNCH(RI) qr_persist_kernel<RI><<<grid, QP_THREADS, 0, stream>>>(pan, mp, ncol, pw, v1, v2, ldn, gi0, A, lda, j0, tau1, betas)
  // two attempts: the second returns at once unless the first aborted at its arrival gate (nothing written by then)
  for (int attempt = 0; attempt < 2; ++attempt) {
    pw.attempt = attempt;
    if (ri <= 1) QP_LAUNCH(1);
    else if (ri <= 2) QP_LAUNCH(2);
    else if (ri <= 3) QP_LAUNCH(3);
    else if (ri <= 5) QP_LAUNCH(5);
    else if (ri <= 8) QP_LAUNCH(8);
    else if (ri <= 12) QP_LAUNCH(12);
    else if (ri <= 16) QP_LAUNCH(16);
    else QP_LAUNCH(20);
  }
#undef QP_LAUNCH
  return true;
}

// ---- 64-row products against a k-major operand:  Out[64][N] = alpha Coef[64][K] B[K][N] + beta Out  (B rows = the k-major
// stacks of the band reduction, leading dimension ldb; K = 64 .. 384, a multiple of 32).  The tile kernels spend 36-69 us on
// these shapes at N = 40 960 (one 16-k pipeline step per 64 bytes of an operand row: 0.3 TB/s); this kernel gives a workgroup
// 256 columns, stages 32 k at a time (B chunk 32 x 256 as float4 rows, Coef chunk 64 x 32) and runs v_mfma_f32_32x32x2f32 from
// LDS: MFMA-bound at ~7 us for K = 128.
struct Sk64Args {
  const float *Coef;   // [64][ldcoef], Coef[m][k]
  const float *B;
  float *Out;
  int64_t ldcoef, ldb, ldo, N;
  int K;
  float alpha, beta;
};
constexpr int SK_LDC = 36, SK_COLS = 128, SK_LDB = SK_COLS + 4, SK_NT = SK_COLS / 64;   // 128 columns per workgroup: 320 workgroups at N = 40 960

__global__ __launch_bounds__(256) void skinny64_kernel(Sk64Args p) {
  __shared__ __attribute__((aligned(16))) float sC[64 * SK_LDC];
  __shared__ __attribute__((aligned(16))) float sB[32 * SK_LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int64_t c0 = (int64_t)blockIdx.x * SK_COLS;
  f32x16 acc[SK_NT];
#pragma unroll
  for (int t = 0; t < SK_NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  constexpr int NLB = 32 * SK_COLS / 4 / 256;   // float4 loads of the B chunk per thread
  float4 rc[2], rb[NLB];
  auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      const int m = idx >> 3, kq = (idx & 7) * 4;
      rc[i] = *reinterpret_cast<const float4 *>(p.Coef + (int64_t)m * p.ldcoef + k0 + kq);
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int idx = tid + 256 * i;
      const int kr = idx / (SK_COLS / 4), cq = (idx % (SK_COLS / 4)) * 4;
      const int64_t c = c0 + cq;
      rb[i] = *reinterpret_cast<const float4 *>(p.B + (int64_t)(k0 + kr) * p.ldb + (c < p.N ? c : 0));   // (N % 4 == 0)
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  gload(0);
  for (int k0 = 0; k0 < p.K; k0 += 32) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      *reinterpret_cast<float4 *>(sC + (idx >> 3) * SK_LDC + (idx & 7) * 4) = rc[i];
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int idx = tid + 256 * i;
      const int kr = idx / (SK_COLS / 4), cq = (idx % (SK_COLS / 4)) * 4;
      *reinterpret_cast<float4 *>(sB + kr * SK_LDB + cq) = c0 + cq < p.N ? rb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (k0 + 32 < p.K) gload(k0 + 32);
    const float *pa = sC + (32 * wm + r) * SK_LDC + 4 * h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // k = 8 j + 4 h + i: lanes 0-31 supply k, lanes 32-63 k + 4
      const float4 a4 = *reinterpret_cast<const float4 *>(pa + 8 * j);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
      float bv[4][SK_NT];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < SK_NT; ++t) bv[i][t] = sB[(8 * j + 4 * h + i) * SK_LDB + 64 * wn + 32 * t + r];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < SK_NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i][t], acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < SK_NT; ++t) {
    const int64_t c = c0 + 64 * wn + 32 * t + r;
    if (c < p.N) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = 32 * wm + (e & 3) + 8 * (e >> 2) + 4 * h;
        float *o = p.Out + (int64_t)m * p.ldo + c;
        float v = p.alpha * acc[t][e];
        if (p.beta != 0.f) v += p.beta * *o;
        *o = v;
      }
    }
  }
}

// applicable: 16-byte aligned rows everywhere, K a multiple of 32, N a multiple of 4
static bool skinny64_enabled() {
  static int on = -1;
  if (on < 0) { const char *e = getenv("VIVIT_SY2SB_FUSED"); on = e ? atoi(e) : 1; }
  return on != 0;
}
static bool skinny64_launch(const float *Coef, int64_t ldcoef, const float *B, int64_t ldb, float *Out, int64_t ldo, int64_t N, int64_t K,
                            float alpha, float beta, hipStream_t stream) {
  if (!skinny64_enabled() || K < 32 || K % 32 != 0 || K > 4096 || N % 4 != 0 || ldb % 4 != 0 || ldcoef % 4 != 0 ||
      (reinterpret_cast<uintptr_t>(Coef) & 15) || (reinterpret_cast<uintptr_t>(B) & 15))
    return false;
  Sk64Args a;
  a.Coef = Coef; a.B = B; a.Out = Out; a.ldcoef = ldcoef; a.ldb = ldb; a.ldo = ldo; a.N = N; a.K = (int)K; a.alpha = alpha; a.beta = beta;
  skinny64_kernel<<<(unsigned)cdiv(N, SK_COLS), 256, 0, stream>>>(a);
  return true;
}

// Coef[64][128] = [ -1/2 T^T G T | T^T ]  (G = P^T-side Gram block Pt Vt^T): with the k-major rows [Vt; Pt] of the stack,
// Wt = Coef [Vt; Pt] = T^T Pt - 1/2 (T^T G T) Vt  -- the panel's W in ONE pass, X = P T never materialised.  One workgroup.
__global__ __launch_bounds__(256) void w_coef_kernel(const float *__restrict__ T, const float *__restrict__ G, float *__restrict__ Coef) {
  // Round 6: both 64^3 products on the fp32 matrix pipe (v_mfma_f32_32x32x2f32: an exact fp32 fma chain in k order), one 32 x 32
  // quadrant of the result per wave -- the scalar LDS loops took 24.4 us per panel (every multiply-add behind an LDS round
  // trip), this form 5.7 us (scripts/probe/ktrace_sy2sb.sh).
  __shared__ float sT[SNB][SNB + 1], sG[SNB][SNB + 1], sU[SNB][SNB + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, qi = (wave >> 1) * 32, qj = (wave & 1) * 32;
  for (int idx = tid; idx < SNB * SNB; idx += 256) {
    sT[idx / SNB][idx % SNB] = T[idx];
    sG[idx / SNB][idx % SNB] = G[idx];
  }
  __syncthreads();
  // U = T^T G:  U[a][b] = sum_k T[k][a] G[k][b]      (A operand: lane (r, h) holds A[qi + r][2 s + h] = T[2 s + h][qi + r])
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll 8
  for (int s2 = 0; s2 < SNB / 2; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sT[2 * s2 + h][qi + r], sG[2 * s2 + h][qj + r], acc, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < 16; ++e) sU[qi + (e & 3) + 8 * (e >> 2) + 4 * h][qj + r] = acc[e];
  __syncthreads();
  // Y = U T:  Y[a][b] = sum_k U[a][k] T[k][b]
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll 8
  for (int s2 = 0; s2 < SNB / 2; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sU[qi + r][2 * s2 + h], sT[2 * s2 + h][qj + r], acc, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int a_ = qi + (e & 3) + 8 * (e >> 2) + 4 * h, b_ = qj + r;
    Coef[a_ * 2 * SNB + b_] = -0.5f * acc[e];
    Coef[a_ * 2 * SNB + SNB + b_] = sT[b_][a_];
  }
}

// R (upper triangular, rows 0..min(NB, mp)-1 of the factored panel) back into A; zeros below it inside
// the band rows.
__global__ __launch_bounds__(256) void sb_panel_store_kernel(float *__restrict__ A, int64_t lda, int64_t j0, int64_t mp,
                                                             const float *__restrict__ pan, const float *__restrict__ betas) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= SNB * SNB) return;
  const int r = idx / SNB, c = idx - r * SNB;
  if (r >= mp) return;
  A[(j0 + SNB + r) * lda + j0 + c] = (r < c) ? pan[(int64_t)r * SNB + c] : (r == c ? betas[c] : 0.f);
}

// T (nb x nb upper triangular, forward/columnwise larft) from S = V^T V and tau; nb = SNB.
// Two diagonal half blocks by column-parallel back substitution (the longest column is 496 multiply-adds instead of 2016), then
// the coupling block T12 = -T11 S12 T22 as two 32^3 products over 256 threads: 43 -> 34 us per panel (round 3); with the column of
// the back substitution in registers instead of LDS (round 6): see below.
__global__ __launch_bounds__(256) void larft_kernel(const float *__restrict__ S, int64_t lds_, const float *__restrict__ tau, int nb,
                                                    int nvalid, float *__restrict__ T) {
  constexpr int H = SNB / 2, LD = SNB + 1;
  __shared__ float Ss[SNB * LD], Ts[SNB * LD], Xs[H * (H + 1)], taus[SNB];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < SNB * SNB; idx += 256) {
    Ss[(idx / SNB) * LD + (idx % SNB)] = S[(int64_t)(idx / SNB) * lds_ + (idx % SNB)];
    Ts[(idx / SNB) * LD + (idx % SNB)] = 0.f;
  }
  if (tid < SNB) taus[tid] = (tid < nvalid && tid < nb) ? tau[tid] : 0.f;
  __syncthreads();
  if (tid < SNB) {
    // Column j of a diagonal half block by back substitution with the column IN REGISTERS (round 6; the same recurrence and
    // summation order as device_utils.h:tfactor_column, whose column lives in LDS: every multiply-add there waits for an LDS
    // round trip behind the store of the previous row -- 496 dependent round trips for the longest column, 25 of the kernel's
    // 34 us).  t[c] = 0 for c > j (T is upper triangular), so the sums need no bounds; S[i][c] is the same address for all
    // threads of a half block: a broadcast read.
    const int half = tid / H, j = tid % H;
    const float *Sb = Ss + half * (H * LD + H), *tb = taus + half * H;
    float t[H];
#pragma unroll
    for (int c = 0; c < H; ++c) t[c] = 0.f;
#pragma unroll
    for (int i = H - 1; i >= 0; --i) {
      float acc = 0.f;
#pragma unroll
      for (int c = i + 1; c < H; ++c) acc += Sb[i * LD + c] * t[c];
      t[i] = i == j ? tb[j] : (i < j ? -tb[i] * acc : 0.f);
    }
    float *Tb = Ts + half * (H * LD + H);
#pragma unroll
    for (int i = 0; i < H; ++i) Tb[i * LD + j] = t[i];
  }
  __syncthreads();
  // the coupling block T12 = -T11 (S12 T22): two 32^3 products on the fp32 matrix pipe by wave 0 (an exact fma chain in k order;
  // the structural zeros of the triangular factors add exact zeros) -- the scalar LDS loops were 4 of the kernel's 14 us.
  // Kernel trace at n = 20 480 (scripts/probe/ktrace_sy2sb.sh): larft_kernel 33.9 -> 14.2 (column in registers) -> 10.2 us per panel
  if (tid < 64) {
    const int r = tid & 31, h = tid >> 5;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll 8
    for (int s2 = 0; s2 < H / 2; ++s2)   // X = S12 T22
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ss[r * LD + H + 2 * s2 + h], Ts[(H + 2 * s2 + h) * LD + H + r], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) Xs[((e & 3) + 8 * (e >> 2) + 4 * h) * (H + 1) + r] = acc[e];
  }
  __syncthreads();
  if (tid < 64) {
    const int r = tid & 31, h = tid >> 5;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll 8
    for (int s2 = 0; s2 < H / 2; ++s2)   // T12 = -T11 X
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ts[r * LD + 2 * s2 + h], Xs[(2 * s2 + h) * (H + 1) + r], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) Ts[((e & 3) + 8 * (e >> 2) + 4 * h) * LD + H + r] = -acc[e];
  }
  __syncthreads();
  for (int idx = tid; idx < SNB * SNB; idx += 256) T[idx] = Ts[(idx / SNB) * LD + (idx % SNB)];
}

// AB[i][d] = A[i][i - 2*NB + d] for NB <= d <= 2*NB (0 <= i-j <= NB), zero bulge room for d < NB
// (ldab: row stride of AB, >= 2 NB + 1; entries beyond the band row -- the padding of the library's own layout -- are zeroed)
__global__ __launch_bounds__(256) void sb_extract_band_kernel(const float *__restrict__ A, int64_t lda, int64_t n,
                                                              float *__restrict__ AB, int64_t ldab) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * ldab) return;
  const int64_t i = idx / ldab;
  const int d = (int)(idx - i * ldab);
  const int64_t j = i - 2 * SNB + d;
  AB[idx] = (d >= SNB && d <= 2 * SNB && j >= 0) ? A[i * lda + j] : 0.f;
}

// split-K scratch: the largest need over the GEMM shapes used per panel (trailing size mp <= n)
static size_t sy2sb_gemm_ws_bytes(int64_t n) {
  size_t m = 0;
  for (int64_t mp = 1; mp <= n; mp = mp < 4096 ? mp * 2 : mp + 4096) {
    const int64_t q = mp < n ? mp : n;
    size_t a = gemm_workspace_bytes(SNB, SNB, q, false);
    size_t b = gemm_workspace_bytes(SNB, q, q, false);
    size_t c = gemm_workspace_bytes(SNB, q, SNB, false);
    size_t d = 0, e = 0, f = 0;
    for (int g = 1; g < SGRP; ++g) {
      const size_t d1 = gemm_workspace_bytes(SNB, 2 * SNB * g + SNB, q, false), e1 = gemm_workspace_bytes(SNB, q, 2 * SNB * g, false);
      const size_t f1 = gemm_workspace_bytes(q, SNB, 2 * SNB * g, false);
      d = d1 > d ? d1 : d; e = e1 > e ? e1 : e; f = f1 > f ? f1 : f;
    }
    m = a > m ? a : m; m = b > m ? b : m; m = c > m ? c : m;
    m = d > m ? d : m; m = e > m ? e : m; m = f > m ? f : m;
    const size_t u = gemm_workspace_bytes(q, q, (int64_t)2 * SNB * SGRP, true);  // trailing update (operand pieces of the bf16 pipe)
    m = u > m ? u : m;
  }
  size_t a = gemm_workspace_bytes(SNB, SNB, n, false), b = gemm_workspace_bytes(SNB, n, n, false);
  m = a > m ? a : m; m = b > m ? b : m;
  return m * 2;  // generous: shapes between the sampled mp values
}

size_t sy2sb_workspace_bytes(int64_t n) {
  const int64_t nwg = cdiv(n, QT) + 1;
  size_t b = 0;
  b += align_up(sizeof(float) * n * SNB, 256);          // pan
  b += align_up(sizeof(float) * 2 * SGRP * SNB * n, 256) * 2;  // stackA, stackB
  b += align_up(sizeof(float) * SNB * n, 256);          // xt
  b += align_up(sizeof(float) * SNB * 2 * SNB * SGRP, 256);    // G12
  b += align_up(sizeof(float) * 2 * nwg * SNB, 256);    // QR partials u (double-buffered)
  b += align_up(sizeof(float) * 2 * SNB, 256);          // QR diagonal row
  b += align_up(QR_PERSIST_WS_BYTES, 256);              // persistent panel QR
  b += align_up(sizeof(float) * SNB * SNB, 256) * 4;    // S T Y3 S2
  b += align_up(sizeof(float) * n, 256);                // tau1
  b += align_up(sizeof(float) * SNB, 256);              // betas
  b += align_up(sy2sb_gemm_ws_bytes(n), 256);
  return b + 1024;
}

// Reduce A (n x n, FULL symmetric storage, lda) to band form in place.  tau1_out receives the pointer
// to the reflector scalars (inside the workspace).
int sy2sb_launch(float *A, int64_t n, int64_t lda, void *wsbase, float **tau1_out, hipStream_t stream) {
  char *p = reinterpret_cast<char *>(align_up(reinterpret_cast<uintptr_t>(wsbase), 256));
  auto take = [&](size_t bytes) {
    char *r = p;
    p += align_up(bytes, 256);
    return r;
  };
  const int64_t nwg = cdiv(n, QT) + 1;
  Sy2sbWs ws;
  ws.pan = (float *)take(sizeof(float) * n * SNB);
  ws.stackA = (float *)take(sizeof(float) * 2 * SGRP * SNB * n);
  ws.stackB = (float *)take(sizeof(float) * 2 * SGRP * SNB * n);
  ws.xt = (float *)take(sizeof(float) * SNB * n);
  ws.G12 = (float *)take(sizeof(float) * SNB * 2 * SNB * SGRP);
  ws.qp.u = (float *)take(sizeof(float) * 2 * nwg * SNB);
  ws.qp.diag = (float *)take(sizeof(float) * 2 * SNB);
  ws.qpw = take(QR_PERSIST_WS_BYTES);
  ws.S = (float *)take(sizeof(float) * SNB * SNB);
  ws.T = (float *)take(sizeof(float) * SNB * SNB);
  ws.Y3 = (float *)take(sizeof(float) * SNB * SNB);
  ws.S2 = (float *)take(sizeof(float) * SNB * SNB);
  ws.tau1 = (float *)take(sizeof(float) * n);
  ws.betas = (float *)take(sizeof(float) * SNB);
  ws.gws_bytes = sy2sb_gemm_ws_bytes(n);
  ws.gws = take(ws.gws_bytes);
  *tau1_out = ws.tau1;
  if (hipMemsetAsync(ws.tau1, 0, sizeof(float) * n, stream) != hipSuccess) return VIVIT_E_LAUNCH;

  const int64_t ldn = n;
  // Householder QR of the panel at column j0 (rows gi0 = j0 + SNB ..): reflector t in row t of v1 and v2
  // Row mode (16-byte aligned rows, n % 4 == 0): the pending updates of a group go to the block ROW that holds the next panel
  // (64 x (n - j0): the 64-row MFMA kernel; the tall 64-column form on the tile kernels took 60-100 us per panel) and the panel
  // is read from there; the block column below the band then keeps stale values that nothing reads.
  const bool rowmode = skinny64_enabled() && (n % 4 == 0) && (lda % 4 == 0) && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
  auto factor_panel = [&](int64_t j0, float *v1, float *v2) -> int {
    const int64_t mp = n - j0 - SNB, gi0 = j0 + SNB;
    const int ncol = (int)(mp < SNB ? mp : SNB);
    const int g = (int)cdiv(mp, QT);
    if (rowmode) sb_panel_load_row_kernel<<<(unsigned)cdiv(mp, SNB), 256, 0, stream>>>(A, lda, j0, mp, ws.pan);
    else sb_panel_load_kernel<<<(unsigned)cdiv(mp * SNB, 256), 256, 0, stream>>>(A, lda, j0, mp, ws.pan);
    if (!qr_persist_launch(ws.pan, mp, ncol, ws.qpw, v1, v2, ldn, gi0, A, lda, j0, ws.tau1, ws.betas, stream))
      for (int c = -1; c < ncol; ++c)  // c = -1: partials of column 0
        qr_step_kernel<<<g, 256, 0, stream>>>(ws.pan, mp, c, ncol, g, ws.qp, v1, v2, ldn, gi0, A, lda, j0, ws.tau1, ws.betas);
    sb_panel_store_kernel<<<(unsigned)cdiv(SNB * SNB, 256), 256, 0, stream>>>(A, lda, j0, mp, ws.pan, ws.betas);
    return launch_status();
  };
  // W = X - 1/2 V (T^T V^T X) with X = A22 V T for the panel at j0; Vt, Wt: [SNB][mp] k-major at column offset gi0.
  // kp > 0: P^T = V^T A22 is corrected for the kp / 2 SNB pending updates of A22 that have not been applied to memory yet.
  auto compute_w = [&](int64_t j0, float *Vt, float *Wt, float *Wcopy, int64_t kp, const float *sAg, const float *sBg) -> int {
    const int64_t mp = n - j0 - SNB, gi0 = j0 + SNB;
    const int ncol = (int)(mp < SNB ? mp : SNB);
    float *A22 = A + gi0 * lda + gi0;
    int st;
    // Pt = Vt A22                      [SNB x mp]   (A22 symmetric: B operand k-major = A22 itself; the big
    // operand is streamed exactly once: gemm64_dma_kernel)
    prof_mark(PROF_STAGE_SY2SB, stream);
    st = gemm_launch(LAY_K, LAY_M, Vt, A22, Wt, SNB, mp, mp, ldn, lda, ldn, 1.f, 0.f, false, ws.gws, ws.gws_bytes, stream);
    prof_mark(PROF_STAGE_SY2SB_PP, stream);
    if (st != VIVIT_OK) return st;
    // [G12 | S] = Vt [V1 W1 .. V]^T: the stack rows 0 .. kp + SNB - 1 end with this panel's V, so the Gram blocks against the
    // pending updates (G12, for the correction of Pt) and against itself (S, for T) are ONE deep-K product
    const int64_t ldg = kp + SNB;
    st = gemm_launch(LAY_K, LAY_K, Vt, sAg, ws.G12, SNB, ldg, mp, ldn, ldn, ldg, 1.f, 0.f, false, ws.gws, ws.gws_bytes, stream);
    if (st != VIVIT_OK) return st;
    larft_kernel<<<1, 256, 0, stream>>>(ws.G12 + kp, ldg, ws.tau1 + j0, SNB, ncol, ws.T);
    if (kp > 0) {   // Pt -= G12 [W1^T; V1^T; ...]: P^T = V^T A22 was formed from the not yet updated trailing matrix
      if (!skinny64_launch(ws.G12, ldg, sBg, ldn, Wt, ldn, mp, kp, -1.f, 1.f, stream)) {
        st = gemm_launch(LAY_K, LAY_M, ws.G12, sBg, Wt, SNB, mp, kp, ldg, ldn, ldn, -1.f, 1.f, false, ws.gws, ws.gws_bytes, stream);
        if (st != VIVIT_OK) return st;
      }
    }
    // W in one pass over the stack rows [Vt; Pt] (they are adjacent: Vt = Wt - SNB rows):  Wt = T^T Pt - 1/2 (T^T G T) Vt,
    // G = Pt Vt^T.  The result goes to the copy in the other stack first (a product cannot overwrite its operand).
    st = gemm_launch(LAY_K, LAY_K, Wt, Vt, ws.S2, SNB, SNB, mp, ldn, ldn, SNB, 1.f, 0.f, false, ws.gws, ws.gws_bytes, stream);
    if (st != VIVIT_OK) return st;
    w_coef_kernel<<<1, 256, 0, stream>>>(ws.T, ws.S2, ws.G12);   // (G12 is free again: T and the correction are done)
    if (Wt == Vt + (int64_t)SNB * ldn &&
        skinny64_launch(ws.G12, 2 * SNB, Vt, ldn, Wcopy, ldn, mp, 2 * SNB, 1.f, 0.f, stream))
      return hipMemcpy2DAsync(Wt, sizeof(float) * ldn, Wcopy, sizeof(float) * ldn, sizeof(float) * mp, SNB, hipMemcpyDeviceToDevice,
                              stream) == hipSuccess ? VIVIT_OK : VIVIT_E_LAUNCH;
    // (unaligned leading dimension: the general products)  Xt = T^T Pt, Wt = Xt - 1/2 Y3 Vt with Y3 = T^T G T
    float *Xt = ws.xt + gi0;
    st = gemm_launch(LAY_M, LAY_M, ws.T, Wt, Xt, SNB, mp, SNB, SNB, ldn, ldn, 1.f, 0.f, false, ws.gws, ws.gws_bytes, stream);
    if (st != VIVIT_OK) return st;
    if (hipMemcpy2DAsync(Wt, sizeof(float) * ldn, Xt, sizeof(float) * ldn, sizeof(float) * mp, SNB, hipMemcpyDeviceToDevice,
                         stream) != hipSuccess)
      return VIVIT_E_LAUNCH;
    // (first half of the coefficient block, ld 2 SNB, is -1/2 Y3)
    st = gemm_launch(LAY_K, LAY_M, ws.G12, Vt, Wt, SNB, mp, SNB, 2 * SNB, ldn, ldn, 1.f, 1.f, false, ws.gws, ws.gws_bytes, stream);
    if (st != VIVIT_OK) return st;
    return hipMemcpy2DAsync(Wcopy, sizeof(float) * ldn, Wt, sizeof(float) * ldn, sizeof(float) * mp, SNB, hipMemcpyDeviceToDevice,
                            stream) == hipSuccess ? VIVIT_OK : VIVIT_E_LAUNCH;
  };

  // Panels are processed in GROUPS of SGRP with ONE rank-(2 SNB SGRP) update of the trailing matrix per group
  // instead of SGRP rank-128 updates - the update is a read-modify-write of the whole trailing matrix and
  // HBM-bound.  The pending updates of the group's earlier panels are applied at once only to the block column
  // that becomes the next panel; that panel's P^T = V^T A22 is formed from the not yet updated trailing matrix
  // and corrected,   V^T (A22 - sum_i V_i W_i^T + W_i V_i^T) = V^T A22 - (V^T [V_i W_i ...]) [W_i^T; V_i^T; ...].
  // stackA = [V1; W1; V2; W2; ...], stackB = [W1; V1; W2; V2; ...] (k-major): the update is stackA^T stackB.
  float *sA = ws.stackA, *sB = ws.stackB;
  for (int64_t j0 = 0; j0 + SNB < n;) {
    int st;
    if (hipMemsetAsync(sA, 0, sizeof(float) * 2 * SGRP * SNB * n, stream) != hipSuccess) return VIVIT_E_LAUNCH;
    if (hipMemsetAsync(sB, 0, sizeof(float) * 2 * SGRP * SNB * n, stream) != hipSuccess) return VIVIT_E_LAUNCH;
    int np = 0;            // panels of this group done so far
    int64_t gi_last = 0;   // first row/column of the trailing matrix after the group's last panel
    for (; np < SGRP && j0 + SNB < n; ++np, j0 += SNB) {
      const int64_t gi = j0 + SNB;
      const int64_t kp = (int64_t)2 * SNB * np;  // stack rows of the pending updates
      if (np > 0) {
        // pending updates on the block column that becomes this panel: rows gi - SNB .., columns gi - SNB .. gi - 1
        const int64_t gc = gi - SNB;
        bool done = false;
        if (rowmode) {   // A[gc : gc + 64, gc :] -= (sA[:, gc : gc + 64])^T sB[:, gc :]   (the update matrix is symmetric)
          coef_gather_kernel<<<(unsigned)cdiv(SNB * kp, 256), 256, 0, stream>>>(sA, ldn, gc, (int)kp, ws.G12);
          done = skinny64_launch(ws.G12, kp, sB + gc, ldn, A + gc * lda + gc, lda, n - gc, kp, -1.f, 1.f, stream);
        }
        if (!done) {
          if (rowmode) return VIVIT_E_UNSUPPORTED;   // (cannot happen: rowmode implies the kernel applies)
          st = gemm_launch(LAY_M, LAY_M, sA + gc, sB + gc, A + gc * lda + gc, n - gc, SNB, kp, ldn, ldn, lda, -1.f, 1.f, false,
                           ws.gws, ws.gws_bytes, stream);
          if (st != VIVIT_OK) return st;
        }
      }
      // V -> sA rows kp.., sB rows kp + SNB..;  W -> sA rows kp + SNB.., sB rows kp..
      float *Vrow = sA + kp * n, *Wrow = sA + (kp + SNB) * n;
      st = factor_panel(j0, Vrow, sB + (kp + SNB) * n);
      if (st != VIVIT_OK) return st;
      st = compute_w(j0, Vrow + gi, Wrow + gi, sB + kp * n + gi, kp, sA + gi, sB + gi);   // W -> sA rows kp + SNB.. and sB rows kp..
      if (st != VIVIT_OK) return st;
      gi_last = gi;
    }
    // ---- trailing matrix (from the last panel's gi) -= stackA^T stackB : one update, lower tiles + mirror
    const int64_t mt = n - gi_last;
    prof_mark(PROF_STAGE_SY2SB, stream);
    st = gemm_launch(LAY_M, LAY_M, sA + gi_last, sB + gi_last, A + gi_last * lda + gi_last, mt, mt, (int64_t)2 * SNB * np, ldn, ldn,
                     lda, -1.f, 1.f, true, ws.gws, ws.gws_bytes, stream);
    prof_mark(PROF_STAGE_SY2SB_UPD, stream);
    if (st != VIVIT_OK) return st;
  }
  return launch_status();
}

// ---- the panel factorisation on its own (the multi-GPU band reduction, vivit_amd/distributed.py: every rank factors
// the broadcast panel itself, the trailing matrix is sharded).  pan: [mp][SNB] row-major, factored in place (R above the
// diagonal of its first SNB rows, diagonal in betas); Vt: [SNB][ldv] k-major reflectors (row c = v_c, v_c[c] = 1, zeros
// before); tau: [SNB]; T: [SNB][SNB] upper triangular, Q = I - V T V^T.
size_t sy2sb_panel_qr_workspace_bytes(int64_t mp) {
  const int64_t nwg = cdiv(mp, QT) + 1;
  return align_up(sizeof(float) * 2 * nwg * SNB, 256) + align_up(sizeof(float) * 2 * SNB, 256) + align_up(sizeof(float) * SNB * SNB, 256) +
         align_up(QR_PERSIST_WS_BYTES, 256) + align_up(gemm_workspace_bytes(SNB, SNB, mp, false), 256) + 1024;
}

int sy2sb_panel_qr_launch(float *pan, int64_t mp, float *Vt, int64_t ldv, float *tau, float *betas, float *T, void *wsbase,
                          size_t ws_bytes, hipStream_t stream) {
  if (ws_bytes < sy2sb_panel_qr_workspace_bytes(mp)) return VIVIT_E_WORKSPACE;
  char *p = reinterpret_cast<char *>(align_up(reinterpret_cast<uintptr_t>(wsbase), 256));
  auto take = [&](size_t bytes) {
    char *r = p;
    p += align_up(bytes, 256);
    return r;
  };
  const int64_t nwg = cdiv(mp, QT) + 1;
  QrPart qp;
  qp.u = (float *)take(sizeof(float) * 2 * nwg * SNB);
  qp.diag = (float *)take(sizeof(float) * 2 * SNB);
  float *S = (float *)take(sizeof(float) * SNB * SNB);
  void *qpw = take(QR_PERSIST_WS_BYTES);
  const size_t gws_bytes = gemm_workspace_bytes(SNB, SNB, mp, false);
  void *gws = take(gws_bytes);
  const int ncol = (int)(mp < SNB ? mp : SNB);
  const int g = (int)cdiv(mp, QT);
  if (hipMemsetAsync(tau, 0, sizeof(float) * SNB, stream) != hipSuccess) return VIVIT_E_LAUNCH;
  if (hipMemsetAsync(betas, 0, sizeof(float) * SNB, stream) != hipSuccess) return VIVIT_E_LAUNCH;
  if (hipMemset2DAsync(Vt, sizeof(float) * ldv, 0, sizeof(float) * mp, SNB, stream) != hipSuccess) return VIVIT_E_LAUNCH;
  // (reflector row of "A" = the same row of Vt: written twice with the same values)
  if (!qr_persist_launch(pan, mp, ncol, qpw, Vt, Vt, ldv, 0, Vt, ldv, 0, tau, betas, stream))
    for (int c = -1; c < ncol; ++c)
      qr_step_kernel<<<g, 256, 0, stream>>>(pan, mp, c, ncol, g, qp, Vt, Vt, ldv, 0, Vt, ldv, 0, tau, betas);
  int st = gemm_launch(LAY_K, LAY_K, Vt, Vt, S, SNB, SNB, mp, ldv, ldv, SNB, 1.f, 0.f, false, gws, gws_bytes, stream);
  if (st != VIVIT_OK) return st;
  larft_kernel<<<1, 256, 0, stream>>>(S, SNB, tau, SNB, ncol, T);
  return launch_status();
}

int sy2sb_extract_band_launch(const float *A, int64_t lda, int64_t n, float *AB, int64_t ldab, hipStream_t stream) {
  sb_extract_band_kernel<<<(unsigned)cdiv(n * ldab, 256), 256, 0, stream>>>(A, lda, n, AB, ldab);
  return launch_status();
}

} // namespace vivit

using namespace vivit;

extern "C" {

size_t vivit_sy2sb_f32_workspace_bytes(int64_t n) { return n > 0 ? sy2sb_workspace_bytes(n) : 0; }

// Stage 1a of the two-stage path, exported for testing: A (FULL symmetric storage) -> band form in
// place, band copied to AB in the row-band layout of vivit_sb2st_f32; tau1: [n] reflector scalars.
int vivit_sy2sb_f32(float *A, int64_t n, int64_t lda, float *AB, float *tau1, void *workspace, size_t workspace_bytes,
                    void *stream) {
  if (n < 1 || !A || !AB || !tau1 || lda < n) return VIVIT_E_BADARG;
  if (!workspace || workspace_bytes < sy2sb_workspace_bytes(n)) return VIVIT_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float *t1;
  int st = sy2sb_launch(A, n, lda, workspace, &t1, s);
  if (st != VIVIT_OK) return st;
  st = sy2sb_extract_band_launch(A, lda, n, AB, 2 * SNB + 1, s);
  if (st != VIVIT_OK) return st;
  if (hipMemcpyAsync(tau1, t1, sizeof(float) * n, hipMemcpyDeviceToDevice, s) != hipSuccess) return VIVIT_E_LAUNCH;
  return VIVIT_OK;
}

size_t vivit_sy2sb_panel_qr_f32_workspace_bytes(int64_t mp) { return mp > 0 ? sy2sb_panel_qr_workspace_bytes(mp) : 0; }

// Householder QR of one sub-band panel (see sy2sb_panel_qr_launch): the replicated step of the multi-GPU band reduction.
int vivit_sy2sb_panel_qr_f32(float *pan, int64_t mp, float *Vt, int64_t ldv, float *tau, float *betas, float *T, void *workspace,
                             size_t workspace_bytes, void *stream) {
  if (mp < 1 || !pan || !Vt || !tau || !betas || !T || ldv < mp) return VIVIT_E_BADARG;
  if (!workspace) return VIVIT_E_WORKSPACE;
  return sy2sb_panel_qr_launch(pan, mp, Vt, ldv, tau, betas, T, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

} // extern "C"
