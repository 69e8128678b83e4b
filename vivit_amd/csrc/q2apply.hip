// Back-transformation through the bulge-chasing reflectors (stage 2 of the two-stage reduction):
//   Zt <- Zt * Q2^T,   Q2 = product of all H(s,k) of sb2st.hip in generation order,
// with Zt holding one eigenvector per ROW.
//
// Reflectors are grouped into blocks (g, k): the QW = 64 consecutive sweeps s in [g*QW, (g+1)*QW)
// at chase level k.  Their vectors are NB = 64 long and shifted by one position per sweep, so a
// block touches a 128-column window of Zt and is applied in compact-WY form
//   S <- S - ((S V^T) T^T) V          S: 128 rows x 128 window columns of Zt,  V: 64 x 128
// by one 256-thread workgroup per 128-row slab of Zt: three MFMA products with S, V, T and the
// intermediate W all resident in LDS (154 KB; one workgroup per CU).
// Blocks only conflict with their neighbours in (g, k); with G the group index counted from the
// last group, all blocks with equal tau = G + k are independent (validated in scripts/sb2st_proto.py),
// so the host issues one prepare (T factors) + one apply launch per wavefront step.
#include "common.h"
#include "device_utils.h"
#include "eig_internal.h"

namespace vivit {

constexpr int QB = 64;            // reflector length (= half bandwidth NB)
constexpr int QW = 64;            // sweeps per block
constexpr int QWIN = QB + QW;     // window width (128)
constexpr int LDS_S = QWIN + 4;   // 132: conflict-free ds_read_b128 fragments
constexpr int LDS_W = QW + 4;     // 68

struct Q2Step {
  const float *R2;
  int64_t ldr;
  const float *tau2;
  int nk, n, ngroups, tau, G_lo;
};

__device__ __forceinline__ void q2_block(const Q2Step &a, int blk, int &g0, int &k, int &c_start) {
  const int G = a.G_lo + blk;
  const int g = a.ngroups - 1 - G;
  g0 = g * QW;
  k = a.tau - G;
  c_start = g0 + 1 + k * QB;
}

// V window element: reflector t of the block at window column i
__device__ __forceinline__ float q2_v(const Q2Step &a, int g0, int c_start, int t, int i) {
  const int s = g0 + t;
  const int c0 = c_start + t;
  if (s > a.n - 3 || c0 >= a.n) return 0.f;
  const int L = (a.n - c0) < QB ? (a.n - c0) : QB;
  if (i < t || i >= t + L) return 0.f;
  return a.R2[(int64_t)s * a.ldr + c_start + i];
}

// ---- T factor of every block of the step ------------------------------------------------------
__global__ __launch_bounds__(256) void q2_prepare_kernel(Q2Step a, float *__restrict__ Tbuf) {
  __shared__ float V[QW][QWIN + 1];
  __shared__ float S[QW][QW + 1];
  __shared__ float Ts[QW][QW + 1];
  __shared__ float col[QW], taus[QW];
  const int tid = threadIdx.x;
  int g0, k, c_start;
  q2_block(a, blockIdx.x, g0, k, c_start);
  for (int idx = tid; idx < QW * QWIN; idx += 256) {
    const int t = idx / QWIN, i = idx - t * QWIN;
    V[t][i] = q2_v(a, g0, c_start, t, i);
  }
  if (tid < QW) {
    const int s = g0 + tid;
    taus[tid] = (s <= a.n - 3 && c_start + tid < a.n) ? a.tau2[(int64_t)s * a.nk + k] : 0.f;
  }
  for (int idx = tid; idx < QW * QW; idx += 256) Ts[idx / QW][idx % QW] = 0.f;
  __syncthreads();
  {  // S = V V^T (64 x 64), 16 entries per thread
    const int r = tid >> 2, cb = (tid & 3) * 16;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    for (int i = 0; i < QWIN; ++i) {
      const float vr = V[r][i];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] += vr * V[cb + c][i];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) S[r][cb + c] = acc[c];
  }
  __syncthreads();
  // forward columnwise larft: T[0:i, i] = -tau_i T[0:i, 0:i] S[0:i, i]
  for (int i = 0; i < QW; ++i) {
    const float ti = taus[i];
    if (tid < i) {
      float acc = 0.f;
      for (int c = tid; c < i; ++c) acc += Ts[tid][c] * S[c][i];
      col[tid] = -ti * acc;
    }
    __syncthreads();
    if (tid < i) Ts[tid][i] = col[tid];
    if (tid == i) Ts[i][i] = ti;
    __syncthreads();
  }
  float *T = Tbuf + (int64_t)blockIdx.x * QW * QW;
  for (int idx = tid; idx < QW * QW; idx += 256) T[idx] = Ts[idx / QW][idx % QW];
}

// ---- apply:  S <- S - ((S V^T) T^T) V  ---------------------------------------------------------
__global__ __launch_bounds__(256) void q2_apply_kernel(Q2Step a, const float *__restrict__ Tbuf, float *__restrict__ Zt,
                                                       int64_t ldz, int nrows) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *sS = lds;                        // [128][LDS_S]
  float *sV = sS + 128 * LDS_S;           // [64][LDS_S]
  float *sT = sV + QW * LDS_S;            // [64][LDS_W]
  float *sW = sT + QW * LDS_W;            // [128][LDS_W]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  int g0, k, c_start;
  q2_block(a, blockIdx.y, g0, k, c_start);
  const int64_t row0 = (int64_t)blockIdx.x * 128;

  for (int idx = tid; idx < 128 * QWIN; idx += 256) {
    const int rr = idx / QWIN, i = idx - rr * QWIN;
    const int64_t row = row0 + rr, colg = (int64_t)c_start + i;
    sS[rr * LDS_S + i] = (row < nrows && colg < a.n) ? Zt[row * ldz + colg] : 0.f;
  }
  for (int idx = tid; idx < QW * QWIN; idx += 256) {
    const int t = idx / QWIN, i = idx - t * QWIN;
    sV[t * LDS_S + i] = q2_v(a, g0, c_start, t, i);
  }
  const float *T = Tbuf + (int64_t)blockIdx.y * QW * QW;
  for (int idx = tid; idx < QW * QW; idx += 256) sT[(idx / QW) * LDS_W + (idx % QW)] = T[idx];
  __syncthreads();

  const int wrow = wave * 32;  // this wave's 32 rows of the slab
  // GEMM 1: W1[r][t] = sum_i S[r][i] V[t][i]          (K = 128)
  f32x16 acc1[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc1[j][e] = 0.f;
#pragma unroll 4
  for (int q = 0; q < QWIN / 8; ++q) {
    const float4 av = *reinterpret_cast<const float4 *>(sS + (wrow + r) * LDS_S + 8 * q + 4 * h);
    const float4 b0 = *reinterpret_cast<const float4 *>(sV + (r)*LDS_S + 8 * q + 4 * h);
    const float4 b1 = *reinterpret_cast<const float4 *>(sV + (32 + r) * LDS_S + 8 * q + 4 * h);
    const float aa[4] = {av.x, av.y, av.z, av.w}, bb0[4] = {b0.x, b0.y, b0.z, b0.w}, bb1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      acc1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[tt], bb0[tt], acc1[0], 0, 0, 0);
      acc1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[tt], bb1[tt], acc1[1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) sW[(wrow + (e & 3) + 8 * (e >> 2) + 4 * h) * LDS_W + 32 * j + r] = acc1[j][e];
  __syncthreads();
  // GEMM 2: W2[r][t'] = sum_t W1[r][t] T[t'][t]        (K = 64)
  f32x16 acc2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[j][e] = 0.f;
#pragma unroll 4
  for (int q = 0; q < QW / 8; ++q) {
    const float4 av = *reinterpret_cast<const float4 *>(sW + (wrow + r) * LDS_W + 8 * q + 4 * h);
    const float4 b0 = *reinterpret_cast<const float4 *>(sT + (r)*LDS_W + 8 * q + 4 * h);
    const float4 b1 = *reinterpret_cast<const float4 *>(sT + (32 + r) * LDS_W + 8 * q + 4 * h);
    const float aa[4] = {av.x, av.y, av.z, av.w}, bb0[4] = {b0.x, b0.y, b0.z, b0.w}, bb1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[tt], bb0[tt], acc2[0], 0, 0, 0);
      acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[tt], bb1[tt], acc2[1], 0, 0, 0);
    }
  }
  __syncthreads();  // every wave has finished reading W1
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) sW[(wrow + (e & 3) + 8 * (e >> 2) + 4 * h) * LDS_W + 32 * j + r] = acc2[j][e];
  __syncthreads();
  // GEMM 3: U[r][i] = sum_t' W2[r][t'] V[t'][i]        (K = 64, N = 128)
  f32x16 acc3[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc3[j][e] = 0.f;
#pragma unroll 2
  for (int q = 0; q < QW / 8; ++q) {
    const float4 av = *reinterpret_cast<const float4 *>(sW + (wrow + r) * LDS_W + 8 * q + 4 * h);
    const float aa[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const float *vrow = sV + (8 * q + 4 * h + tt) * LDS_S;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc3[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[tt], vrow[32 * j + r], acc3[j], 0, 0, 0);
    }
  }
  // S_new = S_old - U, straight to global memory (C/D layout: 128-B row segments)
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rr = wrow + (e & 3) + 8 * (e >> 2) + 4 * h, i = 32 * j + r;
      const int64_t row = row0 + rr, colg = (int64_t)c_start + i;
      if (row < nrows && colg < a.n) Zt[row * ldz + colg] = sS[rr * LDS_S + i] - acc3[j][e];
    }
}

constexpr int Q2_LDS_BYTES = (128 * LDS_S + QW * LDS_S + QW * LDS_W + 128 * LDS_W) * 4;

size_t q2_workspace_bytes(int64_t n) {
  const int64_t ngroups = cdiv(n - 2 > 0 ? n - 2 : 1, QW);
  return (size_t)(ngroups + 2) * QW * QW * sizeof(float) + 256;
}

// Zt[nrows x n] (ldz) <- Zt * Q2^T
int q2_apply_launch(float *Zt, int64_t ldz, int64_t nrows, int64_t n, const float *R2, int64_t ldr, const float *tau2,
                    void *ws, hipStream_t stream) {
  if (n < 3) return VIVIT_OK;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(q2_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            Q2_LDS_BYTES) != hipSuccess)
      return VIVIT_E_LAUNCH;
    attr = true;
  }
  float *Tbuf = reinterpret_cast<float *>(align_up(reinterpret_cast<uintptr_t>(ws), 256));
  const int nsweeps = (int)(n - 2);
  const int ngroups = (int)cdiv(nsweeps, QW);
  // kmax of group g: largest k with a reflector for its first sweep g0: g0 + 1 + k*QB <= n - 1
  auto kmax = [&](int g) { return (int)((n - 2 - (int64_t)g * QW) / QB); };
  Q2Step a;
  a.R2 = R2; a.ldr = ldr; a.tau2 = tau2; a.nk = sb2st_num_levels(n); a.n = (int)n; a.ngroups = ngroups;
  const int tau_max = (ngroups - 1) + kmax(0);
  int G_lo = 0;
  for (int tau = 0; tau <= tau_max; ++tau) {
    // valid G: 0 <= G <= min(tau, ngroups-1) and tau - G <= kmax(group of G); G + kmax(G) grows with G
    while (G_lo < ngroups && G_lo + kmax(ngroups - 1 - G_lo) < tau) ++G_lo;
    const int G_hi = tau < ngroups - 1 ? tau : ngroups - 1;
    if (G_lo > G_hi) continue;
    a.tau = tau; a.G_lo = G_lo;
    const unsigned nblk = (unsigned)(G_hi - G_lo + 1);
    q2_prepare_kernel<<<nblk, 256, 0, stream>>>(a, Tbuf);
    q2_apply_kernel<<<dim3((unsigned)cdiv(nrows, 128), nblk), 256, Q2_LDS_BYTES, stream>>>(a, Tbuf, Zt, ldz, (int)nrows);
  }
  return launch_status();
}

} // namespace vivit
