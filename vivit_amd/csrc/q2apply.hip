// Back-transformation through the bulge-chasing reflectors (stage 2 of the two-stage reduction):
//   Zt <- Zt * Q2^T,   Q2 = product of all H(s,k) of sb2st.hip in generation order,
// with Zt holding one eigenvector per ROW.
//
// Reflectors are grouped into blocks (g, k): the QW = 64 consecutive sweeps s in [g*QW, (g+1)*QW)
// at chase level k.  Their vectors are NB = 64 long and shifted by one position per sweep, so a
// block touches a 128-column window of Zt and is applied in compact-WY form
//   S <- S - ((S V^T) T^T) V          S: rows x 128 window columns of Zt,  V: 64 x 128
// with the slab S held in registers through two chained MFMA products (see q2_apply_kernel).
// Blocks only conflict with their neighbours in (g, k); with G the group index counted from the
// last group, all blocks with equal tau = G + k are independent (validated in scripts/sb2st_proto.py);
// two consecutive levels are paired into a super-block, and the host issues one prepare (T V per block)
// + one apply launch per wavefront step of super-blocks.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "device_utils.h"
#include "eig_internal.h"

namespace vivit {

constexpr int QB = 64;            // reflector length (= half bandwidth NB)
constexpr int QW = 64;            // sweeps per block
constexpr int QWIN = QB + QW;     // window width (128)

struct Q2Step {
  const float *R2;
  int64_t ldr;
  const float *tau2;
  int nk, n, ngroups, tau, G_lo;
};

// super-block blk of super-step a.tau: group (first sweep g0) and level pair K (levels 2K, 2K+1)
__device__ __forceinline__ void q2_sblock(const Q2Step &a, int blk, int &g0, int &K) {
  const int G = a.G_lo + blk;
  const int g = a.ngroups - 1 - G;
  g0 = g * QW;
  K = a.tau - G;
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

// ---- T V (T = compact-WY factor) of every block of the step ------------------------------------
constexpr int Q2_WIN = 16;  // wavefront steps prepared per launch
struct Q2Win {
  int tau[Q2_WIN], G_lo[Q2_WIN], nblk[Q2_WIN];
  int64_t off[Q2_WIN];  // first block of the step in TVbuf
};

// blockIdx.y = step inside the window, blockIdx.x = 2 * block + level (blocks beyond the step's count exit)
__global__ __launch_bounds__(256) void q2_prepare_kernel(Q2Step a, Q2Win win, float *__restrict__ TVbuf) {
  if ((int)blockIdx.x >= 2 * win.nblk[blockIdx.y]) return;
  a.tau = win.tau[blockIdx.y];
  a.G_lo = win.G_lo[blockIdx.y];
  TVbuf += win.off[blockIdx.y] * QW * QWIN;
  __shared__ float V[QW][QWIN + 1];
  __shared__ float S[QW][QW + 1];
  __shared__ float Ts[QW][QW + 1];
  __shared__ float taus[QW];
  const int tid = threadIdx.x;
  int g0, K;
  q2_sblock(a, blockIdx.x >> 1, g0, K);
  const int k = 2 * K + (blockIdx.x & 1);
  const int c_start = g0 + 1 + k * QB;
  for (int idx = tid; idx < QW * QWIN; idx += 256) {
    const int t = idx / QWIN, i = idx - t * QWIN;
    V[t][i] = q2_v(a, g0, c_start, t, i);
  }
  if (tid < QW) {
    const int s = g0 + tid;
    taus[tid] = (s <= a.n - 3 && c_start + tid < a.n) ? a.tau2[(int64_t)s * a.nk + k] : 0.f;
  }
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
  // T factor, one thread per column (device_utils.h:tfactor_column)
  if (tid < QW) tfactor_column(&S[0][0], taus, &Ts[0][0], QW + 1, QW, tid);
  __syncthreads();
  // TV = T V (T upper triangular): the apply kernel multiplies the slab by it directly,
  //   (S V^T) T^T = S (T V)^T,  one MFMA product instead of two
  float *TV = TVbuf + (int64_t)blockIdx.x * QW * QWIN;
  for (int idx = tid; idx < QW * QWIN; idx += 256) {
    const int tp = idx / QWIN, i = idx - tp * QWIN;
    float acc = 0.f;
    for (int t = tp; t < QW; ++t) acc += Ts[tp][t] * V[t][i];
    TV[idx] = acc;
  }
}

// ---- apply:  S <- S - ((S V^T) T^T) V  for the two blocks (g, 2K), (g, 2K+1) of a super-block ----
// Window: 192 columns from wstart = g0 + 128 K (a multiple of 64: 16-byte aligned row segments);
// block A = level 2K works on window columns [0, 128), block B = level 2K+1 on [64, 192); inside a
// block, window column w holds V[t][w - 1], non-zero only for w in [t + 1, t + 64].
// Everything is computed TRANSPOSED so that the slab never leaves the registers: with the 32x32x2
// MFMA the accumulator of  X^T = A * B  (lane (r, h) holds X[row r][4h + (e&3) + 8(e>>2)]) is, with
// the k index permuted accordingly, exactly the B operand of the next product.  A wave owns 32 rows
// of Zt (lane (r, h) keeps S[row r][8q + 4h .. +3], q = 0..23, as loaded by float4) and runs
//     W2^T = (T V) S^T,   U^T = V^T W2^T,   S -= U
// per block with V and T V (prepared per block by q2_prepare_kernel: (S V^T) T^T = S (T V)^T saves the
// product with T) read from LDS as A operands.  Structurally zero 32x32 tiles of V (parallelogram) and
// T V (trapezoid) are skipped: 208 MFMAs per block and wave (320 for the dense three-product form).
// Pairing two levels reads/writes 192 instead of 2 x 128 columns of Zt per row (the kernel is close
// to HBM-bound) and halves the number of launches.  A 512-thread workgroup (8 waves, 2 per SIMD)
// keeps both blocks' V and T V in LDS (135 KB) and walks over several 256-row slabs.
constexpr int LDS_V = QWIN + 4;    // 132: conflict-free ds_read_b128 fragments
constexpr int Q2_THREADS = 512;
constexpr int Q2_SLAB = 32 * (Q2_THREADS / 64);  // 256 rows per workgroup iteration
constexpr int Q2_NQ = 24;                        // float4 per lane: 192 window columns

template <bool VEC>
__global__ __launch_bounds__(Q2_THREADS) void q2_apply_kernel(Q2Step a, const float *__restrict__ Tbuf,
                                                              float *__restrict__ Zt, int64_t ldz, int nrows) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *sV = lds;                      // [2][64][LDS_V]   V[t][w]
  float *sTV = sV + 2 * QW * LDS_V;     // [2][64][LDS_V]   (T V)[t'][w]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  int g0, K;
  q2_sblock(a, blockIdx.y, g0, K);
  const int wstart = g0 + 2 * K * QB;
  const bool haveB = g0 + 1 + (2 * K + 1) * QB < a.n;  // level 2K+1 exists for this group
  const int nslab = (nrows + Q2_SLAB - 1) / Q2_SLAB;

  {
    // V and T V of both blocks into LDS: 32 + 32 loads per thread, ALL in flight before the first LDS write (clamped
    // addresses, masked afterwards).  As two plain loops (load, wait, ds_write per element: 64 serialised L2 round
    // trips, ~50 us per workgroup against ~60 us of MFMA work on its slabs) this prologue was most of the gap
    // between the kernel and its MFMA bound.
    constexpr int NV = 2 * QW * QWIN / Q2_THREADS;  // 32
    const float *TV = Tbuf + (int64_t)blockIdx.y * 2 * QW * QWIN;  // [2][64][128], window index i = w - 1
    float vv[NV], tv[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + Q2_THREADS * u;
      const int b = idx / (QW * QWIN), rem = idx - b * (QW * QWIN);
      const int t = rem / QWIN, w = rem - t * QWIN;
      // q2_v(a, g0, c_start, t, w - 1) with the address clamped instead of the load skipped
      const int c_start = g0 + 1 + (2 * K + b) * QB;
      const int sw = g0 + t, c0 = c_start + t;
      const bool live = sw <= a.n - 3 && c0 < a.n;
      const int L = (a.n - c0) < QB ? (a.n - c0) : QB;
      const int i = w - 1;
      const bool in = live && i >= t && i < t + L;
      vv[u] = a.R2[in ? (int64_t)sw * a.ldr + c_start + i : 0];
      tv[u] = TV[(idx - rem) + t * QWIN + (w >= 1 ? w - 1 : 0)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + Q2_THREADS * u;
      const int b = idx / (QW * QWIN), rem = idx - b * (QW * QWIN);
      const int t = rem / QWIN, w = rem - t * QWIN;
      const int c_start = g0 + 1 + (2 * K + b) * QB;
      const int sw = g0 + t, c0 = c_start + t;
      const bool live = sw <= a.n - 3 && c0 < a.n;
      const int L = (a.n - c0) < QB ? (a.n - c0) : QB;
      const int i = w - 1;
      const bool in = live && i >= t && i < t + L;
      sV[(b * QW + t) * LDS_V + w] = in ? vv[u] : 0.f;
      sTV[(b * QW + t) * LDS_V + w] = (w >= 1) ? tv[u] : 0.f;
    }
  }
  __syncthreads();

  const int64_t colg = (int64_t)wstart + 4 * h;  // + 8 q
  // FAST: window completely inside the matrix and float4-aligned: unguarded loads/stores at
  // immediate offsets from one base pointer (rows past the end read row 0 and are not stored)
  const bool fast = VEC && wstart + 8 * Q2_NQ <= a.n;
  float4 s[Q2_NQ];
  for (int slab = blockIdx.x; slab < nslab; slab += gridDim.x) {
    __asm__ volatile("" ::: "memory");  // keep the (slab-invariant) V/T fragment reads inside the loop
    const int64_t row = (int64_t)slab * Q2_SLAB + wave * 32 + r;
    const bool rok = row < nrows;
    float *base = Zt + (rok ? row * ldz : 0);
    if (fast) {
      const float4 *b4 = reinterpret_cast<const float4 *>(base + colg);
#pragma unroll
      for (int q = 0; q < Q2_NQ; ++q) s[q] = b4[2 * q];
    } else {
#pragma unroll
      for (int q = 0; q < Q2_NQ; ++q) {
        const int64_t c = colg + 8 * q;
        if constexpr (VEC) {
          const bool ok = rok && c < a.n;  // n % 4 == 0 and c % 4 == 0: all four in or out
          const float4 x = *reinterpret_cast<const float4 *>(ok ? base + c : Zt);
          s[q] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
          float e[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool ok = rok && c + u < a.n;
            const float x = *(ok ? base + c + u : Zt);
            e[u] = ok ? x : 0.f;
          }
          s[q] = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
    }

    auto apply = [&](auto q0tag, const float *__restrict__ bV, const float *__restrict__ bTV) {
      constexpr int Q0 = decltype(q0tag)::value;
      // LDS fragment reads are software-pipelined one MFMA group (4 x 64 cycles) ahead by hand.
      // ---- W2^T = (T V) S^T.  (T V)[t'][w] is non-zero for w in [t' + 1, 127]: t'-tile 0 needs all four
      // w-tiles (q = 0 .. 15), t'-tile 1 the last three (q = 4 .. 15)
      f32x16 acc2[2];
      const float *pV = bTV + r * LDS_V + 4 * h;
      float4 av = *reinterpret_cast<const float4 *>(pV);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // prologue fragment
#pragma unroll
      for (int jo = 0; jo < 2; ++jo) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[jo][e] = 0.f;
#pragma unroll
        for (int q = 4 * jo; q < 16; ++q) {
          // next fragment: (jo, q + 1), or the first one of tile jo = 1
          const int nj = (q + 1 < 16) ? jo : jo + 1, nq = (q + 1 < 16) ? q + 1 : 4;
          float4 an = av;
          if (nj < 2) an = *reinterpret_cast<const float4 *>(pV + 32 * nj * LDS_V + 8 * nq);
          acc2[jo] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, s[Q0 + q].x, acc2[jo], 0, 0, 0);
          acc2[jo] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, s[Q0 + q].y, acc2[jo], 0, 0, 0);
          acc2[jo] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, s[Q0 + q].z, acc2[jo], 0, 0, 0);
          acc2[jo] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, s[Q0 + q].w, acc2[jo], 0, 0, 0);
          av = an;
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // next group's LDS fragment first,
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // then this group's 4 MFMAs
        }
      }
      // ---- U^T = V^T W2^T, one w-tile at a time; w-tile ji needs t'-tiles max(0, ji-2) .. min(1, ji).
      // A operand V[t'][w = 32 ji + r]: four ds_read_b32 per MFMA group from the t-major copy.
      // group order: (ji, jo, e4) with jo in the valid range: 6 x 4 = 24 groups
      const float *pU = bV + 4 * h * LDS_V + r;
      auto ldu = [&](int ji, int jo, int e4) {
        const float *q = pU + (32 * jo + 8 * e4) * LDS_V + 32 * ji;
        return make_float4(q[0], q[LDS_V], q[2 * LDS_V], q[3 * LDS_V]);
      };
      av = ldu(0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int ji = 0; ji < 4; ++ji) {
        f32x16 u;
#pragma unroll
        for (int e = 0; e < 16; ++e) u[e] = 0.f;
        const int jlo = (ji == 3 ? 1 : 0), jhi = (ji == 0 ? 1 : 2);
#pragma unroll
        for (int jo = jlo; jo < jhi; ++jo)
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            // next group
            int nji = ji, njo = jo, ne4 = e4 + 1;
            if (ne4 == 4) { ne4 = 0; njo = jo + 1; if (njo == jhi) { nji = ji + 1; njo = (nji == 3 ? 1 : 0); } }
            float4 an = av;
            if (nji < 4) an = ldu(nji, njo, ne4);
            u = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, acc2[jo][4 * e4 + 0], u, 0, 0, 0);
            u = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, acc2[jo][4 * e4 + 1], u, 0, 0, 0);
            u = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, acc2[jo][4 * e4 + 2], u, 0, 0, 0);
            u = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, acc2[jo][4 * e4 + 3], u, 0, 0, 0);
            av = an;
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // next group's LDS fragment first,
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // then this group's 4 MFMAs
          }
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          float4 &x = s[Q0 + 4 * ji + e4];
          x.x -= u[4 * e4 + 0]; x.y -= u[4 * e4 + 1]; x.z -= u[4 * e4 + 2]; x.w -= u[4 * e4 + 3];
        }
      }
    };
    apply(std::integral_constant<int, 0>{}, sV, sTV);
    if (haveB) apply(std::integral_constant<int, 8>{}, sV + QW * LDS_V, sTV + QW * LDS_V);

    // ---- store the slab back (same addresses as loaded)
    if (rok) {
      if (fast) {
        float4 *b4 = reinterpret_cast<float4 *>(base + colg);
#pragma unroll
        for (int q = 0; q < Q2_NQ; ++q) b4[2 * q] = s[q];
      } else {
#pragma unroll
        for (int q = 0; q < Q2_NQ; ++q) {
          const int64_t c = colg + 8 * q;
          if constexpr (VEC) {
            if (c < a.n) *reinterpret_cast<float4 *>(base + c) = s[q];
          } else {
            const float e[4] = {s[q].x, s[q].y, s[q].z, s[q].w};
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (c + u < a.n) base[c + u] = e[u];
          }
        }
      }
    }
  }
}

// ---- the same super-block step with 16-row waves (v_mfma_f32_16x16x4_f32) ------------------------------------------
// A 1024-thread workgroup = 16 waves x 16 rows (same 256-row slabs, same LDS images), 4 waves per SIMD instead of 2:
// lane (n = lane & 15, kq = lane >> 4) keeps S[row n][16 q + 4 kq .. + 3], q = 0..11 (48 registers instead of 96), and
// every global load instruction covers 64 contiguous bytes per row.  MFMA step (q, i) pairs k slot kq with window column
// 16 q + 4 kq + i for both operands, so A fragments stay float4 reads and - as with the 32 x 32 x 2 form - the accumulator
// of W2^T = (T V) S^T (lane (n, g) holds t' = 4 g + e) is the B operand of U^T = V^T W2^T as it stands, whose accumulator
// (w = 16 wt + 4 g + e) is element e of s[wt].  16 x 16 tiles follow the parallelogram / trapezoid more closely: 184
// MFMAs of 32 cycles per block and 16 rows = 11.5 % fewer matrix-pipe cycles than 208 x 64 per 32 rows.
constexpr int Q2W_THREADS = 1024;
constexpr int Q2W_NQ = 12;  // float4 per lane: 192 window columns

__global__ __launch_bounds__(Q2W_THREADS) void q2_apply16_kernel(Q2Step a, const float *__restrict__ Tbuf,
                                                                float *__restrict__ Zt, int64_t ldz, int nrows) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *sV = lds;                      // [2][64][LDS_V]   V[t][w]
  float *sTV = sV + 2 * QW * LDS_V;     // [2][64][LDS_V]   (T V)[t'][w]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kq = lane >> 4;
  int g0, K;
  q2_sblock(a, blockIdx.y, g0, K);
  const int wstart = g0 + 2 * K * QB;
  const bool haveB = g0 + 1 + (2 * K + 1) * QB < a.n;  // level 2K+1 exists for this group
  const int nslab = (nrows + Q2_SLAB - 1) / Q2_SLAB;
  {
    constexpr int NV = 2 * QW * QWIN / Q2W_THREADS;  // 16
    const float *TV = Tbuf + (int64_t)blockIdx.y * 2 * QW * QWIN;  // [2][64][128], window index i = w - 1
    float vv[NV], tv[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + Q2W_THREADS * u;
      const int b = idx / (QW * QWIN), rem = idx - b * (QW * QWIN);
      const int t = rem / QWIN, w = rem - t * QWIN;
      const int c_start = g0 + 1 + (2 * K + b) * QB;
      const int sw = g0 + t, c0 = c_start + t;
      const bool live = sw <= a.n - 3 && c0 < a.n;
      const int L = (a.n - c0) < QB ? (a.n - c0) : QB;
      const int i = w - 1;
      const bool in = live && i >= t && i < t + L;
      vv[u] = a.R2[in ? (int64_t)sw * a.ldr + c_start + i : 0];
      tv[u] = TV[(idx - rem) + t * QWIN + (w >= 1 ? w - 1 : 0)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + Q2W_THREADS * u;
      const int b = idx / (QW * QWIN), rem = idx - b * (QW * QWIN);
      const int t = rem / QWIN, w = rem - t * QWIN;
      const int c_start = g0 + 1 + (2 * K + b) * QB;
      const int sw = g0 + t, c0 = c_start + t;
      const bool live = sw <= a.n - 3 && c0 < a.n;
      const int L = (a.n - c0) < QB ? (a.n - c0) : QB;
      const int i = w - 1;
      const bool in = live && i >= t && i < t + L;
      sV[(b * QW + t) * LDS_V + w] = in ? vv[u] : 0.f;
      sTV[(b * QW + t) * LDS_V + w] = (w >= 1) ? tv[u] : 0.f;
    }
  }
  __syncthreads();

  const int64_t colg = (int64_t)wstart + 4 * kq;  // + 16 q
  const bool fast = wstart + 16 * Q2W_NQ <= a.n;   // window completely inside the matrix
  float4 s[Q2W_NQ];
  for (int slab = blockIdx.x; slab < nslab; slab += gridDim.x) {
    __asm__ volatile("" ::: "memory");  // keep the (slab-invariant) V/T fragment reads inside the loop
    const int64_t row = (int64_t)slab * Q2_SLAB + wave * 16 + n16;
    const bool rok = row < nrows;
    float *base = Zt + (rok ? row * ldz : 0);
    if (fast) {
      const float4 *b4 = reinterpret_cast<const float4 *>(base + colg);
#pragma unroll
      for (int q = 0; q < Q2W_NQ; ++q) s[q] = b4[4 * q];
    } else {
#pragma unroll
      for (int q = 0; q < Q2W_NQ; ++q) {
        const int64_t c = colg + 16 * q;
        const bool ok = c < a.n;  // n % 4 == 0 and c % 4 == 0: all four in or out
        const float4 x = *reinterpret_cast<const float4 *>(ok ? base + c : Zt);
        s[q] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }

    auto apply = [&](auto q0tag, const float *__restrict__ bV, const float *__restrict__ bTV) {
      constexpr int Q0 = decltype(q0tag)::value;
      // ---- W2^T = (T V) S^T: t'-tile ta needs the w-tiles q >= ta  ((T V)[t'][w] is non-zero for w >= t' + 1)
      f32x4 acc2[4];
      const float *pT = bTV + n16 * LDS_V + 4 * kq;
#pragma unroll
      for (int ta = 0; ta < 4; ++ta) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc2[ta][e] = 0.f;
#pragma unroll
        for (int q = ta; q < 8; ++q) {
          const float4 av = *reinterpret_cast<const float4 *>(pT + 16 * ta * LDS_V + 16 * q);
          acc2[ta] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, s[Q0 + q].x, acc2[ta], 0, 0, 0);
          acc2[ta] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, s[Q0 + q].y, acc2[ta], 0, 0, 0);
          acc2[ta] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, s[Q0 + q].z, acc2[ta], 0, 0, 0);
          acc2[ta] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, s[Q0 + q].w, acc2[ta], 0, 0, 0);
        }
      }
      // ---- U^T = V^T W2^T, one w-tile at a time: V[t][w] is non-zero for t + 1 <= w <= t + 64, i.e. t-tiles wt-4 .. wt
      const float *pU = bV + 4 * kq * LDS_V + n16;
#pragma unroll
      for (int wt = 0; wt < 8; ++wt) {
        f32x4 u;
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = 0.f;
        const int lo = wt - 4 > 0 ? wt - 4 : 0, hi = wt < 3 ? wt : 3;
#pragma unroll
        for (int ta = lo; ta <= hi; ++ta) {
          const float *q = pU + 16 * ta * LDS_V + 16 * wt;
          const float a0 = q[0], a1 = q[LDS_V], a2 = q[2 * LDS_V], a3 = q[3 * LDS_V];
          u = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, acc2[ta][0], u, 0, 0, 0);
          u = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, acc2[ta][1], u, 0, 0, 0);
          u = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, acc2[ta][2], u, 0, 0, 0);
          u = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, acc2[ta][3], u, 0, 0, 0);
        }
        float4 &x = s[Q0 + wt];
        x.x -= u[0]; x.y -= u[1]; x.z -= u[2]; x.w -= u[3];
      }
    };
    apply(std::integral_constant<int, 0>{}, sV, sTV);
    if (haveB) apply(std::integral_constant<int, 4>{}, sV + QW * LDS_V, sTV + QW * LDS_V);

    if (rok) {
      if (fast) {
        float4 *b4 = reinterpret_cast<float4 *>(base + colg);
#pragma unroll
        for (int q = 0; q < Q2W_NQ; ++q) b4[4 * q] = s[q];
      } else {
#pragma unroll
        for (int q = 0; q < Q2W_NQ; ++q) {
          const int64_t c = colg + 16 * q;
          if (c < a.n) *reinterpret_cast<float4 *>(base + c) = s[q];
        }
      }
    }
  }
}

constexpr int Q2_LDS_BYTES = 4 * QW * LDS_V * 4;  // V and T V of both blocks: 135 KB

static size_t q2_step_workspace_bytes(int64_t n) {
  const int64_t ngroups = cdiv(n - 2 > 0 ? n - 2 : 1, QW);
  return (size_t)Q2_WIN * 2 * (ngroups + 2) * QW * QWIN * sizeof(float) + 256;
}

// Either form may be selected at run time -- by SHAPE only: the sliding-window form (q2slide.hip) for at least
// VIVIT_Q2_SLIDE_MIN_ROWS rows.  `max_rows` = the largest row count the caller will transform with this workspace; the
// block images of the sliding-window form (up to 2 GB) are only included when such a call could select it (a top-10
// selection at n = 40 960 reserves 0.2 GB instead of 2 GB).  max_rows < 0: unknown (the public mode switch) -> both.
size_t q2_workspace_bytes(int64_t n, int64_t max_rows) {
  const size_t a = q2_step_workspace_bytes(n);
  const size_t b = (max_rows < 0 || q2_slide_possible(max_rows, n)) ? q2_slide_workspace_bytes(n) : 0;
  return a > b ? a : b;
}

// Zt[nrows x n] (ldz) <- Zt * Q2^T
int q2_apply_launch(float *Zt, int64_t ldz, int64_t nrows, int64_t n, const float *R2, int64_t ldr, const float *tau2,
                    void *ws, hipStream_t stream, int mode) {
  if (n < 3) return VIVIT_OK;
  // mode: -1 automatic, 0 block steps (this file), 1 sliding window (q2slide.hip)
  if (mode == 1 || (mode < 0 && q2_slide_ok(nrows, n, Zt, ldz)))
    return q2_slide_launch(Zt, ldz, nrows, n, R2, ldr, tau2, ws, q2_workspace_bytes(n, mode == 1 ? -1 : nrows), stream);
  static unsigned long long attr_done = 0;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return VIVIT_E_LAUNCH;
    if (!(attr_done & (1ull << (dev & 63)))) {
      if (!ensure_dynamic_lds(reinterpret_cast<const void *>(q2_apply_kernel<true>), Q2_LDS_BYTES, attr_done) ||
          !ensure_dynamic_lds(reinterpret_cast<const void *>(q2_apply_kernel<false>), Q2_LDS_BYTES, attr_done) ||
          !ensure_dynamic_lds(reinterpret_cast<const void *>(q2_apply16_kernel), Q2_LDS_BYTES, attr_done))
        return VIVIT_E_LAUNCH;
      attr_done |= 1ull << (dev & 63);
    }
  }
  float *Tbuf = reinterpret_cast<float *>(align_up(reinterpret_cast<uintptr_t>(ws), 256));
  const bool vec = ((reinterpret_cast<uintptr_t>(Zt) & 15) == 0) && (ldz % 4 == 0) && (n % 4 == 0);
  const int nsweeps = (int)(n - 2);
  const int ngroups = (int)cdiv(nsweeps, QW);
  // Kmax of group g: largest level pair K with a reflector for its first sweep g0: g0 + 1 + 2K*QB <= n - 1
  auto Kmax = [&](int g) { return (int)((n - 2 - (int64_t)g * QW) / QB) / 2; };
  Q2Step a;
  a.R2 = R2; a.ldr = ldr; a.tau2 = tau2; a.nk = sb2st_num_levels(n); a.n = (int)n; a.ngroups = ngroups;
  const int tau_max = (ngroups - 1) + Kmax(0);
  const int64_t nslab = cdiv(nrows, Q2_SLAB);
  // T V of the blocks is prepared for Q2_WIN wavefront steps per launch (the per-step prepare launches were 0.1 s
  // of pure latency at n = 40 960), then the steps of the window are applied one launch each
  int G_lo = 0;
  for (int tau0 = 0; tau0 <= tau_max;) {
    Q2Win win;
    int nw = 0, maxblk = 0;
    int64_t off = 0;
    int tau = tau0;
    for (; tau <= tau_max && nw < Q2_WIN; ++tau) {
      // valid G: 0 <= G <= min(tau, ngroups-1) and tau - G <= Kmax(group of G); G + Kmax(G) grows with G
      while (G_lo < ngroups && G_lo + Kmax(ngroups - 1 - G_lo) < tau) ++G_lo;
      const int G_hi = tau < ngroups - 1 ? tau : ngroups - 1;
      if (G_lo > G_hi) continue;
      const int nblk = G_hi - G_lo + 1;
      win.tau[nw] = tau; win.G_lo[nw] = G_lo; win.nblk[nw] = nblk; win.off[nw] = off;
      off += 2 * nblk;
      if (nblk > maxblk) maxblk = nblk;
      ++nw;
    }
    tau0 = tau;
    if (nw == 0) continue;
    for (int i = nw; i < Q2_WIN; ++i) { win.tau[i] = 0; win.G_lo[i] = 0; win.nblk[i] = 0; win.off[i] = 0; }
    q2_prepare_kernel<<<dim3(2 * maxblk, nw), 256, 0, stream>>>(a, win, Tbuf);
    for (int i = 0; i < nw; ++i) {
      a.tau = win.tau[i]; a.G_lo = win.G_lo[i];
      const unsigned nblk = (unsigned)win.nblk[i];
      const float *TV = Tbuf + win.off[i] * QW * QWIN;
      // workgroups = gx * nblk, one per CU at a time: pick the row split whose last round of 256 workgroups is
      // fullest and whose slabs divide evenly, with at least ~4 slabs per V/T load (matters in row-range mode)
      int64_t gx = 1;
      {
        const int64_t hi = cdiv(nslab, 4) < cdiv(4096, nblk) ? cdiv(nslab, 4) : cdiv(4096, nblk);
        double best = -1.0;
        for (int64_t c = 1; c <= (hi < 1 ? 1 : hi); ++c) {
          const int64_t wgs = c * nblk;
          const double fill = (double)wgs / (double)(256 * cdiv(wgs, 256));
          const double even = ((double)nslab / (double)c) / (double)cdiv(nslab, c);
          const double enough = wgs >= 512 ? 1.0 : (double)wgs / 512.0;
          const double score = fill * even * enough;
          if (score > best + 1e-9) { best = score; gx = c; }
        }
      }
      static int wave16 = -1;
      if (wave16 < 0) { const char *e = getenv("VIVIT_Q2_WAVE16"); wave16 = e ? atoi(e) : 1; }   // (0: the 32-row form)
      if (vec && wave16)
        q2_apply16_kernel<<<dim3((unsigned)gx, nblk), Q2W_THREADS, Q2_LDS_BYTES, stream>>>(a, TV, Zt, ldz, (int)nrows);
      else if (vec)
        q2_apply_kernel<true><<<dim3((unsigned)gx, nblk), Q2_THREADS, Q2_LDS_BYTES, stream>>>(a, TV, Zt, ldz, (int)nrows);
      else
        q2_apply_kernel<false><<<dim3((unsigned)gx, nblk), Q2_THREADS, Q2_LDS_BYTES, stream>>>(a, TV, Zt, ldz, (int)nrows);
    }
  }
  return launch_status();
}

} // namespace vivit

using namespace vivit;

extern "C" {

size_t vivit_q2_apply_f32_workspace_bytes(int64_t n) { return n < 3 ? 0 : q2_workspace_bytes(n, -1) + 512; }

int vivit_q2_apply_f32(float *Zt, int64_t ldz, int64_t nrows, int64_t n, const float *R2, int64_t ldr, const float *tau2,
                       void *workspace, size_t workspace_bytes, int mode, void *stream) {
  if (n < 0 || nrows < 0 || mode < -1 || mode > 1) return VIVIT_E_BADARG;
  if (n < 3 || nrows == 0) return VIVIT_OK;
  if (!Zt || !R2 || !tau2 || ldz < n || ldr < n) return VIVIT_E_BADARG;
  if (!workspace || workspace_bytes < vivit_q2_apply_f32_workspace_bytes(n)) return VIVIT_E_WORKSPACE;
  if (mode == 1 && (n < 128 || n % 4 != 0 || ldz % 4 != 0 || (reinterpret_cast<uintptr_t>(Zt) & 15) != 0)) return VIVIT_E_UNSUPPORTED;
  void *ws = reinterpret_cast<void *>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  return q2_apply_launch(Zt, ldz, nrows, n, R2, ldr, tau2, ws, static_cast<hipStream_t>(stream), mode);
}

} // extern "C"
