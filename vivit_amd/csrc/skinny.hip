// K7 / K8 with few directions: out[K, P] = alpha * coef[K, n] @ V[n, P] + beta * out, K <= 16.
//
// HBM-bound: V (n x P, up to 66 GB) is streamed exactly once with 16-byte loads, 8 rows in flight
// per lane; the K coefficient rows are wave-uniform scalars.  The grid tiles P (1024 columns per
// workgroup) x n (row chunks); chunk partials go to a slab and are summed in a fixed order, so the
// result is bit-reproducible.  Replaces the m <= 16 case of the MFMA GEMM, which would waste
// 128/m of its matrix work on an output this skinny (vivit/optim/directional_damped_newton.py:370-373,
// vivit/utils/ggn.py:94-115 with few selected eigenvectors).
#include "common.h"

namespace vivit {

constexpr int SK_MAXK = 16;
constexpr int SK_COLS = 1024;   // columns per workgroup (256 threads x float4)
constexpr int SK_ROWS = 1024;   // rows per chunk

constexpr int SK_SUB = 256;  // rows whose coefficients are staged in LDS at a time

template <int KK, bool VEC>
__global__ __launch_bounds__(256) void skinny_nn_kernel(const float *__restrict__ coef, int64_t ldc_,
                                                        const float *__restrict__ V, int64_t ldv, int64_t n, int64_t P,
                                                        float *__restrict__ part, int Kact) {
  __shared__ __attribute__((aligned(16))) float sc[SK_SUB * KK];  // [row][k]: one broadcast read per row
  const int tid = threadIdx.x;
  const int64_t c = (int64_t)blockIdx.x * SK_COLS + 4 * tid;
  const int64_t r0 = (int64_t)blockIdx.y * SK_ROWS;
  const int64_t r1 = r0 + SK_ROWS < n ? r0 + SK_ROWS : n;
  float4 acc[KK];
#pragma unroll
  for (int k = 0; k < KK; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool colok = c < P;
  const int64_t cc = colok ? c : 0;
  for (int64_t s0 = r0; s0 < r1; s0 += SK_SUB) {
    __syncthreads();
    // stage coef[k][s0 + rr] -> sc[rr][k] (zero beyond the chunk / beyond Kact)
    for (int idx = tid; idx < SK_SUB * KK; idx += 256) {
      const int k = idx / SK_SUB, rr = idx - k * SK_SUB;
      const int64_t row = s0 + rr;
      const bool ok = k < Kact && row < r1;
      const float a = coef[(int64_t)(ok ? k : 0) * ldc_ + (ok ? row : r0)];
      sc[rr * KK + k] = ok ? a : 0.f;
    }
    __syncthreads();
    const int64_t send = s0 + SK_SUB < r1 ? s0 + SK_SUB : r1;
    for (int64_t i = s0; i < send; i += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t row = (i + u < send) ? i + u : send - 1;   // clamped duplicate, its coefficients are zero
        if constexpr (VEC) {
          v[u] = *reinterpret_cast<const float4 *>(V + row * ldv + cc);
        } else {
          const float *q = V + row * ldv;
          v[u] = make_float4(q[cc], q[cc + 1 < P ? cc + 1 : cc], q[cc + 2 < P ? cc + 2 : cc], q[cc + 3 < P ? cc + 3 : cc]);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rr = (int)(i - s0) + u;   // < SK_SUB + 8; rows past `send` hold zeros or are clamped below
        const float *crow = sc + (rr < SK_SUB ? rr : SK_SUB - 1) * KK;
        const float live = (i + u < send) ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < KK; ++k) {
          const float a = crow[k] * live;
          acc[k].x += a * v[u].x; acc[k].y += a * v[u].y; acc[k].z += a * v[u].z; acc[k].w += a * v[u].w;
        }
      }
    }
  }
  if (!colok) return;
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    if (k < Kact) {
      float *o = part + ((int64_t)blockIdx.y * Kact + k) * P + c;
      if (VEC) {
        *reinterpret_cast<float4 *>(o) = acc[k];
      } else {
        o[0] = acc[k].x;
        if (c + 1 < P) o[1] = acc[k].y;
        if (c + 2 < P) o[2] = acc[k].z;
        if (c + 3 < P) o[3] = acc[k].w;
      }
    }
  }
}

__global__ __launch_bounds__(256) void skinny_reduce_kernel(const float *__restrict__ part, float *__restrict__ out,
                                                            int64_t ldo, int K, int64_t P, int nchunk, float alpha,
                                                            float beta) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)K * P) return;
  const int64_t k = idx / P, c = idx - k * P;
  float s = 0.f;
  for (int ch = 0; ch < nchunk; ++ch) s += part[((int64_t)ch * K + k) * P + c];
  float v = alpha * s;
  if (beta != 0.f) v += beta * out[k * ldo + c];
  out[k * ldo + c] = v;
}

size_t skinny_workspace_bytes(int64_t K, int64_t n, int64_t P) {
  return (size_t)cdiv(n, SK_ROWS) * (size_t)K * (size_t)P * sizeof(float);
}

bool skinny_applicable(int64_t M, int64_t N, int64_t K) { return M >= 1 && M <= SK_MAXK && K >= 1 && N >= 1; }

template <int KK>
static void skinny_dispatch(bool vec, dim3 grid, hipStream_t s, const float *coef, int64_t ldc_, const float *V, int64_t ldv,
                            int64_t n, int64_t P, float *part, int Kact) {
  if (vec) skinny_nn_kernel<KK, true><<<grid, 256, 0, s>>>(coef, ldc_, V, ldv, n, P, part, Kact);
  else skinny_nn_kernel<KK, false><<<grid, 256, 0, s>>>(coef, ldc_, V, ldv, n, P, part, Kact);
}

// out[K x P] = alpha * coef[K x n] (ld ldc_) @ V[n x P] (ld ldv) + beta * out
int skinny_nn_launch(const float *coef, int64_t ldc_, const float *V, int64_t ldv, float *out, int64_t ldo, int64_t K,
                     int64_t n, int64_t P, float alpha, float beta, void *ws, size_t ws_bytes, hipStream_t stream) {
  if (!ws || ws_bytes < skinny_workspace_bytes(K, n, P)) return VIVIT_E_WORKSPACE;
  const int nchunk = (int)cdiv(n, SK_ROWS);
  if (nchunk > 65535) return VIVIT_E_UNSUPPORTED;
  float *part = static_cast<float *>(ws);
  const bool vec = ((reinterpret_cast<uintptr_t>(V) & 15) == 0) && (ldv % 4 == 0) && (P % 4 == 0);
  dim3 grid((unsigned)cdiv(P, SK_COLS), (unsigned)nchunk);
  const int kk = K <= 1 ? 1 : K <= 2 ? 2 : K <= 4 ? 4 : K <= 8 ? 8 : 16;
  // rows of coef beyond K are never read: the kernel is instantiated for kk >= K but the slab is
  // laid out with kk rows per chunk
  switch (kk) {
    case 1: skinny_dispatch<1>(vec, grid, stream, coef, ldc_, V, ldv, n, P, part, (int)K); break;
    case 2: skinny_dispatch<2>(vec, grid, stream, coef, ldc_, V, ldv, n, P, part, (int)K); break;
    case 4: skinny_dispatch<4>(vec, grid, stream, coef, ldc_, V, ldv, n, P, part, (int)K); break;
    case 8: skinny_dispatch<8>(vec, grid, stream, coef, ldc_, V, ldv, n, P, part, (int)K); break;
    default: skinny_dispatch<16>(vec, grid, stream, coef, ldc_, V, ldv, n, P, part, (int)K); break;
  }
  if (launch_status() != VIVIT_OK) return VIVIT_E_LAUNCH;
  skinny_reduce_kernel<<<(unsigned)cdiv(K * P, 256), 256, 0, stream>>>(part, out, ldo, (int)K, P, nchunk, alpha, beta);
  return launch_status();
}

} // namespace vivit
