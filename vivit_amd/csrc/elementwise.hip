// HBM-bound epilogue kernels of the Gram path (K1', K5/K6 epilogues, K10) for gfx950.
// All are grid-stride, 16-byte vectorised where the layout allows, and free of float atomics.
#include "common.h"
#include "eig_internal.h"

namespace vivit {

constexpr int EW_BLOCK = 256;
static inline unsigned ew_grid(int64_t work) {
  int64_t g = cdiv(work, EW_BLOCK);
  if (g > 2048) g = 2048;  // ~8 blocks per CU, grid-stride the rest
  if (g < 1) g = 1;
  return (unsigned)g;
}

// G[c,n,d,m] = alpha * Gz[n,m] * Gs[c,n,d,m] + beta * G[c,n,d,m]   (vec4 along m when N % 4 == 0)
template <bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void hadamard_kernel(const float *__restrict__ Gz, const float *__restrict__ Gs,
                                                            float *__restrict__ G, int64_t C, int64_t N, float alpha,
                                                            float beta) {
  const int64_t n = C * N;
  const int64_t total = VEC ? (n * n) >> 2 : n * n;
  for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * EW_BLOCK) {
    const int64_t flat = VEC ? idx << 2 : idx;
    const int64_t row = flat / n, col = flat - row * n;  // row = c*N + nn, col = d*N + mm
    const int64_t nn = row % N, mm = col % N;
    if (VEC) {
      const float4 z = *reinterpret_cast<const float4 *>(Gz + nn * N + mm);
      const float4 s = *reinterpret_cast<const float4 *>(Gs + flat);
      float4 o = make_float4(alpha * z.x * s.x, alpha * z.y * s.y, alpha * z.z * s.z, alpha * z.w * s.w);
      if (beta != 0.f) {
        const float4 g = *reinterpret_cast<const float4 *>(G + flat);
        o.x += beta * g.x; o.y += beta * g.y; o.z += beta * g.z; o.w += beta * g.w;
      }
      *reinterpret_cast<float4 *>(G + flat) = o;
    } else {
      float o = alpha * Gz[nn * N + mm] * Gs[flat];
      if (beta != 0.f) o += beta * G[flat];
      G[flat] = o;
    }
  }
}

// Rectangular block of a factorised product: rows (c, nn) with c < Cr, nn < Nr, columns (d, mm) with d < Cc, mm < Nc:
//   G[(c,nn), (d,mm)] = alpha * Gz[nn, mm] * Gs[(c,nn), (d,mm)] + beta * G[...]      (G: ldg, Gs / Gz contiguous)
// Serves the block row of a batch shard (Nr = local samples, Nc = all samples) and V^T g of a factorised
// Linear weight (Cc = 1: the per-sample gradient has no class axis).
template <bool VEC>
__global__ __launch_bounds__(EW_BLOCK) void hadamard_block_kernel(const float *__restrict__ Gz, const float *__restrict__ Gs,
                                                                  float *__restrict__ G, int64_t Cr, int64_t Nr, int64_t Cc,
                                                                  int64_t Nc, int64_t ldg, float alpha, float beta) {
  const int64_t rows = Cr * Nr, cols = Cc * Nc;
  const int64_t total = VEC ? (rows * cols) >> 2 : rows * cols;
  for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * EW_BLOCK) {
    const int64_t flat = VEC ? idx << 2 : idx;
    const int64_t row = flat / cols, col = flat - row * cols;
    const int64_t nn = row % Nr, mm = col % Nc;
    float *g = G + row * ldg + col;
    if (VEC) {
      const float4 z = *reinterpret_cast<const float4 *>(Gz + nn * Nc + mm);
      const float4 sv = *reinterpret_cast<const float4 *>(Gs + flat);
      float4 o = make_float4(alpha * z.x * sv.x, alpha * z.y * sv.y, alpha * z.z * sv.z, alpha * z.w * sv.w);
      if (beta != 0.f) {
        const float4 old = *reinterpret_cast<const float4 *>(g);
        o.x += beta * old.x; o.y += beta * old.y; o.z += beta * old.z; o.w += beta * old.w;
      }
      *reinterpret_cast<float4 *>(g) = o;
    } else {
      float o = alpha * Gz[nn * Nc + mm] * Gs[flat];
      if (beta != 0.f) o += beta * *g;
      *g = o;
    }
  }
}

// Length-C contractions of the factorised Linear products (vivit/extensions/secondorder/vivit/linear.py:53,64):
//   T[f, o, nn] = sum_c mat[f, c, nn] * s[c, nn, o]         ("vcn,cno->von"; then V mat = T z on MFMA)
__global__ __launch_bounds__(EW_BLOCK) void class_contract_kernel(const float *__restrict__ mat, const float *__restrict__ s,
                                                                  float *__restrict__ T, int64_t F, int64_t C, int64_t N,
                                                                  int64_t O) {
  const int64_t total = F * O * N;
  for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * EW_BLOCK) {
    const int64_t nn = idx % N, o = (idx / N) % O, f = idx / (N * O);
    float acc = 0.f;
    for (int64_t c = 0; c < C; ++c) acc += mat[(f * C + c) * N + nn] * s[(c * N + nn) * O + o];
    T[idx] = acc;
  }
}
//   R[f, c, nn] = sum_o s[c, nn, o] * U[f, o, nn]           ("cno,von->vcn"; U = mat z^T from MFMA)
__global__ __launch_bounds__(EW_BLOCK) void class_expand_kernel(const float *__restrict__ s, const float *__restrict__ U,
                                                                float *__restrict__ R, int64_t F, int64_t C, int64_t N,
                                                                int64_t O) {
  const int64_t total = F * C * N;
  for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * EW_BLOCK) {
    const int64_t nn = idx % N, c = (idx / N) % C, f = idx / (N * C);
    const float *srow = s + (c * N + nn) * O;
    float acc = 0.f;
    for (int64_t o = 0; o < O; ++o) acc += srow[o] * U[(f * O + o) * N + nn];
    R[idx] = acc;
  }
}

// lambdas[nn, k] = scale * sum_c GE[(c*N + nn), k]^2 / evals[k]
__global__ __launch_bounds__(EW_BLOCK) void dir_curvature_kernel(const float *__restrict__ GE,
                                                                 const float *__restrict__ evals,
                                                                 float *__restrict__ lambdas, int64_t C, int64_t N,
                                                                 int64_t K, float scale) {
  const int64_t total = N * K;
  for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * EW_BLOCK) {
    const int64_t nn = idx / K, k = idx - nn * K;
    float acc = 0.f;
    for (int64_t c = 0; c < C; ++c) {
      const float v = GE[(c * N + nn) * K + k];
      acc += v * v;
    }
    lambdas[idx] = scale * acc / evals[k];
  }
}

// X[r, k] *= pre / sqrt(evals[k])
__global__ __launch_bounds__(EW_BLOCK) void scale_cols_rsqrt_kernel(float *__restrict__ X,
                                                                    const float *__restrict__ evals, int64_t rows,
                                                                    int64_t K, int64_t ldx, float pre) {
  const int64_t total = rows * K;
  for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * EW_BLOCK) {
    const int64_t r = idx / K, k = idx - r * K;
    X[r * ldx + k] *= pre / sqrtf(evals[k]);
  }
}

// part[k][b] = sum over block b's slice of X[k, :]^2; fixed-order second pass adds into acc[k].
constexpr int SQN_CHUNK = 8192;
__global__ __launch_bounds__(EW_BLOCK) void row_sqnorm_part_kernel(const float *__restrict__ X, float *__restrict__ part,
                                                                   int64_t len, int nchunk) {
  __shared__ float red[4];
  const int64_t k = blockIdx.y;
  const int64_t j0 = (int64_t)blockIdx.x * SQN_CHUNK;
  const int64_t j1 = j0 + SQN_CHUNK < len ? j0 + SQN_CHUNK : len;
  const float *row = X + k * len;
  float s = 0.f;
  for (int64_t j = j0 + threadIdx.x; j < j1; j += EW_BLOCK) {
    const float v = row[j];
    s += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[k * nchunk + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(EW_BLOCK) void row_sqnorm_final_kernel(const float *__restrict__ part, float *__restrict__ acc,
                                                                    int64_t K, int nchunk) {
  const int64_t k = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x;
  if (k >= K) return;
  float s = 0.f;
  for (int b = 0; b < nchunk; ++b) s += part[k * nchunk + b];
  acc[k] += s;
}

__global__ __launch_bounds__(EW_BLOCK) void scale_rows_rsqrt_kernel(float *__restrict__ X, const float *__restrict__ acc,
                                                                    int64_t K, int64_t len) {
  const int64_t total = K * len;
  for (int64_t idx = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * EW_BLOCK) {
    const int64_t k = idx / len;
    X[idx] *= 1.f / sqrtf(acc[k]);
  }
}

// G[i][j] = G[j][i] for j > i, 32x32 tiles transposed through LDS (coalesced both ways).
__global__ __launch_bounds__(256) void symmetrize_kernel(float *__restrict__ G, int64_t n, int64_t ldg) {
  __shared__ float t[32][33];
  // blockIdx.x enumerates tile pairs (bi >= bj) of the lower triangle.
  const int64_t tb = blockIdx.x;
  int64_t bi = (int64_t)((sqrt(8.0 * (double)tb + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= tb) ++bi;
  while (bi * (bi + 1) / 2 > tb) --bi;
  const int64_t bj = tb - bi * (bi + 1) / 2;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = bi * 32 + r, j = bj * 32 + tx;
    t[r][tx] = (i < n && j < n) ? G[i * ldg + j] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = bj * 32 + r, j = bi * 32 + tx;  // destination in the upper triangle
    if (i < n && j < n && j > i) G[i * ldg + j] = t[tx][r];
  }
}

int symmetrize_launch(float *G, int64_t n, int64_t ldg, hipStream_t stream) {
  const int64_t nb = cdiv(n, 32);
  symmetrize_kernel<<<(unsigned)(nb * (nb + 1) / 2), 256, 0, stream>>>(G, n, ldg);
  return launch_status();
}

// Lower triangle (diagonal included) of a symmetric matrix <-> packed row-major vector, packed[i (i + 1) / 2 + j] = G[i][j]
// for j <= i: what the ranks all-reduce instead of the full partial Gram matrices (half the bytes over xGMI).
// grid.y = row, grid.x = 256-column blocks; both sides coalesced.
__global__ __launch_bounds__(256) void pack_lower_kernel(const float *__restrict__ G, int64_t r0, int64_t ldg,
                                                         float *__restrict__ packed) {
  const int64_t i = r0 + blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j <= i) packed[i * (i + 1) / 2 + j] = G[i * ldg + j];
}
__global__ __launch_bounds__(256) void unpack_lower_kernel(const float *__restrict__ packed, int64_t r0, float *__restrict__ G,
                                                           int64_t ldg) {
  const int64_t i = r0 + blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j <= i) G[i * ldg + j] = packed[i * (i + 1) / 2 + j];
}

} // namespace vivit

using namespace vivit;

extern "C" {

int vivit_gram_hadamard_f32(const float *Gz, const float *Gs, float *G, int64_t C, int64_t N, float alpha, float beta,
                            void *stream) {
  if (C < 0 || N < 0) return VIVIT_E_BADARG;
  if (C == 0 || N == 0) return VIVIT_OK;
  if (!Gz || !Gs || !G) return VIVIT_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t n = C * N;
  const bool vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(Gz) | reinterpret_cast<uintptr_t>(Gs) |
                                     reinterpret_cast<uintptr_t>(G)) & 15) == 0;
  if (vec)
    hadamard_kernel<true><<<ew_grid((n * n) >> 2), EW_BLOCK, 0, s>>>(Gz, Gs, G, C, N, alpha, beta);
  else
    hadamard_kernel<false><<<ew_grid(n * n), EW_BLOCK, 0, s>>>(Gz, Gs, G, C, N, alpha, beta);
  return launch_status();
}

int vivit_gram_hadamard_block_f32(const float *Gz, const float *Gs, float *G, int64_t Cr, int64_t Nr, int64_t Cc,
                                  int64_t Nc, int64_t ldg, float alpha, float beta, void *stream) {
  if (Cr < 0 || Nr < 0 || Cc < 0 || Nc < 0 || ldg < Cc * Nc) return VIVIT_E_BADARG;
  if (Cr == 0 || Nr == 0 || Cc == 0 || Nc == 0) return VIVIT_OK;
  if (!Gz || !Gs || !G) return VIVIT_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t total = Cr * Nr * Cc * Nc;
  const bool vec = (Nc % 4 == 0) && (ldg % 4 == 0) && ((reinterpret_cast<uintptr_t>(Gz) | reinterpret_cast<uintptr_t>(Gs) |
                                                       reinterpret_cast<uintptr_t>(G)) & 15) == 0;
  if (vec)
    hadamard_block_kernel<true><<<ew_grid(total >> 2), EW_BLOCK, 0, s>>>(Gz, Gs, G, Cr, Nr, Cc, Nc, ldg, alpha, beta);
  else
    hadamard_block_kernel<false><<<ew_grid(total), EW_BLOCK, 0, s>>>(Gz, Gs, G, Cr, Nr, Cc, Nc, ldg, alpha, beta);
  return launch_status();
}

int vivit_class_contract_f32(const float *mat, const float *s, float *T, int64_t F, int64_t C, int64_t N, int64_t O,
                             void *stream) {
  if (F < 0 || C < 0 || N < 0 || O < 0) return VIVIT_E_BADARG;
  if (F == 0 || N == 0 || O == 0) return VIVIT_OK;
  if (!T || (C > 0 && (!mat || !s))) return VIVIT_E_BADARG;
  class_contract_kernel<<<ew_grid(F * O * N), EW_BLOCK, 0, static_cast<hipStream_t>(stream)>>>(mat, s, T, F, C, N, O);
  return launch_status();
}

int vivit_class_expand_f32(const float *s, const float *U, float *R, int64_t F, int64_t C, int64_t N, int64_t O,
                           void *stream) {
  if (F < 0 || C < 0 || N < 0 || O < 0) return VIVIT_E_BADARG;
  if (F == 0 || C == 0 || N == 0) return VIVIT_OK;
  if (!R || (O > 0 && (!s || !U))) return VIVIT_E_BADARG;
  class_expand_kernel<<<ew_grid(F * C * N), EW_BLOCK, 0, static_cast<hipStream_t>(stream)>>>(s, U, R, F, C, N, O);
  return launch_status();
}

int vivit_dir_curvature_f32(const float *GE, const float *evals, float *lambdas, int64_t C, int64_t N, int64_t K,
                            float scale, void *stream) {
  if (C < 0 || N < 0 || K < 0) return VIVIT_E_BADARG;
  if (N == 0 || K == 0) return VIVIT_OK;
  if (!GE || !evals || !lambdas) return VIVIT_E_BADARG;
  dir_curvature_kernel<<<ew_grid(N * K), EW_BLOCK, 0, static_cast<hipStream_t>(stream)>>>(GE, evals, lambdas, C, N, K,
                                                                                          scale);
  return launch_status();
}

int vivit_scale_cols_rsqrt_f32(float *X, const float *evals, int64_t rows, int64_t K, int64_t ldx, float pre,
                               void *stream) {
  if (rows < 0 || K < 0 || ldx < K) return VIVIT_E_BADARG;
  if (rows == 0 || K == 0) return VIVIT_OK;
  if (!X || !evals) return VIVIT_E_BADARG;
  scale_cols_rsqrt_kernel<<<ew_grid(rows * K), EW_BLOCK, 0, static_cast<hipStream_t>(stream)>>>(X, evals, rows, K, ldx,
                                                                                                pre);
  return launch_status();
}

// acc[k] += ||X[k, :]||^2 via per-chunk partials in `workspace` (fixed summation order).
int vivit_row_sqnorm_acc_f32(const float *X, float *acc, int64_t K, int64_t len, void *workspace,
                                size_t workspace_bytes, void *stream) {
  if (K < 0 || len < 0) return VIVIT_E_BADARG;
  if (K == 0 || len == 0) return VIVIT_OK;
  if (!X || !acc) return VIVIT_E_BADARG;
  const int nchunk = (int)cdiv(len, SQN_CHUNK);
  const size_t need = (size_t)K * nchunk * sizeof(float);
  if (!workspace || workspace_bytes < need) return VIVIT_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float *part = static_cast<float *>(workspace);
  for (int64_t k0 = 0; k0 < K; k0 += 65535) {  // grid.y is limited to 65535 rows per launch
    const int64_t kc = (K - k0 < 65535) ? K - k0 : 65535;
    row_sqnorm_part_kernel<<<dim3((unsigned)nchunk, (unsigned)kc), EW_BLOCK, 0, s>>>(X + k0 * len, part + k0 * nchunk, len,
                                                                                    nchunk);
  }
  row_sqnorm_final_kernel<<<(unsigned)cdiv(K, EW_BLOCK), EW_BLOCK, 0, s>>>(part, acc, K, nchunk);
  return launch_status();
}

size_t vivit_row_sqnorm_workspace_bytes(int64_t K, int64_t len) {
  if (K <= 0 || len <= 0) return 0;
  return (size_t)K * (size_t)cdiv(len, SQN_CHUNK) * sizeof(float);
}

int vivit_scale_rows_rsqrt_f32(float *X, const float *acc, int64_t K, int64_t len, void *stream) {
  if (K < 0 || len < 0) return VIVIT_E_BADARG;
  if (K == 0 || len == 0) return VIVIT_OK;
  if (!X || !acc) return VIVIT_E_BADARG;
  scale_rows_rsqrt_kernel<<<ew_grid(K * len), EW_BLOCK, 0, static_cast<hipStream_t>(stream)>>>(X, acc, K, len);
  return launch_status();
}

int vivit_symmetrize_lower_f32(float *G, int64_t n, int64_t ldg, void *stream) {
  if (n < 0 || ldg < n) return VIVIT_E_BADARG;
  if (n == 0) return VIVIT_OK;
  if (!G) return VIVIT_E_BADARG;
  return symmetrize_launch(G, n, ldg, static_cast<hipStream_t>(stream));
}

int vivit_pack_lower_f32(const float *G, int64_t n, int64_t ldg, float *packed, void *stream) {
  if (n < 0 || ldg < n) return VIVIT_E_BADARG;
  if (n == 0) return VIVIT_OK;
  if (!G || !packed) return VIVIT_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int64_t r0 = 0; r0 < n; r0 += 65535) {  // grid.y limit
    const int64_t rc = (n - r0) < 65535 ? n - r0 : 65535;
    pack_lower_kernel<<<dim3((unsigned)cdiv(r0 + rc, 256), (unsigned)rc), 256, 0, st>>>(G, r0, ldg, packed);
  }
  return launch_status();
}

int vivit_unpack_lower_f32(const float *packed, int64_t n, float *G, int64_t ldg, void *stream) {
  if (n < 0 || ldg < n) return VIVIT_E_BADARG;
  if (n == 0) return VIVIT_OK;
  if (!G || !packed) return VIVIT_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int64_t r0 = 0; r0 < n; r0 += 65535) {
    const int64_t rc = (n - r0) < 65535 ? n - r0 : 65535;
    unpack_lower_kernel<<<dim3((unsigned)cdiv(r0 + rc, 256), (unsigned)rc), 256, 0, st>>>(packed, r0, G, ldg);
  }
  int s2 = launch_status();
  if (s2 != VIVIT_OK) return s2;
  return symmetrize_launch(G, n, ldg, st);
}

} // extern "C"
