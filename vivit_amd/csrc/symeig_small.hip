// Single-workgroup symmetric eigensolver for n <= SMALL_N_MAX, matrix resident in LDS.
//
//   stage 1  Householder tridiagonalisation  A = Q T Q^T   (LAPACK ssytd2 ordering, lower)
//   stage 2  Q formed in place                              (sorg2r on the shifted reflectors)
//   stage 3  implicit-shift QL on (d, e), rotations applied to the rows of Q held one per thread
//
// Cooperation model: 256 threads; the O(1)-per-step scalar recurrences (reflector scalars, the
// QL shift and its chain of plane rotations) are computed REDUNDANTLY by every lane instead of
// being broadcast -- same instruction stream, same inputs, bit-identical results -- so stage 3
// needs no barrier at all: every wavefront keeps a private copy of (d, e) in LDS and each lane
// applies the rotations to its own row of Z.  Stage 1/2 use workgroup barriers around the
// reductions (norm, p^T v) and the rank-2 update.
//
// Replaces Tensor.symeig for the Gram sizes of the reference's own test problems
// (vivit/linalg/eigvalsh.py:221, vivit/linalg/eigh.py:248-250) and serves as the accuracy anchor
// for the multi-kernel path (symeig_large.hip).
#include "common.h"
#include "device_utils.h"

namespace vivit {

__global__ __launch_bounds__(256) void symeig_small_kernel(const float *__restrict__ Ag, int64_t lda, int n,
                                                           float *__restrict__ wout, float *__restrict__ Zg,
                                                           int64_t ldz, int32_t *__restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int LD = n | 1;
  float *A = sm;
  float *dw = A + n * LD;  // [4][n] per-wave diagonal
  float *ew = dw + 4 * n;  // [4][n] per-wave off-diagonal
  float *pv = ew + 4 * n;  // [n]
  float *wv = pv + n;      // [n]
  float *tau = wv + n;     // [n]
  float *red = tau + n;    // [4]
  int *rank = reinterpret_cast<int *>(red + 4);  // [n]
  const int tid = threadIdx.x, wave = tid >> 6;
  float *d = dw + wave * n, *e = ew + wave * n;
  const bool wantz = Zg != nullptr;

  // ---- load (lower triangle is the source of truth), find the scale.
  float amax = 0.f;
  bool bad = false;
  for (int idx = tid; idx < n * n; idx += 256) {
    const int i = idx / n, j = idx - i * n;
    const float v = (j <= i) ? Ag[(int64_t)i * lda + j] : Ag[(int64_t)j * lda + i];
    A[i * LD + j] = v;
    amax = fmaxf(amax, fabsf(v));
    bad |= !(fabsf(v) <= 3.0e38f);  // NaN or inf
  }
  amax = block_max(amax, red, tid);
  const float anybad = block_max(bad ? 1.f : 0.f, red, tid);
  if (anybad != 0.f) {
    // Non-finite input: report failure the way LAPACK's non-convergence surfaces in the
    // reference (RuntimeError in vivit/utils/eig.py:37-40); outputs are NaN.
    const float qnan = __builtin_nanf("");
    for (int i = tid; i < n; i += 256) wout[i] = qnan;
    if (wantz)
      for (int idx = tid; idx < n * n; idx += 256) Zg[(int64_t)(idx / n) * ldz + idx % n] = qnan;
    if (tid == 0) *info = n;
    return;
  }
  // LAPACK ssyev-style conditional scaling into [rmin, rmax].
  const float rmin = 4.4408921e-16f, rmax = 2.2517998e15f;
  float sigma = 1.f;
  if (amax > 0.f && amax < rmin) sigma = rmin / amax;
  else if (amax > rmax) sigma = rmax / amax;
  if (sigma != 1.f) {
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += 256) A[(idx / n) * LD + idx % n] *= sigma;
  }
  __syncthreads();

  // ---- stage 1: tridiagonalisation.
  for (int j = 0; j + 2 < n; ++j) {
    const int m = n - j - 1;  // rows j+1 .. n-1
    // norm of x[1:] = A[j+2:, j]
    float ss = 0.f;
    for (int k = tid + 1; k < m; k += 256) {
      const float x = A[(j + 1 + k) * LD + j];
      ss += x * x;
    }
    ss = block_sum(ss, red, tid);
    const float alpha = A[(j + 1) * LD + j];
    float tj = 0.f, ej = alpha, scal = 0.f;
    if (ss > 0.f) {
      const float beta = -copysignf(sqrtf(alpha * alpha + ss), alpha);
      tj = (beta - alpha) / beta;
      scal = 1.f / (alpha - beta);
      ej = beta;
    }
    d[j] = A[j * LD + j];
    e[j] = ej;
    __syncthreads();  // everyone has read alpha / the column before it is rescaled
    if (tid == 0) { tau[j] = tj; A[(j + 1) * LD + j] = 1.f; }
    if (tj != 0.f) {
      for (int k = tid + 1; k < m; k += 256) A[(j + 1 + k) * LD + j] *= scal;
      __syncthreads();
      // p = tau * A22 v      (thread per row; v = A[j+1:, j] is a broadcast read)
      float pi = 0.f;
      if (tid < m) {
        const float *row = A + (j + 1 + tid) * LD + (j + 1);
        float acc = 0.f;
        for (int k = 0; k < m; ++k) acc += row[k] * A[(j + 1 + k) * LD + j];
        pi = tj * acc;
        pv[tid] = pi;
      }
      const float vi = tid < m ? A[(j + 1 + tid) * LD + j] : 0.f;
      const float pdotv = block_sum(pi * vi, red, tid);
      const float a2 = -0.5f * tj * pdotv;
      if (tid < m) wv[tid] = pi + a2 * vi;
      __syncthreads();
      // A22 -= v w^T + w v^T   (both triangles)
      if (tid < m) {
        float *row = A + (j + 1 + tid) * LD + (j + 1);
        const float wi = wv[tid];
        for (int k = 0; k < m; ++k) row[k] -= vi * wv[k] + wi * A[(j + 1 + k) * LD + j];
      }
    }
    __syncthreads();
  }
  if (n >= 2) {
    d[n - 2] = A[(n - 2) * LD + (n - 2)];
    e[n - 2] = A[(n - 1) * LD + (n - 2)];
  }
  d[n - 1] = A[(n - 1) * LD + (n - 1)];
  e[n - 1] = 0.f;
  __syncthreads();

  // ---- stage 2: Q in place (only when vectors are wanted).
  if (wantz) {
    if (n >= 3) {
      // shift reflector j from column j to column j+1 (right to left), B = A[1:, 1:].
      for (int j = n - 3; j >= 0; --j) {
        for (int k = tid; k < n - j - 2; k += 256) A[(j + 2 + k) * LD + (j + 1)] = A[(j + 2 + k) * LD + j];
        __syncthreads();
      }
    }
    // first row / column of Q and the last column of B are unit vectors.
    for (int k = tid; k < n; k += 256) {
      A[k] = (k == 0) ? 1.f : 0.f;       // row 0
      A[k * LD] = (k == 0) ? 1.f : 0.f;  // column 0
      if (k >= 1) A[k * LD + (n - 1)] = (k == n - 1) ? 1.f : 0.f;
    }
    __syncthreads();
    // B is (n-1)x(n-1) at A[1:,1:]; reflector i (i = 0..n-3) sits in B[i+1:, i], tau[i].
    for (int i = n - 3; i >= 0; --i) {
      const float ti_ = tau[i];
      const int nb = n - 1;
      // apply H(i) to B[i:, i+1:] from the left: thread per column c.
      const int ncol = nb - i - 1;
      if (tid < ncol) {
        const int c = 1 + i + 1 + tid;   // global column
        float wdot = A[(1 + i) * LD + c];  // v[0] = 1
        for (int k = i + 1; k < nb; ++k) wdot += A[(1 + k) * LD + (1 + i)] * A[(1 + k) * LD + c];
        wdot *= ti_;
        A[(1 + i) * LD + c] -= wdot;
        for (int k = i + 1; k < nb; ++k) A[(1 + k) * LD + c] -= wdot * A[(1 + k) * LD + (1 + i)];
      }
      __syncthreads();
      // column i of B: -tau v below the diagonal, 1 - tau on it, zeros above.
      for (int k = tid; k < nb; k += 256) {
        float *q = A + (1 + k) * LD + (1 + i);
        if (k > i) *q = -ti_ * *q;
        else if (k == i) *q = 1.f - ti_;
        else *q = 0.f;
      }
      __syncthreads();
    }
  }

  // ---- stage 3: implicit QL, no barriers (private d/e per wave, private Z row per lane).
  float *zrow = (wantz && tid < n) ? A + tid * LD : nullptr;
  const int nfail = ql_implicit(d, e, n, zrow);
  __syncthreads();

  // ---- sort ascending (stable rank), undo scaling, write out.
  const float *d0 = dw;  // wave 0's copy
  for (int t = tid; t < n; t += 256) {
    const float dt = d0[t];
    const unsigned kt = sort_key(dt);
    int rk = 0;
    for (int k = 0; k < n; ++k) {
      const unsigned kk = sort_key(d0[k]);
      rk += (kk < kt || (kk == kt && k < t)) ? 1 : 0;
    }
    rank[t] = rk;
    wout[rk] = dt / sigma;
  }
  __syncthreads();
  if (wantz) {
    for (int idx = tid; idx < n * n; idx += 256) {
      const int i = idx / n, c = idx - i * n;
      Zg[(int64_t)i * ldz + rank[c]] = A[i * LD + c];
    }
  }
  if (tid == 0) *info = nfail;
}

size_t symeig_small_lds_bytes(int n) {
  const int LD = n | 1;
  return (size_t)(n * LD + 4 * n + 4 * n + 3 * n + 4 + n) * sizeof(float);
}

int symeig_small_launch(const float *A, int64_t lda, int n, float *w, float *Z, int64_t ldz, int32_t *info,
                        hipStream_t stream) {
  static unsigned long long attr_done = 0;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return VIVIT_E_LAUNCH;
    if (!(attr_done & (1ull << (dev & 63)))) {
      if (!ensure_dynamic_lds(reinterpret_cast<const void *>(symeig_small_kernel), (int)symeig_small_lds_bytes(SMALL_N_MAX),
                              attr_done))
        return VIVIT_E_LAUNCH;
      attr_done |= 1ull << (dev & 63);
    }
  }
  symeig_small_kernel<<<1, 256, symeig_small_lds_bytes(n), stream>>>(A, lda, n, w, Z, ldz, info);
  return launch_status();
}

} // namespace vivit
