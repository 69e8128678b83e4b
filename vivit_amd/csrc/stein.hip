// Selected eigenvectors of a symmetric tridiagonal matrix by inverse iteration (the "K vectors after the criterion
// callback" stage of vivit_symeig_select_f32; reference call sites: evecs[:, keep] at vivit/linalg/eigh.py:248-253,
// vivit/optim/directional_damped_newton.py:315-321).
//
// All arithmetic is fp64 on the fp32 tridiagonal (d, e): the eigenvalues come from the Sturm multisection with a
// bracket of 3.5e-12 of the spectral span, so one inverse-iteration step amplifies the wanted eigenvector by ~1e12
// against its neighbours unless they are closer than ~1e-9 of the span; such numerically multiple eigenvalues
// (any orthonormal basis of the eigenspace is a correct answer) are orthogonalised afterwards by modified Gram-Schmidt
// in ascending order.  One lane per eigenvector; the per-vector LU factors of T - lambda I (partial pivoting, as LAPACK's
// dlagtf/dlagts) live in global memory interleaved over the vectors ([i][Kp]: lanes read/write consecutive addresses).
#include "common.h"
#include "eig_internal.h"

namespace vivit {

constexpr int ST_ITERS = 4;          // inverse-iteration steps (2 would do for isolated eigenvalues)
constexpr double ST_TIGHT = 1e-9;    // eigenvalues closer than this fraction of the span are orthogonalised explicitly

struct SteinWs {
  double *a, *b, *c, *d2, *y;  // [n][Kp]
  unsigned char *piv;          // [n][Kp]
  double *span;                // [2]: Gershgorin span, norm bound
};

static inline int64_t stein_kp(int64_t K) { return (K + 63) / 64 * 64; }

size_t stein_workspace_bytes(int64_t n, int64_t K) {
  const int64_t Kp = stein_kp(K);
  return (size_t)align_up(sizeof(double) * n * Kp, 256) * 5 + align_up((size_t)n * Kp, 256) + 256 + 512;
}

static SteinWs stein_carve(void *base, int64_t n, int64_t K) {
  const int64_t Kp = stein_kp(K);
  char *p = reinterpret_cast<char *>(align_up(reinterpret_cast<uintptr_t>(base), 256));
  auto take = [&](size_t bytes) {
    char *r = p;
    p += align_up(bytes, 256);
    return r;
  };
  SteinWs ws;
  ws.a = (double *)take(sizeof(double) * n * Kp);
  ws.b = (double *)take(sizeof(double) * n * Kp);
  ws.c = (double *)take(sizeof(double) * n * Kp);
  ws.d2 = (double *)take(sizeof(double) * n * Kp);
  ws.y = (double *)take(sizeof(double) * n * Kp);
  ws.piv = (unsigned char *)take((size_t)n * Kp);
  ws.span = (double *)take(256);
  return ws;
}

// span[0] = Gershgorin span of T, span[1] = max row sum (norm bound); one workgroup
__global__ __launch_bounds__(256) void stein_span_kernel(const float *__restrict__ d, const float *__restrict__ e, int n,
                                                         double *__restrict__ span) {
  __shared__ double slo[256], shi[256], snr[256];
  double lo = 1e300, hi = -1e300, nr = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double r = (i > 0 ? fabs((double)e[i - 1]) : 0.0) + (i + 1 < n ? fabs((double)e[i]) : 0.0);
    lo = fmin(lo, (double)d[i] - r);
    hi = fmax(hi, (double)d[i] + r);
    nr = fmax(nr, fabs((double)d[i]) + r);
  }
  slo[threadIdx.x] = lo; shi[threadIdx.x] = hi; snr[threadIdx.x] = nr;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      slo[threadIdx.x] = fmin(slo[threadIdx.x], slo[threadIdx.x + s]);
      shi[threadIdx.x] = fmax(shi[threadIdx.x], shi[threadIdx.x + s]);
      snr[threadIdx.x] = fmax(snr[threadIdx.x], snr[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    span[0] = fmax(shi[0] - slo[0], 1e-300);
    span[1] = fmax(snr[0], 1e-300);
  }
}

// lane k: LU of T - lam_k I, then ST_ITERS solves from a pseudo-random start; y[:, k] is left with unit 2-norm.
// `sel[k]` is the position of the wanted eigenvalue in the ascending list lam64 (strictly ascending in k).
__global__ __launch_bounds__(64) void stein_iterate_kernel(const float *__restrict__ d, const float *__restrict__ e, int n,
                                                           const double *__restrict__ lam64, const int *__restrict__ sel,
                                                           int K, int Kp, SteinWs ws, int32_t *__restrict__ info) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= K) return;
  const double tnorm = ws.span[1];
  const double tol = 2.3e-16 * tnorm;       // smallest pivot magnitude accepted in the back substitution
  double lam = lam64[sel[k]];
  // separate numerically coincident selected eigenvalues a little (as LAPACK's sstein does), so that their start
  // vectors do not produce the same iterate: the j-th member of a run of equal values is moved by j * 10 eps |lam|
  {
    int run = 0;
    for (int j = k - 1; j >= 0 && lam64[sel[j]] >= lam - 1e-15 * tnorm; --j) ++run;
    lam += (double)run * 2.3e-15 * fmax(fabs(lam), 1e-3 * tnorm);
  }
  double *a = ws.a + k, *b = ws.b + k, *c = ws.c + k, *d2 = ws.d2 + k, *y = ws.y + k;
  unsigned char *piv = ws.piv + k;
  // ---- factorisation  T - lam I = P L U   (U: diag a, superdiag b, second superdiag d2; L multipliers c)
  double ak = (double)d[0] - lam;           // running diagonal entry of row k
  double bk = n > 1 ? (double)e[0] : 0.0;   // running superdiagonal entry of row k
  for (int i = 0; i < n - 1; ++i) {
    const double ci = (double)e[i];                                  // subdiagonal entry (row i+1, col i)
    const double anext = (double)d[i + 1] - lam;                     // diagonal of row i+1
    const double bnext = (i + 2 < n) ? (double)e[i + 1] : 0.0;       // superdiagonal of row i+1
    const int64_t o = (int64_t)i * Kp;
    if (fabs(ci) <= fabs(ak)) {  // no interchange
      const double mult = ak != 0.0 ? ci / ak : 0.0;
      a[o] = ak; b[o] = bk; c[o] = mult; d2[o] = 0.0; piv[o] = 0;
      ak = anext - mult * bk;
      bk = bnext;
    } else {                     // rows i and i+1 swap
      const double mult = ak / ci;
      a[o] = ci; b[o] = anext; c[o] = mult; d2[o] = bnext; piv[o] = 1;
      ak = bk - mult * anext;
      bk = -mult * bnext;
    }
  }
  {
    const int64_t o = (int64_t)(n - 1) * Kp;
    a[o] = ak; b[o] = 0.0; c[o] = 0.0; d2[o] = 0.0; piv[o] = 0;
  }
  // ---- start vector: xorshift per (k, i), in [-1, 1)
  unsigned int s = (0x9E3779B9u * (unsigned int)(sel[k] + 1) + 0x85EBCA6Bu) | 1u;   // never 0: xorshift would stay 0
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    y[(int64_t)i * Kp] = (double)(int)s * (1.0 / 2147483648.0);
  }
  double growth = 0.0;   // |x| / |y| of the last solve
  for (int it = 0; it < ST_ITERS; ++it) {
    // forward: y <- L^-1 P y
    double yi = y[0];
    for (int i = 0; i < n - 1; ++i) {
      const int64_t o = (int64_t)i * Kp;
      const double ynext = y[o + Kp];
      const double m = c[o];
      if (piv[o] == 0) {
        y[o] = yi;
        yi = ynext - m * yi;
      } else {
        y[o] = ynext;
        yi = yi - m * ynext;
      }
    }
    y[(int64_t)(n - 1) * Kp] = yi;
    // backward: U x = y; accumulate the squared norm with a running scale to stay in range
    double x1 = 0.0, x2 = 0.0, ss = 0.0, scale = 1.0;
    for (int i = n - 1; i >= 0; --i) {
      const int64_t o = (int64_t)i * Kp;
      double piv_a = a[o];
      if (fabs(piv_a) < tol) piv_a = piv_a < 0.0 ? -tol : tol;
      const double x = (y[o] * scale - b[o] * x1 - d2[o] * x2) / piv_a;
      y[o] = x;   // NOTE: entries written so far carry `scale`; rescaled below when the scale changes
      x2 = x1; x1 = x;
      ss += x * x;
      if (ss > 1e200) {  // rescale everything computed so far (rare: only for an (almost) exact eigenvalue)
        const double f = 1e-100;
        for (int j = i; j < n; ++j) y[(int64_t)j * Kp] *= f;
        x1 *= f; x2 *= f; ss *= f * f; scale *= f;
      }
    }
    const double inv = 1.0 / sqrt(fmax(ss, 1e-300));
    for (int i = 0; i < n; ++i) y[(int64_t)i * Kp] *= inv;
    growth = sqrt(ss) / scale;
  }
  // A converged inverse iteration amplifies a unit vector by ~ 1 / |lam - lam_true| >= 1 / (n eps64 |T|); a shift that is
  // not an eigenvalue (or a broken solve: zero / non-finite vector) shows as a small or non-finite growth factor.  The
  // caller's `info` word counts such vectors (kernels.check_info raises, like a non-converged Tensor.symeig).
  if (tnorm > 1e-290 && (!(growth * tnorm >= 1e6) || !(growth < INFINITY))) atomicAdd(info, 1);
}

// One workgroup: modified Gram-Schmidt inside runs of numerically multiple selected eigenvalues (ascending order),
// then Zt[k][i] = y[i][k] as fp32 (unit norm).  Vectors outside such runs are only converted.
__global__ __launch_bounds__(256) void stein_finish_kernel(int n, const double *__restrict__ lam64, const int *__restrict__ sel,
                                                           int K, int Kp, SteinWs ws, float *__restrict__ Zt, int64_t ldz) {
  __shared__ double red[256];
  const int tid = threadIdx.x;
  const double tight = ST_TIGHT * ws.span[0];
  for (int k = 0; k < K; ++k) {
    double *yk = ws.y + k;
    bool touched = false;
    for (int j = k - 1; j >= 0 && lam64[sel[k]] - lam64[sel[j]] <= tight; --j) {
      if (j < k - 1 && lam64[sel[j + 1]] - lam64[sel[j]] > tight) break;  // chain of neighbours must be unbroken
      const double *yj = ws.y + j;
      double acc = 0.0;
      for (int i = tid; i < n; i += 256) acc += yk[(int64_t)i * Kp] * yj[(int64_t)i * Kp];
      red[tid] = acc;
      __syncthreads();
      for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (tid < s2) red[tid] += red[tid + s2];
        __syncthreads();
      }
      const double dot = red[0];
      __syncthreads();
      for (int i = tid; i < n; i += 256) yk[(int64_t)i * Kp] -= dot * yj[(int64_t)i * Kp];
      __syncthreads();
      touched = true;
    }
    double nrm = 1.0;
    if (touched) {
      double acc = 0.0;
      for (int i = tid; i < n; i += 256) { const double v = yk[(int64_t)i * Kp]; acc += v * v; }
      red[tid] = acc;
      __syncthreads();
      for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (tid < s2) red[tid] += red[tid + s2];
        __syncthreads();
      }
      nrm = 1.0 / sqrt(fmax(red[0], 1e-300));
      __syncthreads();
      for (int i = tid; i < n; i += 256) yk[(int64_t)i * Kp] *= nrm;
      __syncthreads();
    }
    for (int i = tid; i < n; i += 256) Zt[(int64_t)k * ldz + i] = (float)yk[(int64_t)i * Kp];
  }
}

// Zt[k][:] (k < K, ld ldz) = unit eigenvector of the tridiagonal (d, e) for the eigenvalue lam64[sel[k]].
// sel: device int32 [K], strictly ascending positions in the ascending eigenvalue list lam64 [n] (fp64).
int stein_launch(const float *d, const float *e, int64_t n, const double *lam64, const int *sel, int64_t K, float *Zt,
                 int64_t ldz, void *wsbase, int32_t *info, hipStream_t stream) {
  if (K <= 0) return VIVIT_OK;
  SteinWs ws = stein_carve(wsbase, n, K);
  const int Kp = (int)stein_kp(K);
  stein_span_kernel<<<1, 256, 0, stream>>>(d, e, (int)n, ws.span);
  stein_iterate_kernel<<<(unsigned)cdiv(K, 64), 64, 0, stream>>>(d, e, (int)n, lam64, sel, (int)K, Kp, ws, info);
  stein_finish_kernel<<<1, 256, 0, stream>>>((int)n, lam64, sel, (int)K, Kp, ws, Zt, ldz);
  return launch_status();
}

} // namespace vivit
