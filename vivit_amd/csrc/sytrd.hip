// Blocked Householder tridiagonalisation  A = Q T Q^T  for n above the single-workgroup size.
//
// Right-looking panels of PB columns (LAPACK ssytrd/slatrd ordering, lower triangle).  Per
// column j the chip-wide work is one symmetric matrix-vector product with the trailing matrix
// -- the HBM-bound heart of the eigensolver: (2/3) n^3 bytes over the whole reduction because
// only the lower triangle is streamed, each 128x256 tile being read once and used for both
// y_row += A v_col and y_col += A^T v_row -- plus O(n * PB) panel corrections; after a panel the
// trailing matrix receives one rank-2*PB update on MFMA (gemm_lower_launch).
//
// Launch structure: 5 small launches per column on one stream (column update + norm partials,
// reflector, symv tiles + panel dots, dot reduce, finish).  A dependent kernel boundary costs
// ~1.5-2 us on MI355X, cheaper than an in-kernel grid barrier (4-10 us), so the column loop is a
// stream of plain launches rather than a persistent cooperative kernel.  Cross-workgroup
// reductions go through partial slabs with a fixed summation order: no float atomics, results
// are bit-reproducible.
//
// Storage: A is n x n row-major; only A[i][j], i >= j is read.  Reflector j (v_j, v_j[j+1] = 1)
// is written to the dead upper-triangle row j: A[j][j+1 .. n-1]; tau, d, e go to vectors.
#include "common.h"
#include "device_utils.h"
#include "eig_internal.h"

namespace vivit {

constexpr int PB = 64;          // panel width
constexpr int TR = 128;         // symv wave tile: rows
constexpr int TC = 256;         // symv wave tile: cols (64 lanes x float4)


// ------------------------------------------------------------------------------------------
// Scan: max |a_ij| over the lower triangle and a non-finite flag (for LAPACK-style scaling).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trd_scan_kernel(const float *__restrict__ A, int64_t lda, int n,
                                                       float *__restrict__ part) {
  __shared__ float red[4];
  const int i = blockIdx.x;  // one row per block
  float amax = 0.f, bad = 0.f;
  for (int c = threadIdx.x; c <= i; c += 256) {
    const float v = A[(int64_t)i * lda + c];
    amax = fmaxf(amax, fabsf(v));
    if (!(fabsf(v) <= 3.0e38f)) bad = 1.f;
  }
  amax = block_max(amax, red, threadIdx.x);
  bad = block_max(bad, red, threadIdx.x);
  if (threadIdx.x == 0) { part[2 * i] = amax; part[2 * i + 1] = bad; }
}

__global__ __launch_bounds__(256) void trd_sigma_kernel(const float *__restrict__ part, int n, float *__restrict__ scal) {
  __shared__ float red[4];
  float amax = 0.f, bad = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { amax = fmaxf(amax, part[2 * i]); bad = fmaxf(bad, part[2 * i + 1]); }
  amax = block_max(amax, red, threadIdx.x);
  bad = block_max(bad, red, threadIdx.x);
  if (threadIdx.x == 0) {
    const float rmin = 4.4408921e-16f, rmax = 2.2517998e15f;
    float sigma = 1.f;
    if (amax > 0.f && amax < rmin) sigma = rmin / amax;
    else if (amax > rmax) sigma = rmax / amax;
    scal[1] = sigma; scal[2] = bad; scal[3] = amax;
  }
}

__global__ __launch_bounds__(256) void trd_scale_kernel(float *__restrict__ A, int64_t lda, int n,
                                                        const float *__restrict__ scal) {
  const float sigma = scal[1];
  if (sigma == 1.f) return;
  const int i = blockIdx.x;
  for (int c = threadIdx.x; c <= i; c += 256) A[(int64_t)i * lda + c] *= sigma;
}

// ------------------------------------------------------------------------------------------
// K1: finalise W_{jj-1} (w = w' - 1/2 tau (w'.v) v) and form column j with the pending panel
//     update applied:  x[i] = A[i][j] - sum_{t<jj} V_t[i] W_t[j] + W_t[i] V_t[j],   i >= j.
//     Also emits the partial sums of squares of x[j+2:].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trd_col_kernel(const float *__restrict__ A, int64_t lda, int n, int j, int jj,
                                                      int finalize_only, SytrdWs ws, int nprev) {
  __shared__ float red[4];
  __shared__ float s_w[PB], s_v[PB];
  const int tid = threadIdx.x;
  const int64_t i = (int64_t)j + (int64_t)blockIdx.x * 256 + tid;
  const int64_t N = n;
  float *V = ws.vw, *W = ws.vw + (int64_t)PB * N;
  float a2 = 0.f;
  if (jj > 0) {
    float s = 0.f;
    for (int t = tid; t < nprev; t += 256) s += ws.wdotpart[t];
    s = block_sum(s, red, tid);
    a2 = -0.5f * ws.tau[j - 1] * s;
  }
  float wfin = 0.f;
  if (jj > 0 && i < N) {
    const int64_t o = (int64_t)(jj - 1) * N + i;
    wfin = W[o] + a2 * V[o];
    // Element j of this row is read by every block below (s_w[jj-1]) and never needed again
    // afterwards: leave it un-finalised instead of racing with those reads.
    if (finalize_only || i != j) W[o] = wfin;
  }
  if (finalize_only) return;
  if (tid < jj) {
    const int64_t o = (int64_t)tid * N + j;
    float wj = W[o];
    if (tid == jj - 1) wj = wj + a2 * V[o];  // row j of W_{jj-1} may not be written back yet
    s_w[tid] = wj;
    s_v[tid] = V[o];
  }
  __syncthreads();
  float x = 0.f;
  if (i < N) {
    x = A[i * lda + j];
    for (int t = 0; t < jj; ++t) {
      const float wt = (t == jj - 1) ? wfin : W[(int64_t)t * N + i];
      x -= V[(int64_t)t * N + i] * s_w[t] + wt * s_v[t];
    }
    ws.xbuf[i] = x;
    if (i == j) ws.d[j] = x;
    if (i == j + 1) ws.scal[0] = x;
  }
  const float sq = (i < N && i >= j + 2) ? x * x : 0.f;
  const float ss = block_sum(sq, red, tid);
  if (tid == 0) ws.ssqpart[blockIdx.x] = ss;
}

// ------------------------------------------------------------------------------------------
// K1b: Householder reflector from x: beta, tau, v (v[j+1] = 1).  Rows i >= j+1.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trd_reflector_kernel(float *__restrict__ A, int64_t lda, int n, int j, int jj,
                                                            SytrdWs ws, int npart) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float s = 0.f;
  for (int t = tid; t < npart; t += 256) s += ws.ssqpart[t];
  const float ssq = block_sum(s, red, tid);
  const float alpha = ws.scal[0];
  float tauj = 0.f, beta = alpha, sc = 0.f;
  if (ssq > 0.f) {
    beta = -copysignf(sqrtf(alpha * alpha + ssq), alpha);
    tauj = (beta - alpha) / beta;
    sc = 1.f / (alpha - beta);
  }
  const int64_t N = n;
  const int64_t i = (int64_t)j + 1 + (int64_t)blockIdx.x * 256 + tid;
  if (i < N) {
    const float v = (i == j + 1) ? 1.f : ws.xbuf[i] * sc;
    ws.vw[(int64_t)jj * N + i] = v;
    ws.vw[(int64_t)(2 * PB + jj) * N + i] = v;
    A[(int64_t)j * lda + i] = v;  // reflector storage: dead upper-triangle row j
  }
  if (blockIdx.x == 0 && tid == 0) { ws.e[j] = beta; ws.tau[j] = tauj; }
}

// ------------------------------------------------------------------------------------------
// K2: symmetric matrix-vector product over the lower triangle + panel dot products.
//     One wavefront per 128 x 256 tile; lanes span 4 columns each (16-byte loads), rows are
//     walked 8 at a time: row sums need a cross-lane reduction (recursive halving, 10 shuffles
//     per 8 rows), column sums accumulate in registers.  Every tile writes its 128 row partials
//     and 256 column partials to slabs; K3 adds them in a fixed order.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4_guard(const float *__restrict__ p, int64_t c, int64_t n, bool vec) {
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec && c + 3 < n) {
    a = *reinterpret_cast<const float4 *>(p + c);
  } else {
    if (c < n) a.x = p[c];
    if (c + 1 < n) a.y = p[c + 1];
    if (c + 2 < n) a.z = p[c + 2];
    if (c + 3 < n) a.w = p[c + 3];
  }
  return a;
}

// Reduce 8 per-lane values over the 64 lanes; on return every lane holds the total of row
// rho(lane) = 4*bit5 + 2*bit4 + bit3 of its lane index.
__device__ __forceinline__ float reduce8(const float (&v)[8], int lane) {
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
  float a[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float send = b5 ? v[k] : v[k + 4];
    const float keep = b5 ? v[k + 4] : v[k];
    a[k] = keep + __shfl_xor(send, 32, 64);
  }
  float b[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float send = b4 ? a[k] : a[k + 2];
    const float keep = b4 ? a[k + 2] : a[k];
    b[k] = keep + __shfl_xor(send, 16, 64);
  }
  float c;
  {
    const float send = b3 ? b[0] : b[1];
    const float keep = b3 ? b[1] : b[0];
    c = keep + __shfl_xor(send, 8, 64);
  }
  c += __shfl_xor(c, 4, 64);
  c += __shfl_xor(c, 2, 64);
  c += __shfl_xor(c, 1, 64);
  return c;
}

template <bool VEC>
__global__ __launch_bounds__(256) void trd_symv_kernel(const float *__restrict__ A, int64_t lda, int n, int j, int jj,
                                                       SytrdWs ws, int rt0, int nrt, int ct0, int nct) {
  __shared__ float rowsum[4][TR];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t N = n;
  const int ntile = nrt * nct;
  const int widx = blockIdx.x * 4 + wave;
  const float *__restrict__ v = ws.vw + (int64_t)jj * N;

  if (widx >= ntile) {
    // ---- panel-dot wave: chunk of 256 indices, all 2*jj panel rows.
    const int chunk = widx - ntile;
    if (chunk >= nct || jj == 0) return;
    const int64_t c = (int64_t)(ct0 + chunk) * TC + 4 * lane;
    const float4 vc = ld4_guard(v, c, N, VEC);
    float *out = ws.dotpart + (int64_t)chunk * (2 * PB);
    for (int t0 = 0; t0 < jj; t0 += 8) {
      float pv[8], pw[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        pv[u] = 0.f; pw[u] = 0.f;
        const int t = t0 + u;
        if (t < jj) {
          const float4 a = ld4_guard(ws.vw + (int64_t)t * N, c, N, VEC);
          const float4 b = ld4_guard(ws.vw + (int64_t)(PB + t) * N, c, N, VEC);
          pv[u] = a.x * vc.x + a.y * vc.y + a.z * vc.z + a.w * vc.w;
          pw[u] = b.x * vc.x + b.y * vc.y + b.z * vc.z + b.w * vc.w;
        }
      }
      const float rv = reduce8(pv, lane);
      const float rw = reduce8(pw, lane);
      if ((lane & 7) == 0) {
        const int rho = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
        if (t0 + rho < jj) {
          out[t0 + rho] = rv;
          out[PB + t0 + rho] = rw;
        }
      }
    }
    return;
  }

  // ---- symv tile
  const int rt = rt0 + widx / nct, ct = ct0 + widx % nct;
  const int64_t R0 = (int64_t)rt * TR, C0 = (int64_t)ct * TC;
  if (C0 > R0 + TR - 1) return;  // entirely above the diagonal: no partials are ever read from it
  const int64_t c = C0 + 4 * lane;
  const float4 vc = ld4_guard(v, c, N, VEC);           // v[c..c+3]  (zero below j+1 and beyond n)
  float vr0 = 0.f, vr1 = 0.f;                          // v[R0 + lane], v[R0 + 64 + lane]
  if (R0 + lane < N) vr0 = v[R0 + lane];
  if (R0 + 64 + lane < N) vr1 = v[R0 + 64 + lane];
  float4 colacc = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool interior = (C0 + TC - 1 < R0) && (R0 + TR - 1 < N);  // strictly below the diagonal, in bounds

  for (int g = 0; g < TR / 8; ++g) {
    float racc[8];
    float4 a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = R0 + g * 8 + u;
      a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (interior && VEC) {
        a[u] = *reinterpret_cast<const float4 *>(A + i * lda + c);
      } else if (i < N) {
        a[u] = ld4_guard(A + i * lda, c, N, VEC);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = g * 8 + u;
      const int64_t i = R0 + r;
      const float vi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r < 64 ? vr0 : vr1), r & 63));
      float4 x = a[u];
      if (!interior) {
        // triangle mask: keep c+e <= i for the row product, c+e < i for the column product
        if (c + 0 > i) x.x = 0.f;
        if (c + 1 > i) x.y = 0.f;
        if (c + 2 > i) x.z = 0.f;
        if (c + 3 > i) x.w = 0.f;
      }
      racc[u] = x.x * vc.x + x.y * vc.y + x.z * vc.z + x.w * vc.w;
      if (!interior) {
        if (c + 0 == i) x.x = 0.f;
        if (c + 1 == i) x.y = 0.f;
        if (c + 2 == i) x.z = 0.f;
        if (c + 3 == i) x.w = 0.f;
      }
      colacc.x += x.x * vi; colacc.y += x.y * vi; colacc.z += x.z * vi; colacc.w += x.w * vi;
    }
    const float tot = reduce8(racc, lane);
    if ((lane & 7) == 0) {
      const int rho = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
      rowsum[wave][g * 8 + rho] = tot;
    }
  }
  // column partials: one coalesced 1 KiB store per wave
  float *cp = ws.colpart + (int64_t)(rt - rt0) * N;
  if (c + 3 < N && VEC) {
    *reinterpret_cast<float4 *>(cp + c) = colacc;
  } else {
    if (c < N) cp[c] = colacc.x;
    if (c + 1 < N) cp[c + 1] = colacc.y;
    if (c + 2 < N) cp[c + 2] = colacc.z;
    if (c + 3 < N) cp[c + 3] = colacc.w;
  }
  // row partials (same wave wrote rowsum: LDS ops of one wave are ordered)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  float *rp = ws.rowpart + (int64_t)(ct - ct0) * N;
  if (R0 + lane < N) rp[R0 + lane] = rowsum[wave][lane];
  if (R0 + 64 + lane < N) rp[R0 + 64 + lane] = rowsum[wave][64 + lane];
}

// K2b: cvw[t] = sum over chunks of dotpart[chunk][t]
__global__ __launch_bounds__(2 * PB) void trd_dotreduce_kernel(SytrdWs ws, int nchunk, int jj) {
  const int t = threadIdx.x;
  const int tt = t < PB ? t : t - PB;
  float s = 0.f;
  if (tt < jj)
    for (int c = 0; c < nchunk; ++c) s += ws.dotpart[(int64_t)c * (2 * PB) + t];
  ws.cvw[t] = s;
}

// ------------------------------------------------------------------------------------------
// K3: y = sum of partials;  w' = tau (y - V c_w - W c_v);  partial w'.v.   Rows i >= j+1.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trd_finish_kernel(int n, int j, int jj, SytrdWs ws, int rt0, int nrt, int ct0) {
  __shared__ float red[4];
  __shared__ float s_c[2 * PB];
  const int tid = threadIdx.x;
  if (tid < 2 * PB) s_c[tid] = ws.cvw[tid];
  __syncthreads();
  const int64_t N = n;
  const int64_t i = (int64_t)j + 1 + (int64_t)blockIdx.x * 256 + tid;
  float wv = 0.f;
  if (i < N) {
    float y = 0.f;
    const int ctl = (int)(i / TC);
    for (int ct = ct0; ct <= ctl; ++ct) y += ws.rowpart[(int64_t)(ct - ct0) * N + i];
    int rtf = (int)(i / TR);
    if (rtf < rt0) rtf = rt0;
    for (int rt = rtf; rt < rt0 + nrt; ++rt) y += ws.colpart[(int64_t)(rt - rt0) * N + i];
    const float *V = ws.vw, *W = ws.vw + (int64_t)PB * N;
    for (int t = 0; t < jj; ++t) y -= V[(int64_t)t * N + i] * s_c[PB + t] + W[(int64_t)t * N + i] * s_c[t];
    const float w = ws.tau[j] * y;
    ws.vw[(int64_t)(PB + jj) * N + i] = w;
    wv = w * V[(int64_t)jj * N + i];
  }
  const float s = block_sum(wv, red, tid);
  if (tid == 0) ws.wdotpart[blockIdx.x] = s;
}

// Tail: d[n-2], e[n-2], d[n-1] from the last 2x2 block with the pending panel applied.
__global__ void trd_tail_kernel(const float *__restrict__ A, int64_t lda, int n, int jjcount, SytrdWs ws) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int64_t N = n;
  const float *V = ws.vw, *W = ws.vw + (int64_t)PB * N;
  const int64_t rows[3] = {N - 2, N - 1, N - 1}, cols[3] = {N - 2, N - 2, N - 1};
  float out[3];
  for (int q = 0; q < 3; ++q) {
    const int64_t i = rows[q], k = cols[q];
    float x = A[i * lda + k];
    for (int t = 0; t < jjcount; ++t)
      x -= V[(int64_t)t * N + i] * W[(int64_t)t * N + k] + W[(int64_t)t * N + i] * V[(int64_t)t * N + k];
    out[q] = x;
  }
  ws.d[n - 2] = out[0];
  ws.e[n - 2] = out[1];
  ws.d[n - 1] = out[2];
  ws.e[n - 1] = 0.f;
  ws.tau[n - 2] = 0.f;
  ws.tau[n - 1] = 0.f;
}

// ------------------------------------------------------------------------------------------
size_t sytrd_workspace_floats(int64_t n) {
  const int64_t nct = cdiv(n, TC), nrt = cdiv(n, TR), nwg = cdiv(n, 256) + 1;
  int64_t f = 0;
  f += 3 * PB * n;                 // vw
  f += n;                          // xbuf
  f += nct * n + nrt * n;          // rowpart, colpart
  f += nct * 2 * PB + 2 * PB;      // dotpart, cvw
  f += 2 * nwg + 16;               // ssqpart, wdotpart, scal
  f += 3 * n;                      // d, e, tau
  f += 2 * n;                      // scan partials
  return (size_t)(f + 64) / 4 * 4 + 64;
}

// Carve the workspace (all sub-buffers 16-byte aligned).
static SytrdWs sytrd_carve(float *base, int64_t n, float **scanpart) {
  const int64_t nct = cdiv(n, TC), nrt = cdiv(n, TR), nwg = cdiv(n, 256) + 1;
  auto take = [&](int64_t count) {
    float *p = base;
    base += (count + 3) / 4 * 4;
    return p;
  };
  SytrdWs ws;
  ws.vw = take(3 * PB * n);
  ws.xbuf = take(n);
  ws.rowpart = take(nct * n);
  ws.colpart = take(nrt * n);
  ws.dotpart = take(nct * 2 * PB);
  ws.cvw = take(2 * PB);
  ws.ssqpart = take(nwg);
  ws.wdotpart = take(nwg);
  ws.scal = take(16);
  ws.d = take(n);
  ws.e = take(n);
  ws.tau = take(n);
  *scanpart = take(2 * n);
  return ws;
}

// Tridiagonalise A (n x n, lda).  On return ws.d / ws.e / ws.tau hold T and the reflector
// scalars, A's upper-triangle rows hold the reflectors, ws.scal[1] the applied scaling sigma
// and ws.scal[2] the non-finite-input flag.
int sytrd_launch(float *A, int64_t n, int64_t lda, float *wsbase, SytrdWs *out, hipStream_t stream) {
  float *scanpart;
  SytrdWs ws = sytrd_carve(wsbase, n, &scanpart);
  *out = ws;
  const int ni = (int)n;
  const bool vec = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (lda % 4 == 0) && (n % 4 == 0);

  trd_scan_kernel<<<ni, 256, 0, stream>>>(A, lda, ni, scanpart);
  trd_sigma_kernel<<<1, 256, 0, stream>>>(scanpart, ni, ws.scal);
  trd_scale_kernel<<<ni, 256, 0, stream>>>(A, lda, ni, ws.scal);

  for (int64_t j0 = 0; j0 < n - 2; j0 += PB) {
    const int bb = (int)((n - 2 - j0) < PB ? (n - 2 - j0) : PB);
    if (hipMemsetAsync(ws.vw, 0, sizeof(float) * 3 * PB * n, stream) != hipSuccess) return VIVIT_E_LAUNCH;
    for (int jj = 0; jj < bb; ++jj) {
      const int j = (int)(j0 + jj);
      const int g1 = (int)cdiv(n - j, 256);      // rows i >= j
      const int g2 = (int)cdiv(n - j - 1, 256);  // rows i >= j+1
      const int nprev = (int)cdiv(n - j, 256);   // finish-kernel grid of column j-1 (rows >= j)
      trd_col_kernel<<<g1, 256, 0, stream>>>(A, lda, ni, j, jj, 0, ws, nprev);
      trd_reflector_kernel<<<g2, 256, 0, stream>>>(A, lda, ni, j, jj, ws, g1);
      const int rt0 = (j + 1) / TR, nrt = (int)cdiv(n, TR) - rt0;
      const int ct0 = (j + 1) / TC, nct = (int)cdiv(n, TC) - ct0;
      const int nwave = nrt * nct + (jj > 0 ? nct : 0);
      if (vec)
        trd_symv_kernel<true><<<(unsigned)cdiv(nwave, 4), 256, 0, stream>>>(A, lda, ni, j, jj, ws, rt0, nrt, ct0, nct);
      else
        trd_symv_kernel<false><<<(unsigned)cdiv(nwave, 4), 256, 0, stream>>>(A, lda, ni, j, jj, ws, rt0, nrt, ct0, nct);
      trd_dotreduce_kernel<<<1, 2 * PB, 0, stream>>>(ws, nct, jj);
      trd_finish_kernel<<<g2, 256, 0, stream>>>(ni, j, jj, ws, rt0, nrt, ct0);
    }
    // finalise the last W of the panel (rows >= j0 + bb)
    {
      const int j = (int)(j0 + bb);
      const int g1 = (int)cdiv(n - j, 256);
      trd_col_kernel<<<g1, 256, 0, stream>>>(A, lda, ni, j, bb, 1, ws, g1);
    }
    const int64_t off = j0 + bb;
    if (off < n - 2) {
      // trailing update A[off:, off:] -= V W^T + W V^T  (lower tiles, MFMA, K = 2*PB)
      const int st = gemm_lower_launch(ws.vw + off, ws.vw + (int64_t)PB * n + off, A + off * lda + off, n - off, 2 * PB,
                                       n, n, lda, -1.f, 1.f, stream);
      if (st != VIVIT_OK) return st;
    } else {
      trd_tail_kernel<<<1, 64, 0, stream>>>(A, lda, ni, bb, ws);
    }
  }
  return launch_status();
}

} // namespace vivit
