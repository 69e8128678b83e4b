// Multi-kernel symmetric eigensolver for n > SMALL_N_MAX:
//   sytrd.hip  blocked Householder tridiagonalisation          (HBM-bound symv + MFMA rank-2k)
//   stedc.hip  tridiagonal eigenproblem                         (bisection | divide & conquer)
//   here       Householder back-transformation  Z = Q_H Q_T     (compact-WY blocks, MFMA GEMMs)
//
// Eigenvectors are carried transposed (row = eigenvector) until the very end: the D&C produces
// Qt = Q_T^T, the back-transformation applies the block reflectors from the right,
//   Zt <- Zt (I - Y T Y^T)^T = Zt - ((Zt Y) T^T) Y^T,
// three GEMMs per block of KB reflectors whose rows Y^T are exactly the rows sytrd left in the
// upper triangle of A, and a final permuted transpose delivers Tensor.symeig's column layout.
#include <cstdlib>

#include "common.h"
#include "device_utils.h"
#include "eig_internal.h"

namespace vivit {

constexpr int KB = 128;  // reflectors per compact-WY block

// Yt[t][i] = v_{a+t}[i] (zero for i <= a+t and for reflector indices beyond n-3)
// (one-stage: shift = 1, jmax = n-3; two-stage stage-1 reflectors: shift = NB, jmax = n-NB-1)
__global__ __launch_bounds__(256) void bt_extract_kernel(const float *__restrict__ A, int64_t lda, int n, int a,
                                                         float *__restrict__ Yt, int shift, int jmax) {
  const int t = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int j = a + t;
  float v = 0.f;
  if (j <= jmax && i >= j + shift) v = A[(int64_t)j * lda + i];
  Yt[(int64_t)t * n + i] = v;
}

// T (upper triangular, forward/columnwise larft) of ONE block of KB reflectors from its Gram block
// S (ld lds) and tau, written into the diagonal block of the super-block factor (ld ldt): one thread per
// column (device_utils.h:tfactor_column), S and T staged in LDS (2 x 66 KB, dynamic)
constexpr int BT_LD = KB + 1;
constexpr int BT_TF_LDS = (2 * KB * BT_LD + KB) * 4;
__global__ __launch_bounds__(KB) void bt_tfactor_kernel(const float *__restrict__ S, int64_t lds_, const float *__restrict__ tau,
                                                        int jmax, int a, float *__restrict__ T, int64_t ldt) {
  extern __shared__ float tf_lds[];
  float *Ss = tf_lds, *Ts = Ss + KB * BT_LD, *taus = Ts + KB * BT_LD;
  const int r = threadIdx.x;
  // workgroup b factors diagonal block b of the super-block: reflectors a + b KB .., S and T blocks at (b KB, b KB)
  a += blockIdx.x * KB;
  S += (int64_t)blockIdx.x * KB * (lds_ + 1);
  T += (int64_t)blockIdx.x * KB * (ldt + 1);
  for (int idx = r; idx < KB * KB; idx += KB) Ss[(idx / KB) * BT_LD + (idx % KB)] = S[(int64_t)(idx / KB) * lds_ + (idx % KB)];
  taus[r] = (a + r <= jmax) ? tau[a + r] : 0.f;
  __syncthreads();
  tfactor_column(Ss, taus, Ts, BT_LD, KB, r);
  __syncthreads();
  for (int idx = r; idx < KB * KB; idx += KB) T[(int64_t)(idx / KB) * ldt + (idx % KB)] = Ts[(idx / KB) * BT_LD + (idx % KB)];
}

// Blocks of KB reflectors are merged into super-blocks of `nsub` blocks (a power of two, up to 8: 1024
// reflectors) so that the three products of the back-transformation have a long contraction / wide
// output (the 256 x 256 tile kernel; Zt is streamed once per 1024 instead of once per 128 reflectors):
//   H_1 H_2 = I - [Y1 Y2] [[T1, -T1 (Y1^T Y2) T2], [0, T2]] [Y1 Y2]^T        (applied recursively).
// The merge tree T12 = -T1 (Y1^T Y2) T2 as TWO batched products per level (all pairs of a level are independent) instead of
// two launches per pair: at 2048 reflectors 8 launches instead of 30 per super-block (Q1 at n = 40 960: 600 launches of 24 us).
// The descriptors only depend on the buffers, so one tiny kernel writes them once per back-transformation:
// per level (halves of size h, nb = KS / 2h pairs):  desc[at + q] : X_q = T1 S12,   desc[at + nb + q] : T12 = -X_q T2,   at += 2 nb.
constexpr int BT_MAX_DESC = 2 * 32;   // 2 (nsub - 1) descriptors, nsub <= 32
__global__ void bt_merge_desc_kernel(const float *S, float *T, float *X, int64_t KS, int64_t kb, GemmDesc *desc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int at = 0;
  for (int64_t h = kb; h < KS; h *= 2) {
    const int nb = (int)(KS / (2 * h));
    for (int64_t o = 0, q = 0; o < KS; o += 2 * h, ++q) {
      float *Xq = X + q * h * h;
      GemmDesc d1, d2;
      d1.A = T + o * (KS + 1); d1.B = S + o * KS + (o + h); d1.C = Xq;
      d1.M = h; d1.N = h; d1.K = h; d1.lda = KS; d1.ldb = KS; d1.ldc = h;
      d2.A = Xq; d2.B = T + (o + h) * (KS + 1); d2.C = T + o * KS + (o + h);
      d2.M = h; d2.N = h; d2.K = h; d2.lda = h; d2.ldb = KS; d2.ldc = KS;
      desc[at + q] = d1; desc[at + nb + q] = d2;
    }
    at += 2 * nb;
  }
}

static int bt_nsub(int64_t n) {   // largest super-block (workspace sizing)
  static int forced = -2;
  if (forced == -2) { const char *e = getenv("VIVIT_BT_NSUB"); forced = e ? atoi(e) : -1; }
  if (forced > 0 && n >= 2048) return forced;   // (power of two; experiments)
  return n >= 16384 ? 16 : (n >= 8192 ? 8 : (n >= 4096 ? 4 : (n >= 2048 ? 2 : 1)));
}
// Super-block actually used for `nrows` rows of Zt: the read-modify-write of Zt per super-block favours 2048
// reflectors when (nearly) all rows are transformed (Q1 at n = 40 960: 1037 / 816 / 767 / 757 ms for 512 / 1024 /
// 2048 / 4096), while the T factor work (2 KS n^2 flop, independent of nrows) favours 1024 for a few selected rows.
static int bt_nsub_rows(int64_t n, int64_t nrows) {
  const int nmax = bt_nsub(n);
  return (nmax > 8 && nrows < 4096) ? 8 : nmax;
}

// split-K slab of the products for ANY nrows <= n (row-range mode): a one-tile-wide output gets up to
// 2048 / tiles splits, so splits * nrows <= 2048 * 128 and the slab is bounded by 2048 * 128 * KS floats
static size_t bt_gemm_ws_bytes(int64_t n) {
  const int64_t KS = (int64_t)KB * bt_nsub(n);
  size_t b = gemm_workspace_bytes(n, KS, n, false);
  const size_t b2 = gemm_workspace_bytes(KS, KS, n, false), b3 = gemm_workspace_bytes(KS, KS, n, true);
  const size_t bound = (size_t)2048 * 128 * KS * sizeof(float);
  if (b2 > b) b = b2;
  if (b3 > b) b = b3;
  return b > bound ? b : bound;
}

static size_t bt_workspace_bytes(int64_t n) {
  const int64_t KS = (int64_t)KB * bt_nsub(n);
  size_t b = 0;
  b += align_up(sizeof(float) * KS * n, 256);      // Yt
  b += align_up(sizeof(float) * n * KS, 256) * 2;  // W1, W2
  b += align_up(sizeof(float) * KS * KS, 256) * 3; // S, T, X
  b += align_up(sizeof(GemmDesc) * BT_MAX_DESC, 256);   // merge-tree descriptors
  b += align_up(bt_gemm_ws_bytes(n), 256);
  return b + 512;
}

// Zt[nrows x n] (ld ldq) <- Zt * Q^T for Q = H_0 H_1 ... (reflector j in row j of A, support i >= j + shift,
// j <= jmax), compact-WY super-blocks, last one first.  Every row is transformed independently (nrows = n for
// the full eigenvector matrix, a slice of the rows when the back-transformation is sharded over GPUs).
template <class Take>
static int backtransform_launch(const float *A, int64_t n, int64_t lda, const float *tau, int shift, int64_t jmax,
                                float *Qt, int64_t ldq, int64_t nrows, Take &take, hipStream_t stream) {
  const int ni = (int)n;
  const int nsub = bt_nsub_rows(n, nrows);
  const int64_t KS = (int64_t)KB * nsub, KSmax = (int64_t)KB * bt_nsub(n);   // (buffers are carved for the largest)
  float *Yt = (float *)take(sizeof(float) * KSmax * n);
  float *W1 = (float *)take(sizeof(float) * n * KSmax);
  float *W2 = (float *)take(sizeof(float) * n * KSmax);
  float *S = (float *)take(sizeof(float) * KSmax * KSmax);
  float *T = (float *)take(sizeof(float) * KSmax * KSmax);
  float *X = (float *)take(sizeof(float) * KSmax * KSmax);
  GemmDesc *mdesc = (GemmDesc *)take(sizeof(GemmDesc) * BT_MAX_DESC);
  const size_t gws_bytes = bt_gemm_ws_bytes(n);
  void *gws = take(gws_bytes);
  if (jmax < 0 || nrows <= 0) return VIVIT_OK;
  static unsigned long long tf_done = 0;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return VIVIT_E_LAUNCH;
    if (!(tf_done & (1ull << (dev & 63)))) {
      if (!ensure_dynamic_lds(reinterpret_cast<const void *>(bt_tfactor_kernel), BT_TF_LDS, tf_done)) return VIVIT_E_LAUNCH;
      tf_done |= 1ull << (dev & 63);
    }
  }
  int st;
  if (nsub > 1) {
    if (2 * (nsub - 1) > BT_MAX_DESC) return VIVIT_E_UNSUPPORTED;
    bt_merge_desc_kernel<<<1, 64, 0, stream>>>(S, T, X, KS, (int64_t)KB, mdesc);
  }
  for (int64_t a = (jmax / KS) * KS; a >= 0; a -= KS) {
    // reflector rows of the super-block (zero rows for indices beyond jmax), their Gram matrix, block T factors
    bt_extract_kernel<<<dim3((unsigned)cdiv(n, 256), (unsigned)KS), 256, 0, stream>>>(A, lda, ni, (int)a, Yt, shift, (int)jmax);
    st = gemm_launch(LAY_K, LAY_K, Yt, Yt, S, KS, KS, n, n, n, KS, 1.f, 0.f, true, gws, gws_bytes, stream);  // SYRK: lower tiles + mirror
    if (st != VIVIT_OK) return st;
    if (nsub > 1 && hipMemsetAsync(T, 0, sizeof(float) * KS * KS, stream) != hipSuccess) return VIVIT_E_LAUNCH;
    // block T factors of the nsub diagonal blocks: one workgroup each, one launch
    bt_tfactor_kernel<<<nsub, KB, BT_TF_LDS, stream>>>(S, KS, tau, (int)jmax, (int)a, T, KS);
    // merge tree: T12 = -T1 S12 T2 for halves of size h = KB, 2 KB, ...: the pairs of a level as one batched launch per product
    {
      int at = 0;
      for (int64_t h = KB; h < KS; h *= 2) {
        const int batch = (int)(KS / (2 * h));
        st = gemm_batched_launch(LAY_K, LAY_M, mdesc + at, batch, h, h, 1.f, 0.f, stream);
        if (st != VIVIT_OK) return st;
        st = gemm_batched_launch(LAY_K, LAY_M, mdesc + at + batch, batch, h, h, -1.f, 0.f, stream);
        if (st != VIVIT_OK) return st;
        at += 2 * batch;
      }
    }
    const int64_t m = n - a;  // components a+shift .. n-1 carry the super-block's reflectors (the columns before
                              // are zero in Yt: starting at the aligned offset a keeps the operands 16-byte aligned)
    // W1[nrows x KS] = Zt[:, a:] * Yt[:, a:]^T
    st = gemm_launch(LAY_K, LAY_K, Qt + a, Yt + a, W1, nrows, KS, m, ldq, n, KS, 1.f, 0.f, false, gws, gws_bytes, stream);
    if (st != VIVIT_OK) return st;
    // W2 = W1 * T^T
    st = gemm_launch(LAY_K, LAY_K, W1, T, W2, nrows, KS, KS, KS, KS, KS, 1.f, 0.f, false, gws, gws_bytes, stream);
    if (st != VIVIT_OK) return st;
    // Zt[:, a:] -= W2 * Yt[:, a:]
    st = gemm_launch(LAY_K, LAY_M, W2, Yt + a, Qt + a, nrows, m, KS, KS, n, ldq, -1.f, 1.f, false, gws, gws_bytes, stream);
    if (st != VIVIT_OK) return st;
  }
  return VIVIT_OK;
}

// ---- two-stage reduction (sy2sb + sb2st) --------------------------------------------------------
constexpr int TS_NB = 64;

// 1 = use the two-stage tridiagonalisation.  Values-only solves switch at n >= 2048 (the band
// reduction is MFMA-bound, the one-stage reduction HBM-bound; 53.1 / 53.6 ms at the crossover).  With eigenvectors the
// second back-transformation (Q2, q2apply.hip) has to be paid for: measured at the end of round 2 (one-stage /
// two-stage, ms) n = 2048: 58 / 63, 3072: 92 / 95, 4096: 137 / 130, 6144: 259 / 199, 8192: 420 / 272, 40960: 16400 / 4500
// (round 1 had the crossover at 8192: bulge chasing, band reduction and Q2 have since become 1.5-2x faster).
// VIVIT_TWO_STAGE=0/1 overrides.
static bool use_two_stage(int64_t n, bool vectors) {
  static int forced = -2;
  if (forced == -2) {
    const char *e = getenv("VIVIT_TWO_STAGE");
    forced = e ? atoi(e) : -1;
  }
  if (forced >= 0) return forced != 0 && n > 2 * TS_NB;
  (void)vectors;
  // Round 3: the persistent one-stage reduction (n <= 2048) beats the two stages with and without vectors (n = 2048: 24.9 /
  // 31.1 ms against 37 / ~47); above it the two stages win everywhere now that their launch chains are persistent kernels
  // (one-stage chain / two-stage, ms, eigvalsh | symeig: n = 2304: 61 / 43 | 65 / 52, 3072: 85 / 58 | 91 / 72, 4096: 129 / 79 |
  // 136 / 96 -- scripts/probe/crossover2.py).
  if (sytrd_persist_ok(n)) return false;
  return n >= 2048;
}

static size_t two_stage_workspace_bytes(int64_t n, bool vectors) {
  size_t b = 0;
  b += align_up(sizeof(float) * 16, 256) + align_up(sizeof(float) * 2 * n, 256);       // scal, scan partials
  b += align_up(sy2sb_workspace_bytes(n), 256);
  b += align_up(sizeof(float) * n * SB2ST_LDP, 256);                              // AB
  b += align_up(sizeof(float) * (vectors ? n : sb2st_ring_rows(n)) * n, 256);           // R2
  b += align_up(sizeof(float) * n * sb2st_num_levels(n), 256);                          // tau2
  b += align_up(sizeof(float) * n, 256) * 2;                                            // d, e
  return b + 1024;
}

size_t symeig_large_workspace_bytes(int64_t n, bool vectors) {
  size_t one = align_up(sizeof(float) * sytrd_workspace_floats(n), 256) + 512;
  size_t two = two_stage_workspace_bytes(n, vectors);
  size_t b = one > two ? one : two;   // either reduction may be selected at run time
  b += stedc_workspace_bytes(n, vectors);
  if (vectors) b += bt_workspace_bytes(n) + q2_workspace_bytes(n, n) + 512;
  return b;
}

// values only: prescale -> mirror -> band -> tridiagonal -> bisection
static int symeig_two_stage_values(float *A, int64_t n, int64_t lda, float *w, void *ws, int32_t *info,
                                   hipStream_t stream) {
  char *p = reinterpret_cast<char *>(align_up(reinterpret_cast<uintptr_t>(ws), 256));
  auto take = [&](size_t bytes) {
    char *r = p;
    p += align_up(bytes, 256);
    return r;
  };
  float *scal = (float *)take(sizeof(float) * 16);
  float *part = (float *)take(sizeof(float) * 2 * n);
  void *sbws = take(sy2sb_workspace_bytes(n));
  float *AB = (float *)take(sizeof(float) * n * SB2ST_LDP);
  const int64_t rrows = sb2st_ring_rows(n);
  float *R2 = (float *)take(sizeof(float) * rrows * n);
  float *tau2 = (float *)take(sizeof(float) * n * sb2st_num_levels(n));
  float *d = (float *)take(sizeof(float) * n);
  float *e = (float *)take(sizeof(float) * n);
  prof_mark(PROF_STAGE_BEGIN, stream);
  int st = prescale_launch(A, n, lda, scal, part, stream);
  if (st != VIVIT_OK) return st;
  st = symmetrize_launch(A, n, lda, stream);
  if (st != VIVIT_OK) return st;
  prof_mark(PROF_STAGE_PREP, stream);
  float *tau1;
  st = sy2sb_launch(A, n, lda, sbws, &tau1, stream);
  if (st != VIVIT_OK) return st;
  st = sy2sb_extract_band_launch(A, lda, n, AB, SB2ST_LDP, stream);
  if (st != VIVIT_OK) return st;
  prof_mark(PROF_STAGE_SY2SB, stream);
  st = sb2st_launch(AB, n, d, e, R2, n, rrows, tau2, stream);
  if (st != VIVIT_OK) return st;
  prof_mark(PROF_STAGE_SB2ST, stream);
  st = stebz_launch(d, e, n, w, scal, stream);
  if (st != VIVIT_OK) return st;
  st = info_finalize_launch(info, n, scal, stream);
  prof_mark(PROF_STAGE_TRIDIAG, stream);
  return st;
}

// rows mode (r1 >= 0): Z receives the eigenvectors r0 .. r1-1 (ascending eigenvalue order) as ROWS, [r1-r0][ldz];
// reduction and divide & conquer are done in full, only the back-transformations are restricted to those rows
// (rows of Zt are independent: this is what the multi-GPU path shards).  Default mode: Z = column eigenvectors.
static int symeig_large_impl(float *A, int64_t n, int64_t lda, float *w, float *Z, int64_t ldz, int64_t r0, int64_t r1,
                             void *ws, size_t ws_bytes, int32_t *info, hipStream_t stream) {
  const bool vectors = Z != nullptr;
  const bool rows_mode = r1 >= 0;
  if (n > 0x7fffffffLL / 8) return VIVIT_E_UNSUPPORTED;
  if (!ws || ws_bytes < symeig_large_workspace_bytes(n, vectors)) return VIVIT_E_WORKSPACE;
  if (hipMemsetAsync(info, 0, sizeof(int32_t), stream) != hipSuccess) return VIVIT_E_LAUNCH;
  char *p = reinterpret_cast<char *>(align_up(reinterpret_cast<uintptr_t>(ws), 256));
  auto take = [&](size_t bytes) {
    char *r = p;
    p += align_up(bytes, 256);
    return r;
  };

  if (!vectors && use_two_stage(n, false)) return symeig_two_stage_values(A, n, lda, w, ws, info, stream);

  if (vectors && use_two_stage(n, true)) {
    // ---- two-stage with vectors: A = Q1 B Q1^T (band), B = Q2 T Q2^T, T = Q_T diag(w) Q_T^T;
    //      Zt = Q_T^T Q2^T Q1^T, applied right to left on the rows of Qt
    float *scal = (float *)take(sizeof(float) * 16);
    float *part = (float *)take(sizeof(float) * 2 * n);
    void *sbws = take(sy2sb_workspace_bytes(n));
    float *AB = (float *)take(sizeof(float) * n * SB2ST_LDP);
    float *R2 = (float *)take(sizeof(float) * n * n);
    const size_t tau2_bytes = sizeof(float) * n * sb2st_num_levels(n);
    float *tau2 = (float *)take(tau2_bytes);
    float *d = (float *)take(sizeof(float) * n);
    float *e = (float *)take(sizeof(float) * n);
    prof_mark(PROF_STAGE_BEGIN, stream);
    int st = prescale_launch(A, n, lda, scal, part, stream);
    if (st != VIVIT_OK) return st;
    st = symmetrize_launch(A, n, lda, stream);
    if (st != VIVIT_OK) return st;
    prof_mark(PROF_STAGE_PREP, stream);
    float *tau1;
    st = sy2sb_launch(A, n, lda, sbws, &tau1, stream);
    if (st != VIVIT_OK) return st;
    st = sy2sb_extract_band_launch(A, lda, n, AB, SB2ST_LDP, stream);
    if (st != VIVIT_OK) return st;
    prof_mark(PROF_STAGE_SY2SB, stream);
    if (hipMemsetAsync(tau2, 0, tau2_bytes, stream) != hipSuccess) return VIVIT_E_LAUNCH;
    st = sb2st_launch(AB, n, d, e, R2, n, n, tau2, stream);
    if (st != VIVIT_OK) return st;
    prof_mark(PROF_STAGE_SB2ST, stream);
    void *dc_base = take(stedc_workspace_bytes(n, true));
    float *Qt, *dd;
    int *order;
    st = stedc_dc_launch(d, e, n, dc_base, &Qt, &dd, &order, info, stream);
    if (st != VIVIT_OK) return st;
    void *q2ws = take(q2_workspace_bytes(n, n));
    if (rows_mode) {
      st = dc_rows_launch(n, dd, Qt, n, order, w, Z, ldz, r0, r1, scal, stream);
      if (st != VIVIT_OK) return st;
      prof_mark(PROF_STAGE_TRIDIAG, stream);
      st = q2_apply_launch(Z, ldz, r1 - r0, n, R2, n, tau2, q2ws, stream);
      if (st != VIVIT_OK) return st;
      prof_mark(PROF_STAGE_Q2, stream);
      st = backtransform_launch(A, n, lda, tau1, TS_NB, n - TS_NB - 1, Z, ldz, r1 - r0, take, stream);
      if (st != VIVIT_OK) return st;
      prof_mark(PROF_STAGE_Q1, stream);
      return info_scal_launch(info, n, scal, stream);
    }
    prof_mark(PROF_STAGE_TRIDIAG, stream);
    st = q2_apply_launch(Qt, n, n, n, R2, n, tau2, q2ws, stream);
    if (st != VIVIT_OK) return st;
    prof_mark(PROF_STAGE_Q2, stream);
    st = backtransform_launch(A, n, lda, tau1, TS_NB, n - TS_NB - 1, Qt, n, n, take, stream);
    if (st != VIVIT_OK) return st;
    prof_mark(PROF_STAGE_Q1, stream);
    st = dc_output_launch(n, dd, Qt, n, order, w, Z, ldz, scal, info, stream);
    prof_mark(PROF_STAGE_OUTPUT, stream);
    return st;
  }

  // ---- stage 1: A = Q_H T Q_H^T
  SytrdWs tw;
  float *trd_base = (float *)take(sizeof(float) * sytrd_workspace_floats(n));
  prof_mark(PROF_STAGE_BEGIN, stream);
  int st = sytrd_launch(A, n, lda, trd_base, &tw, stream);
  if (st != VIVIT_OK) return st;
  prof_mark(PROF_STAGE_SYTRD, stream);

  if (!vectors) {
    // ---- stage 2 (values only): bisection, undo the scaling
    st = stebz_launch(tw.d, tw.e, n, w, tw.scal, stream);
    if (st != VIVIT_OK) return st;
    st = info_finalize_launch(info, n, tw.scal, stream);
    prof_mark(PROF_STAGE_TRIDIAG, stream);
    return st;
  }

  // ---- stage 2: T = Q_T diag(w) Q_T^T by divide and conquer (Qt = Q_T^T, rows unsorted)
  void *dc_base = take(stedc_workspace_bytes(n, true));
  float *Qt, *dd;
  int *order;
  st = stedc_dc_launch(tw.d, tw.e, n, dc_base, &Qt, &dd, &order, info, stream);
  if (st != VIVIT_OK) return st;

  if (rows_mode) {
    st = dc_rows_launch(n, dd, Qt, n, order, w, Z, ldz, r0, r1, tw.scal, stream);
    if (st != VIVIT_OK) return st;
    prof_mark(PROF_STAGE_TRIDIAG, stream);
    st = backtransform_launch(A, n, lda, tw.tau, 1, n - 3, Z, ldz, r1 - r0, take, stream);
    if (st != VIVIT_OK) return st;
    prof_mark(PROF_STAGE_Q1, stream);
    return info_scal_launch(info, n, tw.scal, stream);
  }
  prof_mark(PROF_STAGE_TRIDIAG, stream);

  // ---- stage 3: Zt = Qt * Q_H^T
  st = backtransform_launch(A, n, lda, tw.tau, 1, n - 3, Qt, n, n, take, stream);
  if (st != VIVIT_OK) return st;
  prof_mark(PROF_STAGE_Q1, stream);

  // ---- sort ascending, undo the scaling, deliver column eigenvectors
  st = dc_output_launch(n, dd, Qt, n, order, w, Z, ldz, tw.scal, info, stream);
  prof_mark(PROF_STAGE_OUTPUT, stream);
  return st;
}

// ---- two-phase solver: reduction + all eigenvalues, host-side criterion, then only the selected eigenvectors --------
// Phase 1 (symeig_reduce_launch) leaves in the workspace everything phase 2 needs: the tridiagonal (d, e), the
// eigenvalues in fp64, the reflector scalars (the reflectors themselves are in A and, two-stage, in R2).
// Phase 2 (symeig_select_launch): K <= SELECT_STEIN_MAX eigenvectors by inverse iteration on T (stein.hip),
// otherwise divide & conquer + gather; then the back-transformations on those K rows only (4 K n^2 flop).
constexpr int64_t SELECT_STEIN_MAX = 256;

struct SelectLayout {
  bool two_stage;
  // two-stage
  float *scal, *part;
  void *sbws;
  float *AB, *R2, *tau2, *tau1, *d, *e;
  // one-stage
  float *trd_base;
  SytrdWs tw;
  double *lam64;
  int *order_id;  // [n] identity permutation (rows mode of the D&C output wants an order array)
  char *rest;     // phase-2 scratch (stein | stedc, q2, back-transformation)
};

static bool select_two_stage(int64_t n) { return use_two_stage(n, false); }

static SelectLayout select_layout(void *ws, int64_t n) {
  char *p = reinterpret_cast<char *>(align_up(reinterpret_cast<uintptr_t>(ws), 256));
  auto take = [&](size_t bytes) {
    char *r = p;
    p += align_up(bytes, 256);
    return r;
  };
  SelectLayout L;
  L.two_stage = select_two_stage(n);
  if (L.two_stage) {
    L.scal = (float *)take(sizeof(float) * 16);
    L.part = (float *)take(sizeof(float) * 2 * n);
    L.sbws = take(sy2sb_workspace_bytes(n));
    L.AB = (float *)take(sizeof(float) * n * SB2ST_LDP);
    L.R2 = (float *)take(sizeof(float) * n * n);
    L.tau2 = (float *)take(sizeof(float) * n * sb2st_num_levels(n));
    L.tau1 = (float *)take(sizeof(float) * n);
    L.d = (float *)take(sizeof(float) * n);
    L.e = (float *)take(sizeof(float) * n);
    L.trd_base = nullptr;
  } else {
    L.trd_base = (float *)take(sizeof(float) * sytrd_workspace_floats(n));
    sytrd_layout(L.trd_base, n, &L.tw);
    L.scal = L.tw.scal;
    L.d = L.tw.d;
    L.e = L.tw.e;
    L.tau1 = L.tw.tau;
  }
  L.lam64 = (double *)take(sizeof(double) * n);
  L.rest = p;
  return L;
}

// state workspace (phase 1 writes it, phase 2 reads it)
size_t symeig_reduce_workspace_bytes(int64_t n) {
  size_t b = 0;
  if (select_two_stage(n)) {
    b += align_up(sizeof(float) * 16, 256) + align_up(sizeof(float) * 2 * n, 256);
    b += align_up(sy2sb_workspace_bytes(n), 256);
    b += align_up(sizeof(float) * n * SB2ST_LDP, 256);
    b += align_up(sizeof(float) * n * n, 256);
    b += align_up(sizeof(float) * n * sb2st_num_levels(n), 256);
    b += align_up(sizeof(float) * n, 256) * 3;
  } else {
    b += align_up(sizeof(float) * sytrd_workspace_floats(n), 256);
  }
  b += align_up(sizeof(double) * n, 256);
  return b + 1024;
}

// scratch workspace of phase 2 for K selected eigenvectors
size_t symeig_select_workspace_bytes(int64_t n, int64_t K) {
  if (K < 1) K = 1;
  if (K > n) K = n;
  size_t b = select_two_stage(n) ? q2_workspace_bytes(n, K) + 512 : 0;
  const size_t after = bt_workspace_bytes(n);
  if (K > SELECT_STEIN_MAX) {  // D&C buffers, reused by the back-transformations afterwards
    const size_t dc = align_up(stedc_workspace_bytes(n, true), 256);
    b += align_up(sizeof(float) * n, 256) + (dc > after ? dc : after);
  } else {
    b += align_up(stein_workspace_bytes(n, K), 256) + after;
  }
  return b + 1024;
}

int symeig_reduce_launch(float *A, int64_t n, int64_t lda, float *w, void *ws, size_t ws_bytes, int32_t *info,
                         hipStream_t stream) {
  if (n > 0x7fffffffLL / 8) return VIVIT_E_UNSUPPORTED;
  if (!ws || ws_bytes < symeig_reduce_workspace_bytes(n)) return VIVIT_E_WORKSPACE;
  if (hipMemsetAsync(info, 0, sizeof(int32_t), stream) != hipSuccess) return VIVIT_E_LAUNCH;
  SelectLayout L = select_layout(ws, n);
  prof_mark(PROF_STAGE_BEGIN, stream);
  int st;
  if (L.two_stage) {
    st = prescale_launch(A, n, lda, L.scal, L.part, stream);
    if (st != VIVIT_OK) return st;
    st = symmetrize_launch(A, n, lda, stream);
    if (st != VIVIT_OK) return st;
    prof_mark(PROF_STAGE_PREP, stream);
    float *tau1;
    st = sy2sb_launch(A, n, lda, L.sbws, &tau1, stream);
    if (st != VIVIT_OK) return st;
    if (hipMemcpyAsync(L.tau1, tau1, sizeof(float) * n, hipMemcpyDeviceToDevice, stream) != hipSuccess) return VIVIT_E_LAUNCH;
    st = sy2sb_extract_band_launch(A, lda, n, L.AB, SB2ST_LDP, stream);
    if (st != VIVIT_OK) return st;
    prof_mark(PROF_STAGE_SY2SB, stream);
    if (hipMemsetAsync(L.tau2, 0, sizeof(float) * n * sb2st_num_levels(n), stream) != hipSuccess) return VIVIT_E_LAUNCH;
    st = sb2st_launch(L.AB, n, L.d, L.e, L.R2, n, n, L.tau2, stream);
    if (st != VIVIT_OK) return st;
    prof_mark(PROF_STAGE_SB2ST, stream);
  } else {
    SytrdWs tw;
    st = sytrd_launch(A, n, lda, L.trd_base, &tw, stream);
    if (st != VIVIT_OK) return st;
    prof_mark(PROF_STAGE_SYTRD, stream);
  }
  st = stebz_launch(L.d, L.e, n, w, L.scal, stream, L.lam64);
  if (st != VIVIT_OK) return st;
  st = info_finalize_launch(info, n, L.scal, stream);
  prof_mark(PROF_STAGE_TRIDIAG, stream);
  return st;
}

// order[i] = sel[i] for i < K (device copy with the identity elsewhere is not needed: dc_rows_launch sorts itself)
int symeig_select_launch(const float *A, int64_t n, int64_t lda, const int *sel, int64_t K, float *Zt, int64_t ldz,
                         void *state, size_t state_bytes, void *ws, size_t ws_bytes, int32_t *info, hipStream_t stream) {
  if (K < 0 || K > n || (K > 0 && (!sel || !Zt || ldz < n))) return VIVIT_E_BADARG;
  if (!state || state_bytes < symeig_reduce_workspace_bytes(n)) return VIVIT_E_WORKSPACE;
  if (K == 0) return VIVIT_OK;
  if (!ws || ws_bytes < symeig_select_workspace_bytes(n, K)) return VIVIT_E_WORKSPACE;
  SelectLayout L = select_layout(state, n);
  char *p = reinterpret_cast<char *>(align_up(reinterpret_cast<uintptr_t>(ws), 256));
  auto take = [&](size_t bytes) {
    char *r = p;
    p += align_up(bytes, 256);
    return r;
  };
  prof_mark(PROF_STAGE_BEGIN, stream);
  int st;
  if (K <= SELECT_STEIN_MAX) {
    void *stws = take(stein_workspace_bytes(n, K));
    st = stein_launch(L.d, L.e, n, L.lam64, sel, K, Zt, ldz, stws, info, stream);
    if (st != VIVIT_OK) return st;
  } else {  // many eigenvectors: divide & conquer for all of them, keep the selected rows
    float *wscratch = (float *)take(sizeof(float) * n);
    void *dc_base = take(stedc_workspace_bytes(n, true));
    float *Qt, *dd;
    int *order;
    st = stedc_dc_launch(L.d, L.e, n, dc_base, &Qt, &dd, &order, info, stream);
    if (st != VIVIT_OK) return st;
    st = dc_select_launch(n, dd, Qt, n, order, wscratch, sel, K, Zt, ldz, stream);
    if (st != VIVIT_OK) return st;
    p = reinterpret_cast<char *>(dc_base);  // the D&C buffers are free again for the back-transformations
  }
  prof_mark(PROF_STAGE_TRIDIAG, stream);
  if (L.two_stage) {
    void *q2ws = take(q2_workspace_bytes(n, K));
    st = q2_apply_launch(Zt, ldz, K, n, L.R2, n, L.tau2, q2ws, stream);
    if (st != VIVIT_OK) return st;
    prof_mark(PROF_STAGE_Q2, stream);
    st = backtransform_launch(A, n, lda, L.tau1, TS_NB, n - TS_NB - 1, Zt, ldz, K, take, stream);
  } else {
    st = backtransform_launch(A, n, lda, L.tau1, 1, n - 3, Zt, ldz, K, take, stream);
  }
  if (st != VIVIT_OK) return st;
  prof_mark(PROF_STAGE_Q1, stream);
  return VIVIT_OK;
}

int symeig_large_launch(float *A, int64_t n, int64_t lda, float *w, float *Z, int64_t ldz, void *ws, size_t ws_bytes,
                        int32_t *info, hipStream_t stream) {
  return symeig_large_impl(A, n, lda, w, Z, ldz, 0, -1, ws, ws_bytes, info, stream);
}

int symeig_large_rows_launch(float *A, int64_t n, int64_t lda, float *w, float *Zt, int64_t ldz, int64_t r0, int64_t r1,
                             void *ws, size_t ws_bytes, int32_t *info, hipStream_t stream) {
  if (!Zt || r0 < 0 || r1 < r0 || r1 > n) return VIVIT_E_BADARG;
  return symeig_large_impl(A, n, lda, w, Zt, ldz, r0, r1, ws, ws_bytes, info, stream);
}

// ---- the two-stage solver in two halves, for a band reduction done elsewhere (multi-GPU: vivit_amd/distributed.py).
// prepare: LAPACK-style scaling + mirror (what symeig_large_impl does in front of sy2sb_launch); scal: device [16].
int symeig_prepare_launch(float *A, int64_t n, int64_t lda, float *scal, void *ws, size_t ws_bytes, hipStream_t stream) {
  if (ws_bytes < sizeof(float) * 2 * (size_t)n + 256) return VIVIT_E_WORKSPACE;
  float *part = reinterpret_cast<float *>(align_up(reinterpret_cast<uintptr_t>(ws), 256));
  int st = prescale_launch(A, n, lda, scal, part, stream);
  if (st != VIVIT_OK) return st;
  return symmetrize_launch(A, n, lda, stream);
}

int symeig_banded_rows_launch(float *A, int64_t n, int64_t lda, const float *tau1, const float *scal_in, float *w, float *Zt,
                              int64_t ldz, int64_t r0, int64_t r1, void *ws, size_t ws_bytes, int32_t *info, hipStream_t stream) {
  if (!Zt || r0 < 0 || r1 < r0 || r1 > n || n <= 2 * TS_NB) return VIVIT_E_BADARG;
  if (!ws || ws_bytes < symeig_large_workspace_bytes(n, true)) return VIVIT_E_WORKSPACE;
  if (hipMemsetAsync(info, 0, sizeof(int32_t), stream) != hipSuccess) return VIVIT_E_LAUNCH;
  char *p = reinterpret_cast<char *>(align_up(reinterpret_cast<uintptr_t>(ws), 256));
  auto take = [&](size_t bytes) {
    char *r = p;
    p += align_up(bytes, 256);
    return r;
  };
  float *scal = (float *)take(sizeof(float) * 16);
  float *AB = (float *)take(sizeof(float) * n * SB2ST_LDP);
  float *R2 = (float *)take(sizeof(float) * n * n);
  const size_t tau2_bytes = sizeof(float) * n * sb2st_num_levels(n);
  float *tau2 = (float *)take(tau2_bytes);
  float *d = (float *)take(sizeof(float) * n);
  float *e = (float *)take(sizeof(float) * n);
  if (hipMemcpyAsync(scal, scal_in, sizeof(float) * 16, hipMemcpyDeviceToDevice, stream) != hipSuccess) return VIVIT_E_LAUNCH;
  prof_mark(PROF_STAGE_BEGIN, stream);
  int st = sy2sb_extract_band_launch(A, lda, n, AB, SB2ST_LDP, stream);
  if (st != VIVIT_OK) return st;
  prof_mark(PROF_STAGE_SY2SB, stream);
  if (hipMemsetAsync(tau2, 0, tau2_bytes, stream) != hipSuccess) return VIVIT_E_LAUNCH;
  st = sb2st_launch(AB, n, d, e, R2, n, n, tau2, stream);
  if (st != VIVIT_OK) return st;
  prof_mark(PROF_STAGE_SB2ST, stream);
  void *dc_base = take(stedc_workspace_bytes(n, true));
  float *Qt, *dd;
  int *order;
  st = stedc_dc_launch(d, e, n, dc_base, &Qt, &dd, &order, info, stream);
  if (st != VIVIT_OK) return st;
  void *q2ws = take(q2_workspace_bytes(n, n));
  st = dc_rows_launch(n, dd, Qt, n, order, w, Zt, ldz, r0, r1, scal, stream);
  if (st != VIVIT_OK) return st;
  prof_mark(PROF_STAGE_TRIDIAG, stream);
  st = q2_apply_launch(Zt, ldz, r1 - r0, n, R2, n, tau2, q2ws, stream);
  if (st != VIVIT_OK) return st;
  prof_mark(PROF_STAGE_Q2, stream);
  st = backtransform_launch(A, n, lda, tau1, TS_NB, n - TS_NB - 1, Zt, ldz, r1 - r0, take, stream);
  if (st != VIVIT_OK) return st;
  prof_mark(PROF_STAGE_Q1, stream);
  return info_scal_launch(info, n, scal, stream);
}

} // namespace vivit

using namespace vivit;

extern "C" {

// The two halves of the two-stage solver around an externally computed band reduction (see vivit_hip.h).
int vivit_symeig_prepare_f32(float *A, int64_t n, int64_t lda, float *scal, void *workspace, size_t workspace_bytes, void *stream) {
  if (n < 1 || !A || !scal || lda < n || !workspace) return VIVIT_E_BADARG;
  return symeig_prepare_launch(A, n, lda, scal, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int vivit_symeig_banded_rows_f32(float *A, int64_t n, int64_t lda, const float *tau1, const float *scal, float *w, float *Zt,
                                 int64_t ldz, int64_t row_begin, int64_t row_end, void *workspace, size_t workspace_bytes,
                                 int32_t *info, void *stream) {
  if (n < 1 || !A || !tau1 || !scal || !w || !info || lda < n || ldz < n) return VIVIT_E_BADARG;
  return symeig_banded_rows_launch(A, n, lda, tau1, scal, w, Zt, ldz, row_begin, row_end, workspace, workspace_bytes, info,
                                   static_cast<hipStream_t>(stream));
}

size_t vivit_sytrd_f32_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return align_up(sizeof(float) * sytrd_workspace_floats(n), 256) + 512;
}

int vivit_sytrd_f32(float *A, int64_t n, int64_t lda, float *d, float *e, float *tau, void *workspace,
                    size_t workspace_bytes, void *stream) {
  if (n < 3 || !A || !d || !e || !tau || lda < n) return VIVIT_E_BADARG;
  if (!workspace || workspace_bytes < vivit_sytrd_f32_workspace_bytes(n)) return VIVIT_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float *base = reinterpret_cast<float *>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  SytrdWs tw;
  int st = sytrd_launch(A, n, lda, base, &tw, s);
  if (st != VIVIT_OK) return st;
  if (hipMemcpyAsync(d, tw.d, sizeof(float) * n, hipMemcpyDeviceToDevice, s) != hipSuccess) return VIVIT_E_LAUNCH;
  if (hipMemcpyAsync(e, tw.e, sizeof(float) * (n - 1), hipMemcpyDeviceToDevice, s) != hipSuccess) return VIVIT_E_LAUNCH;
  if (hipMemcpyAsync(tau, tw.tau, sizeof(float) * n, hipMemcpyDeviceToDevice, s) != hipSuccess) return VIVIT_E_LAUNCH;
  return VIVIT_OK;
}

} // extern "C"
