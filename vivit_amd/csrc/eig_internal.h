// Internal interfaces between the eigensolver translation units (host-side launch wrappers only:
// kernels are launched from the file that defines them).
#pragma once
#include "common.h"

namespace vivit {

struct SytrdWs {
  float *vw;        // [3*PB][n]: V (PB rows) | W (PB rows) | V again  (so [V;W] and [W;V] are both contiguous)
  float *xbuf;      // [n]
  float *rowpart;   // [nct][n]
  float *colpart;   // [nrt][n]
  float *dotpart;   // [nct][2*PB]
  float *cvw;       // [2*PB]   c_v = V^T v | c_w = W^T v
  float *ssqpart;   // [nwg]
  float *wdotpart;  // [nwg]
  float *scal;      // [16]  0: alpha  1: sigma  2: bad-input flag  3: amax
  float *d, *e, *tau;
};

// sytrd.hip
size_t sytrd_workspace_floats(int64_t n);
int sytrd_launch(float *A, int64_t n, int64_t lda, float *wsbase, SytrdWs *out, hipStream_t stream);

// sytrd_persist.hip: the same reduction as one persistent launch on the 32 CUs of one XCD (n <= 1280; after prescale_launch)
bool sytrd_persist_ok(int64_t n);
int sytrd_persist_launch(float *A, int64_t n, int64_t lda, const SytrdWs &ws, hipStream_t stream);

// stedc.hip
size_t stedc_workspace_bytes(int64_t n, bool vectors);
int stedc_dc_launch(const float *d, const float *e, int64_t n, void *wsbase, float **Qt_out, float **d_out,
                    int **order_scratch, int32_t *info, hipStream_t stream);
// w[m] = m-th smallest eigenvalue of (d, e) by bisection, divided by scal[1] when scal != nullptr
// (w64, optional: the same eigenvalues in fp64, NOT divided by the scale)
int stebz_launch(const float *d, const float *e, int64_t n, float *w, const float *scal, hipStream_t stream,
                 double *w64 = nullptr);
// stein.hip: selected eigenvectors of the tridiagonal (d, e) by inverse iteration, rows of Zt
size_t stein_workspace_bytes(int64_t n, int64_t K);
int stein_launch(const float *d, const float *e, int64_t n, const double *lam64, const int *sel, int64_t K, float *Zt,
                 int64_t ldz, void *wsbase, int32_t *info, hipStream_t stream);
// sytrd.hip: pointers into a workspace laid out by sytrd_launch (same base, same n) without launching anything
void sytrd_layout(float *wsbase, int64_t n, SytrdWs *out);
// two-phase eigensolver for criterion-selected eigenvectors (symeig_large.hip)
size_t symeig_reduce_workspace_bytes(int64_t n);
size_t symeig_select_workspace_bytes(int64_t n, int64_t K);
int symeig_reduce_launch(float *A, int64_t n, int64_t lda, float *w, void *ws, size_t ws_bytes, int32_t *info,
                         hipStream_t stream);
int symeig_select_launch(const float *A, int64_t n, int64_t lda, const int *sel, int64_t K, float *Zt, int64_t ldz,
                         void *state, size_t state_bytes, void *ws, size_t ws_bytes, int32_t *info, hipStream_t stream);
// w = sorted(dcur) / sigma;  Z[i][p] = Qt[order[p]][i];  info = n if the input was non-finite
int symeig_large_rows_launch(float *A, int64_t n, int64_t lda, float *w, float *Zt, int64_t ldz, int64_t r0, int64_t r1,
                             void *ws, size_t ws_bytes, int32_t *info, hipStream_t stream);
int dc_rows_launch(int64_t n, const float *dcur, const float *Qt, int64_t ldq, int *order, float *w, float *Zs,
                   int64_t ldz, int64_t r0, int64_t r1, const float *scal, hipStream_t stream);
int info_scal_launch(int32_t *info, int64_t n, const float *scal, hipStream_t stream);
// rows Zs[s] = eigenvector at ascending position sel[s] (device int32 [K]) of a finished divide & conquer
int dc_select_launch(int64_t n, const float *dcur, const float *Qt, int64_t ldq, int *order, float *wscratch,
                     const int *sel, int64_t K, float *Zs, int64_t ldz, hipStream_t stream);
int dc_output_launch(int64_t n, const float *dcur, const float *Qt, int64_t ldq, int *order, float *w, float *Z,
                     int64_t ldz, const float *scal, int32_t *info, hipStream_t stream);

// sy2sb.hip / sb2st.hip (two-stage tridiagonalisation)
size_t sy2sb_workspace_bytes(int64_t n);
int sy2sb_launch(float *A, int64_t n, int64_t lda, void *wsbase, float **tau1_out, hipStream_t stream);
// Row stride of the band array INSIDE the library: the 2 NB + 1 = 129 entries of a band row + 3 floats of padding, so that the
// bulge chase can write a row's [E | D] segment with 16-byte stores (what such a store writes beyond the diagonal entry lands in
// the padding).  The public entry points (vivit_sy2sb_f32, vivit_sb2st_f32) keep rows of 129.
constexpr int SB2ST_LDP = 132;
int sy2sb_extract_band_launch(const float *A, int64_t lda, int64_t n, float *AB, int64_t ldab, hipStream_t stream);
size_t sy2sb_panel_qr_workspace_bytes(int64_t mp);
int sy2sb_panel_qr_launch(float *pan, int64_t mp, float *Vt, int64_t ldv, float *tau, float *betas, float *T, void *wsbase,
                          size_t ws_bytes, hipStream_t stream);
// symeig_large.hip: the two-stage solver entered AFTER the band reduction (A holds band + first-stage reflectors, as
// sy2sb_launch leaves it; tau1, scal: device) -- rows r0 .. r1-1 of the eigenvector matrix
int symeig_banded_rows_launch(float *A, int64_t n, int64_t lda, const float *tau1, const float *scal_in, float *w, float *Zt,
                              int64_t ldz, int64_t r0, int64_t r1, void *ws, size_t ws_bytes, int32_t *info, hipStream_t stream);
int symeig_prepare_launch(float *A, int64_t n, int64_t lda, float *scal, void *ws, size_t ws_bytes, hipStream_t stream);
int sb2st_num_levels(int64_t n);
int64_t sb2st_ring_rows(int64_t n);
int sb2st_launch(float *AB, int64_t n, float *d, float *e, float *R2, int64_t ldr, int64_t r2rows, float *tau2,
                 hipStream_t stream);
// sytrd.hip: scan (amax, non-finite flag) + LAPACK-style scaling of the lower triangle; scal: [16], part: [2n]
int prescale_launch(float *A, int64_t n, int64_t lda, float *scal, float *part, hipStream_t stream);
// elementwise.hip
int symmetrize_launch(float *G, int64_t n, int64_t ldg, hipStream_t stream);

// q2apply.hip: Zt[nrows x n] <- Zt * Q2^T (Q2 = bulge-chasing reflectors of sb2st_launch, R2 with r2rows = n)
size_t q2_workspace_bytes(int64_t n, int64_t max_rows);   // max_rows: most rows one call will transform (< 0: unknown)
int q2_apply_launch(float *Zt, int64_t ldz, int64_t nrows, int64_t n, const float *R2, int64_t ldr, const float *tau2,
                    void *ws, hipStream_t stream, int mode = -1);
// q2slide.hip: the same transformation with a sliding window per row slab on the bf16 matrix pipe (many rows)
size_t q2_slide_workspace_bytes(int64_t n);
bool q2_slide_possible(int64_t nrows, int64_t n);   // by shape and environment only (workspace queries)
bool q2_slide_ok(int64_t nrows, int64_t n, const float *Zt, int64_t ldz);
int q2_slide_launch(float *Zt, int64_t ldz, int64_t nrows, int64_t n, const float *R2, int64_t ldr, const float *tau2, void *ws,
                    size_t ws_bytes, hipStream_t stream);

// info = n when the scan flagged non-finite input (scal[2] != 0)
int info_finalize_launch(int32_t *info, int64_t n, const float *scal, hipStream_t stream);

} // namespace vivit
