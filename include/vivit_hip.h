/*
 * vivit_hip.h -- C ABI of libvivit_hip.so, the MI355X (gfx950) kernel library behind the
 * low-rank GGN curvature path of ViViT (Gram build V^T V  +  symmetric eigendecomposition
 * + Gram-space -> parameter-space maps).
 *
 * The reference (f-dangel/vivit) is pure Python and has no FFI; each entry point below names
 * the reference call site (path:line under /root/reference) whose tensor op it replaces.  A
 * Python maintainer binds these with ctypes; nothing here uses torch
 * types.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to fp32 row-major data owned by the caller, unless the
 *     parameter is documented as a host pointer; leading dimensions are in ELEMENTS;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls only enqueue
 *     work and never synchronise, allocate or free device memory; scratch memory comes from the
 *     caller through (workspace, workspace_bytes) sized by the matching *_workspace_bytes query;
 *   - return value: 0 = ok; <0 = bad argument (VIVIT_E_*); a numerical failure of the
 *     eigensolver is reported asynchronously through the device-side `info` word
 *     (0 = converged, k>0 = k off-diagonal elements did not converge; n = non-finite input), which the host maps
 *     to the reference's RuntimeError (vivit/utils/eig.py:37-40,103-106).  VIVIT_INFO_PERSIST_TIMEOUT (< 0) is a
 *     failure of a different kind -- see "Persistent kernels" below;
 *   - results are deterministic (bit-identical from call to call and from process to process on the same device
 *     type; tests/test_determinism_gpu.py): every reduction has a fixed order.  The one kernel that uses float
 *     atomics, the bf16-pipe 256 x 256 tile product, adds the partial sum of each 4096-k accumulation chain into C
 *     with no-return fp32 atomics executed at the memory side -- each output element has exactly ONE writing
 *     workgroup and ONE writing lane, whose adds to that address are issued and applied in program order, so the
 *     sum is a fixed-order sum; no two workgroups ever add into the same element.
 */
#ifndef VIVIT_HIP_H
#define VIVIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIVIT_OK 0
#define VIVIT_E_BADARG (-1)     /* null pointer, negative size, ld too small               */
#define VIVIT_E_WORKSPACE (-2)  /* workspace missing or smaller than *_workspace_bytes      */
#define VIVIT_E_LAUNCH (-3)     /* hipLaunch reported an error (hipGetLastError != success) */
#define VIVIT_E_UNSUPPORTED (-4)

/* ---------------------------------------------------------------------------------------------
 * Persistent kernels.  Three stages of the eigensolver run as ONE launch whose workgroups exchange data through the
 * L2 and therefore must all be resident at once: the tridiagonalisation for n <= 2048 (sytrd_persist.hip), the panel QR
 * of the band reduction (sy2sb.hip) and the bulge chase (sb2st.hip).  Whether they are is decided once per launch by an
 * atomic arrival gate BEFORE anything is written; a launch that does not become resident within 2 s (another process or
 * a long-running kernel of another stream holds its compute units) aborts untouched and ONE retry queued behind it
 * runs.  If that fails too -- or an exchange stalls later -- the stage gives up, the solve's `info` word becomes
 * VIVIT_INFO_PERSIST_TIMEOUT (its results are garbage) and the library stays usable.  This status is distinct from the
 * numerical ones: the input may be perfectly fine.  The host wrapper (vivit_amd/kernels.py) raises
 * PersistentKernelTimeout and, when it still holds the input, repeats the solve ONCE with vivit_persistent_kernels(0),
 * i.e. on the launch chains (VIVIT_SYTRD_PERSIST / VIVIT_QR_PERSIST / VIVIT_SB2ST_PERSIST = 0 select them for good).
 * ------------------------------------------------------------------------------------------- */
#define VIVIT_INFO_PERSIST_TIMEOUT (-1000)
/* 1 / 0: allow / forbid the persistent kernels for the following calls of this process (overrides the environment
 * variables); -1: back to the environment's choice.  Returns the previous setting (-1, 0 or 1).  Host-side state only. */
int vivit_persistent_kernels(int on);
/* For callers of the STAGE-level entry points (vivit_sytrd_f32, vivit_sy2sb_f32, vivit_sy2sb_panel_qr_f32,
 * vivit_sb2st_f32), which have no info word: *info (device) = VIVIT_INFO_PERSIST_TIMEOUT if a persistent kernel launched on
 * `stream` gave up since that stream's failure word was last taken, and clears the word (there is one word per device and
 * stream, so a solve on another stream is never blamed); *info is left alone otherwise.  The vivit_symeig*_f32 entry
 * points do this themselves.  (A stage that gave up also poisons its output with NaN, so nothing fails silently.) */
int vivit_take_persist_timeout(int32_t *info, void *stream);

/* Library/ABI version (major*1000 + minor) and the gfx target it was compiled for. */
int vivit_hip_abi_version(void);   /* 1007 in this release; _lib.py refuses any other library */
const char *vivit_hip_target(void);
/* Provenance: the content hash (32 hex digits, vivit_amd/_build.py:source_hash) of the csrc/ tree + this header the library
 * was compiled from ("unknown" for a build that did not go through _build.py).  _lib.load() compares it with the sources
 * beside the package and refuses a stale binary. */
const char *vivit_hip_source_hash(void);
const char *vivit_hip_status_string(int status);

/* ---------------------------------------------------------------------------------------------
 * K1  Gram contraction (SYRK):  G = alpha * A A^T + beta * G
 *   A: [n, p] (one parameter's V_t viewed as [C*N, P_param]; K-contiguous rows, lda >= p)
 *   G: [n, n] full symmetric output (both triangles written), ldg >= n.
 * Replaces partial_contract(V, V, (2, 2)) = einsum("cn<p>,dm<p>->cndm")
 *   vivit/utils/gram.py:206-232 via pairwise_dot :9-35; callers
 *   vivit/extensions/secondorder/vivit/base.py:118-124,
 *   vivit/extensions/secondorder/sqrt_ggn/gram_sqrt_ggn.py:50-52,
 *   vivit/optim/directional_damped_newton.py:254, and the `gram += gram_p` accumulation of
 *   vivit/utils/gram.py:104-116 (beta = 1).
 * Only the lower-triangular tiles (256x256 for large outputs with a long contraction, 128x128 otherwise) are
 * computed on MFMA (n(n+1)p flops); the mirror tiles are transposed through LDS and stored as well.
 * ------------------------------------------------------------------------------------------- */
/* Matrix pipe of the large K-contiguous products (the Gram SYRK above a few hundred 256 x 256 tiles and K >= 1024, the NT
 * GEMMs of the same size, and Gram matrices of small batches with a deep contraction): 6 (default) = bf16 MFMA pipe --
 * every fp32 operand is split EXACTLY into three bf16 pieces (a = hi + mid + lo, 24 significand bits) and a product is
 * the sum of the six partial products that are >= 2^-16 of it, each exact in the MFMA, accumulated in fp32 (the three
 * dropped ones are < 2^-24 |a b|, below the rounding of an fp32 product); 9 = all nine partial products; 3 = three
 * (per-product error 2^-16: exists so that tests can show they would notice); 0 = v_mfma_f32_32x32x2_f32 on the fp32
 * operands.  Set once per process by the environment variable VIVIT_GEMM_SPLIT.  The split pieces of a 65 536-column
 * chunk live in the workspace (6 bytes per element of the chunk).
 *
 * INPUT RANGE CONTRACT (tests/test_gram_precision_gpu.py).  The split is exact only for finite values whose smallest
 * piece is a normal bf16 number.  The split pass therefore flags, per column chunk, (bit 0) any value that is +-inf,
 * NaN or rounds to +-inf as bf16 (|a| >= 3.3962e38) and (bit 1) any non-zero value below 2^-100; the bf16-pipe launch
 * of a flagged chunk returns at once and the fp32 MFMA kernel, always launched behind it on the same columns (and
 * returning at once for an unflagged chunk -- there is no host synchronisation), computes that chunk instead.  Every
 * entry point in this header thus has the semantics of a k-ordered fp32 fma chain for EVERY input: inf/NaN propagate
 * as in IEEE fp32, 3.4e38 * 0.5 is finite, denormal inputs are multiplied as fp32 denormals.  (Internal products of
 * the eigensolver honour bit 0 only.) */
int vivit_gemm_split_mode(void);
size_t vivit_gram_syrk_f32_workspace_bytes(int64_t n, int64_t p);
int vivit_gram_syrk_f32(const float *A, int64_t n, int64_t p, int64_t lda, float *G, int64_t ldg,
                        float alpha, float beta, void *workspace, size_t workspace_bytes,
                        void *stream);

/* ---------------------------------------------------------------------------------------------
 * K2/K5/K9  C = alpha * A B^T + beta * C      A: [m, k] lda, B: [n, k] ldb, C: [m, n] ldc
 * Replaces partial_contract(V, g, (2, 1))  vivit/optim/directional_damped_newton.py:255,
 *   mVp  vivit/utils/gram.py:182-203, and the gamma einsum "in,id->nd" (on transposed views)
 *   vivit/optim/directional_damped_newton.py:342.
 * ------------------------------------------------------------------------------------------- */
size_t vivit_gemm_f32_workspace_bytes(int64_t m, int64_t n, int64_t k);
int vivit_gemm_nt_f32(const float *A, const float *B, float *C, int64_t m, int64_t n, int64_t k,
                      int64_t lda, int64_t ldb, int64_t ldc, float alpha, float beta,
                      void *workspace, size_t workspace_bytes, void *stream);

/* K6/K7/K8  C = alpha * A B + beta * C        A: [m, k] lda, B: [k, n] ldb, C: [m, n] ldc
 * Replaces Vmp  vivit/utils/ggn.py:94-115 (eigenvector back-projection, eigh.py:267-270),
 *   the Newton-step back-projection einsum("cn,cn...->...")
 *   vivit/optim/directional_damped_newton.py:370-373 (m = 1), and "cni,id->cnd" :348-350. */
int vivit_gemm_nn_f32(const float *A, const float *B, float *C, int64_t m, int64_t n, int64_t k,
                      int64_t lda, int64_t ldb, int64_t ldc, float alpha, float beta,
                      void *workspace, size_t workspace_bytes, void *stream);

/* C = alpha * A^T B + beta * C                A: [k, m] lda, B: [k, n] ldb, C: [m, n] ldc
 * Replaces einsum("in,id->nd")  vivit/optim/directional_damped_newton.py:342 without a
 * transposed copy, and Y^T Z in the Householder back-transformation of the eigensolver. */
int vivit_gemm_tn_f32(const float *A, const float *B, float *C, int64_t m, int64_t n, int64_t k,
                      int64_t lda, int64_t ldb, int64_t ldc, float alpha, float beta,
                      void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * K1'  Factorised Gram of a Linear weight:  G[c,n,d,m] = alpha * Gz[n,m] * Gs[c,n,d,m] + beta*G
 *   Gz: [N, N] (z z^T), Gs: [C*N, C*N] (s s^T), G: [C*N, C*N], all contiguous.
 * Replaces einsum("nm,cndm->cndm")  vivit/extensions/secondorder/vivit/linear.py:72-75.
 * ------------------------------------------------------------------------------------------- */
int vivit_gram_hadamard_f32(const float *Gz, const float *Gs, float *G, int64_t C, int64_t N,
                            float alpha, float beta, void *stream);

/* Rectangular block of the same product: rows (c, n) with c < Cr, n < Nr; columns (d, m) with d < Cc, m < Nc:
 *   G[(c,n), (d,m)] = alpha * Gz[n, m] * Gs[(c,n), (d,m)] + beta * G[(c,n), (d,m)]
 *   Gz: [Nr, Nc], Gs: [Cr*Nr, Cc*Nc] contiguous; G: [Cr*Nr, Cc*Nc] with leading dimension ldg.
 * Used for (i) the block row G[N_g, :] of a batch shard in the data-parallel Gram build (Nr = local samples,
 * Nc = all samples; the reference has no multi-device code: SURVEY.md 8e) and (ii) V^T g of a factorised Linear
 * weight, einsum("cno,ni,mo,mi->cnm"): the factorised form of partial_contract(V, g, (2, 1))
 * vivit/optim/directional_damped_newton.py:255 with V as in linear.py:41-42 (Cc = 1). */
int vivit_gram_hadamard_block_f32(const float *Gz, const float *Gs, float *G, int64_t Cr, int64_t Nr, int64_t Cc,
                                  int64_t Nc, int64_t ldg, float alpha, float beta, void *stream);

/* Length-C contractions of the factorised Linear products (the MFMA GEMM with z does the rest):
 *   vivit_class_contract_f32:  T[f, o, n] = sum_c mat[f, c, n] * s[c, n, o]      mat: [F,C,N], s: [C,N,O], T: [F,O,N]
 *     first half of einsum("cno,vcn,ni->voi")  vivit/extensions/secondorder/vivit/linear.py:53
 *   vivit_class_expand_f32:    R[f, c, n] = sum_o s[c, n, o] * U[f, o, n]        U: [F,O,N], R: [F,C,N]
 *     second half of einsum("cno,voi,ni->vcn")  vivit/extensions/secondorder/vivit/linear.py:64 */
int vivit_class_contract_f32(const float *mat, const float *s, float *T, int64_t F, int64_t C, int64_t N, int64_t O,
                             void *stream);
int vivit_class_expand_f32(const float *s, const float *U, float *R, int64_t F, int64_t C, int64_t N, int64_t O,
                           void *stream);

/* ---------------------------------------------------------------------------------------------
 * Factor materialisation (f1): per-sample parameter Jacobian-transpose products, the `param_mjp(..., sum_batch=False)`
 * call of vivit/extensions/secondorder/vivit/base.py:84-92 (BackPACK LinearDerivatives / Conv2DDerivatives).
 *   vivit_linear_weight_mjp_f32: V[(c,n), o, i] = s[(c,n), o] * z[n, i]       s: [C*N, O], z: [N, I], V: [C*N, O*I]
 *     (einsum "vno,ni->vnoi"; what linear.py:41-42 keeps factorised, materialised for SqrtGGN{Exact,MC} / BatchGrad)
 *   vivit_conv2d_weight_mjp_f32: V[r, o, c, kh, kw] = sum_{oh,ow} M[r, o, oh, ow] * x[r % N, c, oh*sh-ph+kh*dh, ow*sw-pw+kw*dw]
 *     M: [rows, Cout, OH, OW], x: [N, Cin, H, W], V: [rows, Cout*Cin*KH*KW]; rows = C*N (class-major), groups = 1,
 *     zero padding (unfold + einsum "vnol,nkl->vnok" without the im2col buffer).  Per row a GEMM on the fp32 matrix pipe
 *     when the sample's zero-bordered planes and the row of M fit the LDS (and OW >= 4), else a scalar kernel; the two
 *     differ in summation order only (VIVIT_CONV_MFMA=0 selects the scalar kernels).
 * The Linear rule is bound by the 4*rows*P bytes it writes.
 * ------------------------------------------------------------------------------------------- */
int vivit_linear_weight_mjp_f32(const float *s, const float *z, float *V, int64_t C, int64_t N, int64_t O, int64_t I,
                                void *stream);
int vivit_conv2d_weight_mjp_f32(const float *M, const float *x, float *V, int64_t rows, int64_t N, int64_t Cin, int64_t H,
                                int64_t W, int64_t Cout, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int64_t sh,
                                int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw, void *stream);

/* ---------------------------------------------------------------------------------------------
 * f1  Back-propagation of the sqrt-GGN factor through the layers (jacobians.hip): the transposed input-Jacobian products
 *   M [V, N, *out] -> [V, N, *in]  that BackPACK's derivative classes supply to the reference through `MatToJacMat`
 *   (vivit/extensions/secondorder/vivit/__init__.py:84-118, base.py:19,41), the loss-Hessian square roots that seed it
 *   (SqrtGGNCrossEntropyLoss, __init__.py:84-86) and the reductions of the bias / BatchNorm parameter rules
 *   (base.py:84-92 -> param_mjp).  All tensors contiguous fp32, `V` slices outermost.
 * ------------------------------------------------------------------------------------------- */
/* out[v, e] = M[v, e] * f'(x[e]), e < per_v (= N * features).  kind: 0 ReLU, 1 Sigmoid, 2 Tanh, 3 LeakyReLU (param =
 * negative slope), 4 LogSigmoid, 5 ELU (param = alpha), 6 SELU.  Replaces SqrtGGN{ReLU,Sigmoid,Tanh,LeakyReLU,LogSigmoid,
 * ELU,SELU} (__init__.py:87-93). */
int vivit_act_jac_t_f32(const float *M, const float *x, float *out, int64_t V, int64_t per_v, int kind, float param,
                        void *stream);
/* out[r, c, l] = M[r, c, l] * scale[c]: BatchNorm in eval mode, scale = weight / sqrt(running_var + eps)
 * (vivit/extensions/secondorder/vivit/batchnormnd.py:8-13 -> BatchNormNdDerivatives). */
int vivit_channel_scale_f32(const float *M, const float *scale, float *out, int64_t rows, int64_t C, int64_t L, void *stream);
/* MaxPool2d / AvgPool2d (SqrtGGNMaxPool2d / SqrtGGNAvgPool2d, __init__.py:97-102): planes = N * C of the layer input
 * x [planes, H, W]; M [V * planes, OH, OW] -> out [V * planes, H, W].  idx_ws: planes * OH * OW ints of scratch (the
 * arg-max of every window: first maximum in scan order, as torch).  No ceil_mode, no dilation; average pooling counts
 * the padding (count_include_pad). */
int vivit_maxpool2d_jac_t_f32(const float *M, const float *x, float *out, int *idx_ws, int64_t V, int64_t planes, int64_t H,
                              int64_t W, int64_t OH, int64_t OW, int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph,
                              int64_t pw, void *stream);
int vivit_avgpool2d_jac_t_f32(const float *M, float *out, int64_t rows_planes, int64_t H, int64_t W, int64_t OH, int64_t OW,
                              int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, void *stream);
/* Conv2d (groups = 1, zero padding): out[r, ci, h, w] = sum_{co, a, b} M[r, co, oh, ow] weight[co, ci, a, b] with
 * oh sh - ph + a dh = h, ow sw - pw + b dw = w  (the transposed convolution; base.py:19 with Conv2DDerivatives,
 * convnd.py:17-22).  rows = V * N.  Filter slices of up to Cout*KH*KW = 1024 run on a scalar kernel (16 input channels per
 * thread, the slice in LDS), larger ones as a per-row GEMM on the fp32 matrix pipe (output channels in chunks that fit the
 * LDS next to the zero-inserted planes of M); VIVIT_E_UNSUPPORTED when neither applies. */
int vivit_conv2d_jac_t_f32(const float *M, const float *weight, float *out, int64_t rows, int64_t Cin, int64_t H, int64_t W,
                           int64_t Cout, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int64_t sh, int64_t sw, int64_t ph,
                           int64_t pw, int64_t dh, int64_t dw, void *stream);
/* out[r] = sum_l M[r, l] * (X ? X[r % rows_x, l] : 1): bias rules (sum over the spatial positions) and the BatchNorm
 * weight rule (X = normalised input of the rows_x = N * C planes, shared by the V slices). */
int vivit_row_dot_f32(const float *M, const float *X, float *out, int64_t rows, int64_t rows_x, int64_t L, void *stream);
/* BatchNorm in eval mode: every rule of the module in ONE pass over the incoming factor M [rows = V N C, L] (L = spatial
 * positions; X [rows_x = N C, L] the module's input): mx[r] = sum_l M[r, l] X[r % rows_x, l] and msum[r] = sum_l M[r, l]
 * (the weight rule sum_l M xhat = (mx - mean_c msum) rstd_c and the bias rule; batchnormnd.py:3 with
 * BatchNormNdDerivatives.param_mjp), out[r, l] = M[r, l] scale[r % C] (the input rule, scale = weight_c rstd_c).  Any of
 * out / mx / msum may be NULL.  Same summation order as vivit_row_dot_f32 (bit-identical sums).  wmean / wrstd ([C], both or
 * neither): mx then holds the finished weight rule (mx - wmean_c msum) wrstd_c with wmean = running_mean, wrstd = 1/sqrt(var + eps). */
int vivit_bn_eval_rules_f32(const float *M, const float *X, const float *scale, float *out, float *mx, float *msum, int64_t rows,
                            int64_t rows_x, int64_t C, int64_t L, const float *wmean, const float *wrstd, void *stream);
/* Cross-entropy loss-Hessian square root from the logits [N, C]: p = softmax.  onehot == NULL (exact, V must equal C):
 * S[v, n, c] = sqrt(p_nv) (delta_vc - p_nc) scale;  onehot [V, N, C] (sampled): S[v, n, c] = (p_nc - onehot[v, n, c]) scale. */
int vivit_ce_sqrt_hessian_f32(const float *logits, const float *onehot, float *S, int64_t N, int64_t C, int64_t V, float scale,
                              void *stream);

/* ---------------------------------------------------------------------------------------------
 * K3/K4  Symmetric eigendecomposition (Householder tridiagonalisation + implicit-shift QL,
 * divide-and-conquer merges above the single-workgroup size).
 *   A: [n, n] symmetric, lda >= n.  DESTROYED (holds the Householder reflectors on return).
 *      Only the lower triangle (A[i][j], i >= j) is read.
 *   w: [n] eigenvalues, ascending.
 *   Z: [n, n] ldz >= n or NULL.  Z[:, i] is the unit eigenvector of w[i] (column-wise, as
 *      Tensor.symeig returned them: vivit/utils/eig.py:24-26).  NULL = values only.
 *   info: device int32; 0 on success, >0 = number of unconverged eigenvalues.
 * Replaces Tensor.symeig(eigenvectors=False)  vivit/linalg/eigvalsh.py:221 and
 *   Tensor.symeig(eigenvectors=True)  vivit/linalg/eigh.py:248-250,
 *   vivit/optim/directional_damped_newton.py:315, vivit/optim/directional_derivatives.py:291,
 *   vivit/utils/eig.py:38,104.
 * ------------------------------------------------------------------------------------------- */
size_t vivit_symeig_f32_workspace_bytes(int64_t n, int want_vectors);
int vivit_symeig_f32(float *A, int64_t n, int64_t lda, float *w, float *Z, int64_t ldz,
                     void *workspace, size_t workspace_bytes, int32_t *info, void *stream);

/* Row-range variant for the multi-GPU path: all n eigenvalues (ascending) plus the eigenvectors
 * row_begin .. row_end-1 (in that order) as ROWS of Zt: [row_end - row_begin, n], ldz >= n.  Reduction and
 * tridiagonal solve run in full on every caller; only the back-transformations, which act on each eigenvector
 * independently, are restricted to the requested range - R ranks that hold the same A (after the all-reduce
 * of the partial Gram matrices) each pay 1/R of that stage and exchange their slices.
 * Same workspace as vivit_symeig_f32(want_vectors = 1).  n <= 192: VIVIT_E_UNSUPPORTED (use vivit_symeig_f32).
 * Replaces the same Tensor.symeig(eigenvectors=True) call sites as vivit_symeig_f32. */
int vivit_symeig_rows_f32(float *A, int64_t n, int64_t lda, float *w, float *Zt, int64_t ldz, int64_t row_begin,
                          int64_t row_end, void *workspace, size_t workspace_bytes, int32_t *info, void *stream);

/* Two-phase variant for callers that keep only a few eigenvectors (SURVEY.md 8b `symeig_select_f32`).  The reference
 * computes all n eigenvectors and slices them with the index list its `criterion` callback returns from the
 * eigenvalues: evals, evecs = symeig(...); keep = criterion(evals); evecs[:, keep]
 *   vivit/linalg/eigh.py:248-253, vivit/optim/directional_damped_newton.py:315-321,
 *   vivit/optim/directional_derivatives.py:291-297.
 * Here the callback runs between two calls:
 *   vivit_symeig_reduce_f32: A (destroyed, as in vivit_symeig_f32) -> tridiagonal form; w: all n eigenvalues,
 *     ascending (Sturm multisection).  `state` (vivit_symeig_reduce_f32_workspace_bytes(n) bytes) receives the
 *     tridiagonal, the fp64 eigenvalues, the reflector scalars and the second-stage reflectors; A keeps the
 *     first-stage reflectors.  Neither may be touched before phase 2.
 *   vivit_symeig_select_f32: idx: DEVICE int32 [K], strictly ascending positions into w; Zt: [K, n] (ldz >= n),
 *     row k = unit eigenvector of w[idx[k]].  K <= 256: inverse iteration on the tridiagonal (fp64), else divide &
 *     conquer; then only those K rows are back-transformed (4 K n^2 flop instead of 4 n^3).  May be called more than
 *     once per reduction.  Same (A, n, lda, state) as phase 1; `workspace` is scratch of
 *     vivit_symeig_select_f32_workspace_bytes(n, K) bytes.
 * n <= 192: VIVIT_E_UNSUPPORTED (use vivit_symeig_f32 and slice). */
size_t vivit_symeig_reduce_f32_workspace_bytes(int64_t n);
size_t vivit_symeig_select_f32_workspace_bytes(int64_t n, int64_t K);
int vivit_symeig_reduce_f32(float *A, int64_t n, int64_t lda, float *w, void *state, size_t state_bytes,
                            int32_t *info, void *stream);
int vivit_symeig_select_f32(const float *A, int64_t n, int64_t lda, const int32_t *idx, int64_t K, float *Zt, int64_t ldz,
                            void *state, size_t state_bytes, void *workspace, size_t workspace_bytes, int32_t *info,
                            void *stream);

/* Stage 1 of vivit_symeig_f32, exported for testing: Householder tridiagonalisation
 * A = Q T Q^T (lower triangle read).  d: [n], e: [n-1], tau: [n]; on return row j of A's upper
 * triangle, A[j][j+1:], holds reflector v_j (v_j[j+1] = 1), Q = H_0 ... H_{n-3},
 * H_j = I - tau[j] v_j v_j^T.  n >= 3. */
size_t vivit_sytrd_f32_workspace_bytes(int64_t n);
int vivit_sytrd_f32(float *A, int64_t n, int64_t lda, float *d, float *e, float *tau,
                    void *workspace, size_t workspace_bytes, void *stream);

/* Stage 1a of the two-stage path, exported for testing: full symmetric A (BOTH triangles valid,
 * lda >= n) -> symmetric band with half bandwidth vivit_sb2st_half_bandwidth() by blocked
 * Householder QR of the sub-band panels and two-sided MFMA updates.  On return the band is in A
 * (A[i][j], 0 <= i-j <= NB) and, copied, in AB ([n][2*NB+1] row-band layout); reflector c of
 * panel p sits in A[p*NB + c][(p+1)*NB + c ..]; tau1: [n]. */
size_t vivit_sy2sb_f32_workspace_bytes(int64_t n);
int vivit_sy2sb_f32(float *A, int64_t n, int64_t lda, float *AB, float *tau1, void *workspace,
                    size_t workspace_bytes, void *stream);

/* The pieces of the two-stage solver around a band reduction that is done by the CALLER -- the multi-GPU path
 * (vivit_amd/distributed.py:sy2sb_sharded): the trailing matrix is sharded by block columns over the
 * ranks, the sub-band panel is broadcast and factored by every rank, the rest of the solve runs as in
 * vivit_symeig_rows_f32.
 *   vivit_symeig_prepare_f32: LAPACK-style scaling of A + mirror of the lower triangle (A then has BOTH triangles);
 *     scal: DEVICE float[16] (scal[1] = the factor applied, scal[2] = non-finite-input flag), to be handed to
 *     vivit_symeig_banded_rows_f32.  workspace: 8 n + 256 bytes.
 *   vivit_sy2sb_panel_qr_f32: Householder QR of one panel, pan: [mp][NB] row-major (NB = vivit_sb2st_half_bandwidth()),
 *     factored in place: R's strict upper triangle in rows 0..NB-1, its diagonal in betas[NB].  Vt: [NB][ldv] receives
 *     the reflectors as ROWS (Vt[c][r] = v_c[r], v_c[c] = 1, zeros in front), tau: [NB], T: [NB][NB] upper triangular
 *     with Q = I - V T V^T (LAPACK larft, forward / columnwise).
 *   vivit_symeig_banded_rows_f32: A as vivit_sy2sb_f32 leaves it (band in A[i][j], 0 <= i-j <= NB; reflector c of
 *     panel p in A[p*NB + c][(p+1)*NB + c ..]), tau1: DEVICE [n]; continues with bulge chasing, divide & conquer and the
 *     two back-transformations restricted to the eigenvectors row_begin .. row_end-1 (ROWS of Zt, as
 *     vivit_symeig_rows_f32).  Workspace of vivit_symeig_f32(want_vectors = 1).  n > 2 NB.
 * Replace the same Tensor.symeig(eigenvectors=True) call sites as vivit_symeig_f32. */
int vivit_symeig_prepare_f32(float *A, int64_t n, int64_t lda, float *scal, void *workspace, size_t workspace_bytes,
                             void *stream);
size_t vivit_sy2sb_panel_qr_f32_workspace_bytes(int64_t mp);
int vivit_sy2sb_panel_qr_f32(float *pan, int64_t mp, float *Vt, int64_t ldv, float *tau, float *betas, float *T,
                             void *workspace, size_t workspace_bytes, void *stream);
int vivit_symeig_banded_rows_f32(float *A, int64_t n, int64_t lda, const float *tau1, const float *scal, float *w,
                                 float *Zt, int64_t ldz, int64_t row_begin, int64_t row_end, void *workspace,
                                 size_t workspace_bytes, int32_t *info, void *stream);

/* Stage 1b of the two-stage path, exported for testing: symmetric BAND -> tridiagonal by bulge
 * chasing.  Half bandwidth NB = vivit_sb2st_half_bandwidth() (64).  AB: [n][2*NB+1] row-band
 * layout, AB[i][j - i + 2*NB] = A[i][j] for i - 2*NB <= j <= i (entries with i - j > NB must be
 * zero on entry: bulge room); contents unspecified afterwards (the chase runs on a copy with padded
 * rows inside the workspace).  d: [n], e: [n].  R2: [n][n], row s receives the
 * Householder vectors of sweep s (for the back-transformation). */
int vivit_sb2st_half_bandwidth(void);
size_t vivit_sb2st_f32_workspace_bytes(int64_t n);
int vivit_sb2st_f32(float *AB, int64_t n, float *d, float *e, float *R2, void *workspace,
                    size_t workspace_bytes, void *stream);

/* Stage 3a of the two-stage path, exported for testing: the back-transformation through the bulge-chasing reflectors,
 *   Zt <- Zt * Q2^T,   Q2 = product of all H(s, k) = I - tau2[s][k] v(s,k) v(s,k)^T in generation order,
 * Zt: [nrows][ldz] (rows = eigenvectors of the tridiagonal matrix on entry, of the band matrix on return; any subset of
 * rows: rows are independent), R2 / tau2 as vivit_sb2st_f32 leaves them (tau2: [n][n / NB + 1 rounded up, see
 * vivit_sb2st_f32_workspace_bytes] in its workspace).  mode 0: one launch per wavefront step of reflector blocks
 * (fp32 MFMA, q2apply.hip); mode 1: one persistent launch per ~2 GB of block images, a row slab per workgroup with a
 * sliding column window, fp32 products from exact three-way bf16 splits (q2slide.hip; needs n % 4 == 0, ldz % 4 == 0,
 * 16-byte aligned Zt: VIVIT_E_UNSUPPORTED otherwise); mode -1: what vivit_symeig_f32 would pick for this shape.
 * Replaces the eigenvector half of Tensor.symeig(eigenvectors=True), vivit/linalg/eigh.py:248-250. */
size_t vivit_q2_apply_f32_workspace_bytes(int64_t n);
int vivit_q2_apply_f32(float *Zt, int64_t ldz, int64_t nrows, int64_t n, const float *R2, int64_t ldr, const float *tau2,
                       void *workspace, size_t workspace_bytes, int mode, void *stream);

/* Eigen-decomposition of a symmetric TRIDIAGONAL matrix (d: [n] diagonal, e: [n-1]
 * off-diagonal; both destroyed).  Stage 2 of vivit_symeig_f32, exported for testing. */
size_t vivit_stedc_f32_workspace_bytes(int64_t n, int want_vectors);
int vivit_stedc_f32(float *d, float *e, int64_t n, float *w, float *Z, int64_t ldz,
                    void *workspace, size_t workspace_bytes, int32_t *info, void *stream);

/* ---------------------------------------------------------------------------------------------
 * K6  Directional curvatures from the Gram eigenvectors:
 *   lambdas[n, k] = scale * sum_c ( sum_i G[(c,n), i] E[i, k] )^2 / evals[k]
 *   G: [C*N, C*N], E: [C*N, K] (ld = lde), evals: [K], lambdas: [N, K].
 *   GE: [C*N, K] scratch holding G @ E (computed by vivit_gemm_nn_f32 beforehand).
 * Replaces (V_n_T_V_e_d ** 2).sum(0) / evals
 *   vivit/optim/directional_damped_newton.py:348-351, directional_derivatives.py:322-325.
 * ------------------------------------------------------------------------------------------- */
int vivit_dir_curvature_f32(const float *GE, const float *evals, float *lambdas, int64_t C,
                            int64_t N, int64_t K, float scale, void *stream);

/* K5 epilogue / K10: out[r, k] = in[r, k] * (pre) / sqrt(evals[k])   (column scaling)
 * Replaces "/ evals.sqrt()"  vivit/optim/directional_damped_newton.py:342. */
int vivit_scale_cols_rsqrt_f32(float *X, const float *evals, int64_t rows, int64_t K, int64_t ldx,
                               float pre, void *stream);

/* K10  Squared 2-norms of K stacked vectors: acc[k] += sum_j X[k, j]^2, X: [K, len] contiguous;
 *      then vivit_scale_rows_f32 applies X[k, :] *= rsqrt(acc[k]).
 * Replaces normalize  vivit/linalg/utils.py:67-76. */
size_t vivit_row_sqnorm_workspace_bytes(int64_t K, int64_t len);
int vivit_row_sqnorm_acc_f32(const float *X, float *acc, int64_t K, int64_t len, void *workspace,
                             size_t workspace_bytes, void *stream);
int vivit_scale_rows_rsqrt_f32(float *X, const float *acc, int64_t K, int64_t len, void *stream);

/* Optional kernel timing for roofline reports (bench.py): between begin and end, every Gram SYRK
 * launch and every `symv_stride`-th symmetric matrix-vector launch of the tridiagonalisation is
 * bracketed by HIP events on its stream.  vivit_profile_end synchronises on those events and
 * fills out[6] = {syrk launches, syrk ms, syrk flops n(n+1)p,  symv launches, symv ms,
 * symv algorithmic bytes 4*m(m+1)/2}.  Not thread-safe; leave off in production. */
int vivit_profile_begin(int symv_stride);
int vivit_profile_end(double *out);
/* Per-stage time of the vivit_symeig*_f32 calls issued since vivit_profile_begin (HIP events at the stage
 * boundaries, on the launch stream), summed over calls, in ms.  out_ms[k], k < num (host pointer):
 *   1 prescale + mirror, 2 full -> band (sy2sb), 3 band -> tridiagonal (sb2st), 4 tridiagonal eigenproblem
 *   (divide & conquer | Sturm multisection | inverse iteration), 5 back-transformation Q2, 6 back-transformation
 *   Q1 / Q, 7 sort + transpose into the output, 8 one-stage tridiagonalisation (sytrd); two parts of stage 2 are
 *   reported on their own and NOT included in out_ms[2]: 9 its streaming panel products P^T = V^T A22 (fp32 MFMA),
 *   10 its delayed trailing updates (bf16 pipe).  Synchronises on the
 *   recorded events and clears them; call before vivit_profile_end or after, once. */
int vivit_profile_stages(double *out_ms, int num);

/* Mirror the lower triangle of G into the upper triangle (G[i][j] = G[j][i], i < j). */
int vivit_symmetrize_lower_f32(float *G, int64_t n, int64_t ldg, void *stream);

/* Lower triangle of a symmetric matrix <-> packed vector, packed[i (i + 1) / 2 + j] = G[i][j] (j <= i; n (n + 1) / 2 floats):
 * the partial Gram matrices of the ranks are all-reduced in this form (half the bytes of `gram += gram_p` summed over
 * ranks, vivit/utils/gram.py:104-116).  unpack writes the lower triangle and mirrors it (both triangles valid). */
int vivit_pack_lower_f32(const float *G, int64_t n, int64_t ldg, float *packed, void *stream);
int vivit_unpack_lower_f32(const float *packed, int64_t n, float *G, int64_t ldg, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VIVIT_HIP_H */
