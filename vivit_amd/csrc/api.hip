// C-ABI glue: version queries and the symeig dispatcher (single-workgroup vs multi-kernel path).
#include "common.h"

namespace vivit {
constexpr int SMALL_N_MAX = 192;
int symeig_small_launch(const float *A, int64_t lda, int n, float *w, float *Z, int64_t ldz, int32_t *info,
                        hipStream_t stream);
size_t symeig_large_workspace_bytes(int64_t n, bool vectors);
int symeig_large_launch(float *A, int64_t n, int64_t lda, float *w, float *Z, int64_t ldz, void *ws, size_t ws_bytes,
                        int32_t *info, hipStream_t stream);
int symeig_large_rows_launch(float *A, int64_t n, int64_t lda, float *w, float *Zt, int64_t ldz, int64_t r0, int64_t r1,
                             void *ws, size_t ws_bytes, int32_t *info, hipStream_t stream);
size_t symeig_reduce_workspace_bytes(int64_t n);
size_t symeig_select_workspace_bytes(int64_t n, int64_t K);
int symeig_reduce_launch(float *A, int64_t n, int64_t lda, float *w, void *ws, size_t ws_bytes, int32_t *info,
                         hipStream_t stream);
int symeig_select_launch(const float *A, int64_t n, int64_t lda, const int *sel, int64_t K, float *Zt, int64_t ldz,
                         void *state, size_t state_bytes, void *ws, size_t ws_bytes, int32_t *info, hipStream_t stream);
} // namespace vivit

using namespace vivit;

extern "C" {

// vivit_hip_source_hash() is NOT defined here: vivit_amd/_build.py generates it at link time (csrc/obj/link/buildinfo.c)
// from the content hash of the tree AND the digest of the compile flags of every object it links, so a library relinked
// from product objects plus a differently-compiled one (timing-only variants) cannot carry the product's hash.

int vivit_hip_abi_version(void) { return 1007; }
const char *vivit_hip_target(void) { return "gfx950"; }

const char *vivit_hip_status_string(int status) {
  switch (status) {
    case VIVIT_OK: return "ok";
    case VIVIT_E_BADARG: return "bad argument (null pointer, negative size or leading dimension too small)";
    case VIVIT_E_WORKSPACE: return "workspace missing or too small";
    case VIVIT_E_LAUNCH: return "HIP kernel launch failed";
    case VIVIT_E_UNSUPPORTED: return "unsupported problem size";
    default: return "unknown status";
  }
}

size_t vivit_symeig_f32_workspace_bytes(int64_t n, int want_vectors) {
  if (n <= SMALL_N_MAX) return 0;
  return symeig_large_workspace_bytes(n, want_vectors != 0);
}

int vivit_symeig_f32(float *A, int64_t n, int64_t lda, float *w, float *Z, int64_t ldz, void *workspace,
                     size_t workspace_bytes, int32_t *info, void *stream) {
  if (n < 0) return VIVIT_E_BADARG;
  if (!info) return VIVIT_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n == 0) return hipMemsetAsync(info, 0, sizeof(int32_t), s) == hipSuccess ? VIVIT_OK : VIVIT_E_LAUNCH;
  if (!A || !w || lda < n || (Z && ldz < n)) return VIVIT_E_BADARG;
  if (n <= SMALL_N_MAX) return symeig_small_launch(A, lda, (int)n, w, Z, ldz, info, s);
  return symeig_large_launch(A, n, lda, w, Z, ldz, workspace, workspace_bytes, info, s);
}

// Eigenvalues (all n, ascending) and the eigenvectors row_begin .. row_end-1 as ROWS of Zt [row_end-row_begin][ldz].
// Reduction and tridiagonal solve run in full; only the back-transformations are restricted to the requested
// eigenvectors, so R ranks holding the same A each pay 1/R of them (vivit_amd/distributed.py gathers the slices).
int vivit_symeig_rows_f32(float *A, int64_t n, int64_t lda, float *w, float *Zt, int64_t ldz, int64_t row_begin,
                          int64_t row_end, void *workspace, size_t workspace_bytes, int32_t *info, void *stream) {
  if (n < 0 || !info) return VIVIT_E_BADARG;
  if (!A || !w || !Zt || lda < n || ldz < n || row_begin < 0 || row_end < row_begin || row_end > n) return VIVIT_E_BADARG;
  if (n <= SMALL_N_MAX) return VIVIT_E_UNSUPPORTED;  // single-workgroup sizes: use vivit_symeig_f32 and slice
  return symeig_large_rows_launch(A, n, lda, w, Zt, ldz, row_begin, row_end, workspace, workspace_bytes, info,
                                  static_cast<hipStream_t>(stream));
}

size_t vivit_symeig_reduce_f32_workspace_bytes(int64_t n) {
  if (n <= SMALL_N_MAX) return 0;
  return symeig_reduce_workspace_bytes(n);
}

size_t vivit_symeig_select_f32_workspace_bytes(int64_t n, int64_t K) {
  if (n <= SMALL_N_MAX) return 0;
  return symeig_select_workspace_bytes(n, K);
}

// Phase 1 of the selected-eigenvector solver: reduction to tridiagonal form + ALL eigenvalues (ascending).
int vivit_symeig_reduce_f32(float *A, int64_t n, int64_t lda, float *w, void *state, size_t state_bytes,
                            int32_t *info, void *stream) {
  if (n < 0 || !info) return VIVIT_E_BADARG;
  if (!A || !w || lda < n) return VIVIT_E_BADARG;
  if (n <= SMALL_N_MAX) return VIVIT_E_UNSUPPORTED;  // single-workgroup sizes: use vivit_symeig_f32 and slice
  return symeig_reduce_launch(A, n, lda, w, state, state_bytes, info, static_cast<hipStream_t>(stream));
}

// Phase 2: the eigenvectors of the K eigenvalues at ascending positions idx[0] < idx[1] < ... as ROWS of Zt.
int vivit_symeig_select_f32(const float *A, int64_t n, int64_t lda, const int32_t *idx, int64_t K, float *Zt, int64_t ldz,
                            void *state, size_t state_bytes, void *workspace, size_t workspace_bytes, int32_t *info,
                            void *stream) {
  if (n < 0 || !info || !A || lda < n) return VIVIT_E_BADARG;
  if (n <= SMALL_N_MAX) return VIVIT_E_UNSUPPORTED;
  return symeig_select_launch(A, n, lda, idx, K, Zt, ldz, state, state_bytes, workspace, workspace_bytes, info,
                              static_cast<hipStream_t>(stream));
}

} // extern "C"
