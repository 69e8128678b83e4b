// Shared declarations for libvivit_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/vivit_hip.h"

namespace vivit {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline int launch_status() {
  return hipGetLastError() == hipSuccess ? VIVIT_OK : VIVIT_E_LAUNCH;
}

// Large dynamic-LDS kernels need hipFuncAttributeMaxDynamicSharedMemorySize once PER DEVICE (the attribute lives
// on the device's code object): `done` is a per-call-site bitmap over device ordinals.
static inline bool ensure_dynamic_lds(const void *fn, int bytes, unsigned long long &done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done & bit) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  return true;  // the caller sets the bit after ALL kernels of its group succeeded
}

// Operand storage of a GEMM input X that is logically [rows, k]:
//   LAY_K: X[row * ld + k]   (k contiguous)
//   LAY_M: X[k * ld + row]   (row contiguous)
enum { LAY_K = 0, LAY_M = 1 };

// Per-problem descriptor for the batched mode: lives in DEVICE memory and is written by
// device-side planning kernels (sizes such as the number of non-deflated eigenvalues of a
// divide-and-conquer merge are only known on the device), so batched launches need no host sync.
struct GemmDesc {
  const float *A;
  const float *B;
  float *C;
  int64_t M, N, K, lda, ldb, ldc;
};

struct GemmArgs {
  const float *A;
  const float *B;
  float *C;
  int64_t M, N, K, lda, ldb, ldc;
  float alpha, beta;
  int ksplit;       // >1: partial tiles go to `slab`, reduced by gemm_reduce_kernel
  int64_t kchunk;   // K range per split (multiple of the K tile)
  float *slab;      // [ksplit][M][N]
  int tiles_m, tiles_n;
  int syrk;         // 1: lower-triangular tiles only + mirrored store; 2: lower tiles, no mirror
  int a_vec, b_vec; // 16-byte loads legal for the operand
  const GemmDesc *desc;  // non-null: batched mode, problem blockIdx.z is desc[blockIdx.z]
  int sbw;          // super-block width in tiles (host and kernel must agree on the tile map)
  // gemm256_kernel only: non-null = the launch stands in for a bf16-pipe launch and runs iff (*gate & gate_mask) != 0
  const int *gate = nullptr;
  int gate_mask = 0;
};

// Generic launcher (gemm_f32.hip).  alay/blay in {LAY_K, LAY_M}.
int gemm_launch(int alay, int blay, const float *A, const float *B, float *C, int64_t M, int64_t N,
                int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float alpha, float beta, bool syrk,
                void *workspace, size_t workspace_bytes, hipStream_t stream);
size_t gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, bool syrk);
// SYR2K-style update: C(lower tiles) = alpha * A B^T + beta * C with A, B in LAY_M ([k][row]);
// the result must be symmetric (A B^T + B A^T form); the strict upper triangle outside the
// diagonal tiles is NOT touched.
int gemm_lower_launch(const float *A, const float *B, float *C, int64_t n, int64_t K, int64_t lda, int64_t ldb,
                      int64_t ldc, float alpha, float beta, hipStream_t stream);
// Batched C_b = alpha * op(A_b) op(B_b)^T + beta * C_b over `batch` device-resident descriptors;
// the grid is sized for (maxM, maxN), problems may be smaller (or empty).
int gemm_batched_launch(int alay, int blay, const GemmDesc *desc, int batch, int64_t maxM, int64_t maxN, float alpha,
                        float beta, hipStream_t stream);

// skinny.hip: out[K x P] = alpha * coef[K x n] @ V[n x P] + beta * out for K <= 16 (HBM-bound streaming)
bool skinny_applicable(int64_t M, int64_t N, int64_t K);
size_t skinny_workspace_bytes(int64_t K, int64_t n, int64_t P);
int skinny_nn_launch(const float *coef, int64_t ldc_, const float *V, int64_t ldv, float *out, int64_t ldo, int64_t K,
                     int64_t n, int64_t P, float alpha, float beta, void *ws, size_t ws_bytes, hipStream_t stream);

// profile.hip: optional event timing.  kind 0 = Gram SYRK (work = flops), 1 = symv (work = bytes).
bool prof_enabled();
int prof_stride();
void prof_begin(int kind, double work, hipStream_t stream);
void prof_end(int kind, hipStream_t stream);
// stage marks of the eigensolver (vivit_profile_stages): indices of out_ms
enum { PROF_STAGE_BEGIN = 0, PROF_STAGE_PREP = 1, PROF_STAGE_SY2SB = 2, PROF_STAGE_SB2ST = 3, PROF_STAGE_TRIDIAG = 4,
       PROF_STAGE_Q2 = 5, PROF_STAGE_Q1 = 6, PROF_STAGE_OUTPUT = 7, PROF_STAGE_SYTRD = 8,
       // parts of the band reduction that run on different pipes (the rest of it stays under PROF_STAGE_SY2SB):
       // 9 the streaming panel product P^T = V^T A22 (fp32 MFMA), 10 the delayed trailing updates (bf16 pipe)
       PROF_STAGE_SY2SB_PP = 9, PROF_STAGE_SY2SB_UPD = 10, PROF_NUM_STAGES = 11 };
void prof_mark(int stage, hipStream_t stream);

// profile.hip: the persistent kernels' failure word of (CURRENT device, `stream`) -- device memory, no allocation: an entry
// of a __device__ array.  A persistent kernel that gave up (co-residency not reached in two attempts, or a stalled
// exchange) ORs its PERSIST_TMO_* bit into it; the next info finalisation of an eigensolver call ON THE SAME STREAM reports
// VIVIT_INFO_PERSIST_TIMEOUT and clears it.  nullptr on failure of the symbol lookup.
int *persist_timeout_word(hipStream_t stream);
// 1 / 0: persistent kernels allowed by vivit_persistent_kernels(); -1: no override (the environment variables decide)
int persist_override();
// VIVIT_PERSIST_FAULT (tests): bit a set = attempt a of every persistent kernel's arrival gate gives up at once
int persist_fault();

// Compute units of the current device (cached per device): the one-XCD persistent kernels (sytrd_persist.hip, the panel
// QR of sy2sb.hip) need 32 co-resident workgroups of one per CU on XCD 0, i.e. an 8-XCD part with all 256 CUs visible.
inline int device_cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  int &c = cached[dev & 63];
  if (c == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = -1;
    c = v > 0 ? v : -1;
  }
  return c;
}

} // namespace vivit
