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

// Operand storage of a GEMM input X that is logically [rows, k]:
//   LAY_K: X[row * ld + k]   (k contiguous)
//   LAY_M: X[k * ld + row]   (row contiguous)
enum { LAY_K = 0, LAY_M = 1 };

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
  int syrk;         // 1: B == A, lower-triangular tiles only, mirrored store
  int a_vec, b_vec; // 16-byte loads legal for the operand
};

// Generic launcher (gemm_f32.hip).  alay/blay in {LAY_K, LAY_M}.
int gemm_launch(int alay, int blay, const float *A, const float *B, float *C, int64_t M, int64_t N,
                int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float alpha, float beta, bool syrk,
                void *workspace, size_t workspace_bytes, hipStream_t stream);
size_t gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, bool syrk);

} // namespace vivit
