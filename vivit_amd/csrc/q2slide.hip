// Back-transformation through the bulge-chasing reflectors, sliding-window form on the bf16 matrix pipe:
//   Zt <- Zt * Q2^T     (same operation as q2apply.hip; rows of Zt = eigenvectors)
//
// q2apply.hip applies one super-block (two levels of one sweep group: 192 window columns) per workgroup and launch
// step, so every row of Zt is read and written once per super-block: 6.4 TB at n = 40 960, a 1.5 s memory floor under a
// 1.26 s fp32-MFMA bound.  Here a workgroup OWNS a slab of rows for the whole transformation and walks the blocks in the
// topological order "level pair K outer, group inner" (q2_slide_proto.py):
//     pass K:  for g = gmax(K) .. 0:   block (g, 2K), block (g, 2K + 1)
// Block (g, k) touches the 64-column units g + k and g + k + 1, so inside a pass the three-unit window slides LEFT by one
// unit per group: per two blocks one unit is loaded and one is stored -- a third of the traffic (2.1 TB) -- and the slab
// never leaves the registers between blocks.  The products run on v_mfma_f32_16x16x32_bf16 with the exact three-way bf16
// split of both operands and six partial products (the arithmetic of gemm_f32.hip:gemm256_bx_kernel; chains are at most
// 128 long here): a wave owns 16 rows, lane (n16, kq) keeps S[row n16][16 q + 4 kq .. + 3] as float4 q = 0..11, and per
// block
//     W2^T = (T V) S^T     (A operand: T V tile from LDS,  B operand: the slab, split in registers)
//     U^T  = V^T W2^T      (A operand: V^T tile from LDS,  B operand: the accumulator of W2^T, split in registers)
//     S   -= U
// where, as in q2_apply16_kernel, the accumulator layout of one product is the B operand layout of the next once the
// k index of a 32-deep MFMA step is taken in the order  k(kq, j) = 32 step + 16 (j >> 2) + 4 kq + (j & 3)  by BOTH
// operands.  The A operands of a block are prepared once per solve (qs_prepare_kernel) as an IMAGE of 78 fragments of
// 1 KB -- [lane][8 bf16] exactly as the MFMA wants them, in consumption order, three pieces each -- so a block's image is
// 78 global -> LDS DMA instructions (no VALU, no registers) into one of two LDS buffers while the previous block
// computes, and an A operand is ONE conflict-free ds_read_b128.  Structurally zero 16 x 32 tiles of the parallelogram V
// and of the trapezoid T V are skipped: 26 tile steps x 6 = 156 MFMAs per block and wave.  Two loader waves per workgroup
// request the images; the compute waves (up to ten, 16 rows each) never wait for a request of their own.  Slabs are
// independent, so there is no synchronisation between workgroups at all: any number of them may be resident.
// Measured at n = 40 960: 1.15 - 1.2 s against 1.66 s for q2apply.hip, matrix pipe busy 56 %.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "device_utils.h"
#include "eig_internal.h"

namespace vivit {

// QS_VAR: timing-only builds (WRONG results; scripts/probe/q2_variants.sh builds them as separate libraries) that leave one
// ingredient out -- 1 no arithmetic (images, barriers and Zt traffic only), 2 no image requests by the compute waves, 3 no
// image requests and no barriers, 4 no Zt loads / stores, 5 no split arithmetic, 6 no LDS fragment reads, 7 = 3 + 4,
// 8 = 5 + 6 + 7, 9 no Zt loads, 10 no Zt stores.  (Mind what hipcc removes with them: without the final store of a unit
// the MFMAs that only feed it go as well.)
#ifndef QS_VAR
#define QS_VAR 0
#endif
#define QS_NO_SYNC (QS_VAR == 3 || QS_VAR == 7 || QS_VAR == 8)
#define QS_NO_ZT (QS_VAR == 4 || QS_VAR == 7 || QS_VAR == 8)
#if QS_VAR == 9
#define QS_NO_ZLOAD 1
#elif QS_VAR == 10
#define QS_NO_ZSTORE 1
#endif

typedef __bf16 qbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 qbf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned qu32x4 __attribute__((ext_vector_type(4)));

constexpr int QS_B = 64;     // reflector length (half bandwidth)
constexpr int QS_W = 64;     // sweeps per block
constexpr int QS_WIN = 128;  // window columns of a block

// tile steps with a non-zero A operand.  W2^T = (T V) S^T: (T V)[t'][w] != 0 needs w >= t' + 1 (window column w holds
// reflector component w - 1);  U^T = V^T W2^T: V[t][w] != 0 needs w - 64 <= t <= w - 1.
__host__ __device__ constexpr bool qs_w2_need(int ks, int ta) { return 32 * ks + 31 >= 16 * ta + 1; }
__host__ __device__ constexpr bool qs_u_need(int wt, int kt) { return 32 * kt <= 16 * wt + 14 && 32 * kt + 31 >= 16 * wt - 64; }
__host__ __device__ constexpr int qs_count_w2() {
  int c = 0;
  for (int ks = 0; ks < 4; ++ks)
    for (int ta = 0; ta < 4; ++ta) c += qs_w2_need(ks, ta) ? 1 : 0;
  return c;
}
__host__ __device__ constexpr int qs_count_u() {
  int c = 0;
  for (int wt = 0; wt < 8; ++wt)
    for (int kt = 0; kt < 2; ++kt) c += qs_u_need(wt, kt) ? 1 : 0;
  return c;
}
constexpr int QS_NW2 = qs_count_w2(), QS_NU = qs_count_u();
static_assert(QS_NW2 == 14 && QS_NU == 12, "fragment lists");
constexpr int QS_NFRAG = 3 * (QS_NW2 + QS_NU);  // 78
constexpr int QS_IMG = QS_NFRAG * 1024;        // bytes per block image

// k index of element j of the lanes kq in MFMA step `step` (both operands)
__host__ __device__ constexpr int qs_kcol(int step, int kq, int j) { return 32 * step + 16 * (j >> 2) + 4 * kq + (j & 3); }

// exact three-way split of two fp32 values into packed bf16 pairs (low half = first value)
// a - b as ONE v_sub_f32: hipcc pairs the residual subtractions into v_pk_add_f32, which costs ~13 cycles more than two
// plain subtractions beside MFMAs (measured: -1.3 % of the kernel's time)
__device__ __forceinline__ float qs_sub(float a, float b) {
  float r;
  __asm__("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ void qs_split2(float a, float b, unsigned &hi, unsigned &mid, unsigned &lo) {
  const qbf16x2 h = {(__bf16)a, (__bf16)b};
  hi = __builtin_bit_cast(unsigned, h);
  const float ra = qs_sub(a, __uint_as_float(hi << 16)), rb = qs_sub(b, __uint_as_float(hi & 0xffff0000u));
  const qbf16x2 m = {(__bf16)ra, (__bf16)rb};
  mid = __builtin_bit_cast(unsigned, m);
  const float sa = qs_sub(ra, __uint_as_float(mid << 16)), sb = qs_sub(rb, __uint_as_float(mid & 0xffff0000u));
  const qbf16x2 l = {(__bf16)sa, (__bf16)sb};
  lo = __builtin_bit_cast(unsigned, l);
}

// The same split for EIGHT values (one MFMA operand: four bf16 pairs per piece) as ONE asm block: 44 vector instructions,
// 5.5 per value.  Why asm: the apply kernel is bound by vector ISSUE (a 16x16x32 MFMA holds the SIMD's issue port 8 of its
// 16 cycles, every other vector instruction 4, three waves per SIMD), and hipcc spends 6.5 instructions per value on the
// C++ form above -- it folds `packed << 16` back into a scalar conversion of `a` and emits a second v_cvt_pk_bf16_f32 for it
// (5 conversions per pair instead of 3).  With the conversion alone as asm the extra conversions go, but hipcc then
// serialises each pair and pads every dependent asm -> asm step with an s_nop (16 per eight values, an issue slot each).
// In one block two pairs at a time are interleaved (no instruction uses the result of its predecessor) and nothing is
// padded; the closing s_nop 1 is the two wait states hipcc itself puts between a vector instruction that writes an MFMA
// operand and the MFMA (it cannot see the writes inside the asm).
#define QS_SPLIT_PAIRS(H0, H1, M0, M1, L0, L1, A0, B0, A1, B1)   \
  "v_cvt_pk_bf16_f32 " H0 ", " A0 ", " B0 "\n\t"                 \
  "v_cvt_pk_bf16_f32 " H1 ", " A1 ", " B1 "\n\t"                 \
  "v_lshlrev_b32 %12, 16, " H0 "\n\t"                            \
  "v_lshlrev_b32 %14, 16, " H1 "\n\t"                            \
  "v_and_b32 %13, 0xffff0000, " H0 "\n\t"                        \
  "v_and_b32 %15, 0xffff0000, " H1 "\n\t"                        \
  "v_sub_f32 %12, " A0 ", %12\n\t"                               \
  "v_sub_f32 %14, " A1 ", %14\n\t"                               \
  "v_sub_f32 %13, " B0 ", %13\n\t"                               \
  "v_sub_f32 %15, " B1 ", %15\n\t"                               \
  "v_cvt_pk_bf16_f32 " M0 ", %12, %13\n\t"                       \
  "v_cvt_pk_bf16_f32 " M1 ", %14, %15\n\t"                       \
  "v_lshlrev_b32 " L0 ", 16, " M0 "\n\t"                         \
  "v_lshlrev_b32 " L1 ", 16, " M1 "\n\t"                         \
  "v_sub_f32 %12, %12, " L0 "\n\t"                               \
  "v_sub_f32 %14, %14, " L1 "\n\t"                               \
  "v_and_b32 " L0 ", 0xffff0000, " M0 "\n\t"                     \
  "v_and_b32 " L1 ", 0xffff0000, " M1 "\n\t"                     \
  "v_sub_f32 %13, %13, " L0 "\n\t"                               \
  "v_sub_f32 %15, %15, " L1 "\n\t"                               \
  "v_cvt_pk_bf16_f32 " L0 ", %12, %13\n\t"                       \
  "v_cvt_pk_bf16_f32 " L1 ", %14, %15\n\t"

__device__ __forceinline__ void qs_split8_asm(const float4 x, const float4 y, qu32x4 &h, qu32x4 &m, qu32x4 &l) {
  unsigned h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
  float t0, t1, t2, t3;
  // (volatile: the two splits that the apply kernel hoists in front of a workgroup barrier must stay there)
  __asm__ volatile(QS_SPLIT_PAIRS("%0", "%1", "%4", "%5", "%8", "%9", "%16", "%17", "%18", "%19")
          QS_SPLIT_PAIRS("%2", "%3", "%6", "%7", "%10", "%11", "%20", "%21", "%22", "%23")
          "s_nop 1"
          : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3), "=&v"(l0), "=&v"(l1),
            "=&v"(l2), "=&v"(l3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
          : "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w), "v"(y.x), "v"(y.y), "v"(y.z), "v"(y.w));
  h[0] = h0; h[1] = h1; h[2] = h2; h[3] = h3;
  m[0] = m0; m[1] = m1; m[2] = m2; m[3] = m3;
  l[0] = l0; l[1] = l1; l[2] = l2; l[3] = l3;
}

struct QsPieces {
  qbf16x8 h, m, l;
};

// eight fp32 values (element j of an MFMA operand: x = j 0..3, y = j 4..7) -> three bf16x8 operands
__device__ __forceinline__ QsPieces qs_split8(const float4 x, const float4 y) {
#if QS_VAR == 5 || QS_VAR == 8   // timing only: no split arithmetic
  {
    QsPieces p;
    qu32x4 a = {__float_as_uint(x.x), __float_as_uint(x.y), __float_as_uint(x.z), __float_as_uint(x.w)};
    qu32x4 b = {__float_as_uint(y.x), __float_as_uint(y.y), __float_as_uint(y.z), __float_as_uint(y.w)};
    p.h = __builtin_bit_cast(qbf16x8, a);
    p.m = __builtin_bit_cast(qbf16x8, b);
    p.l = p.h;
    return p;
  }
#endif
  qu32x4 h, m, l;
  qs_split8_asm(x, y, h, m, l);
  QsPieces p;
  p.h = __builtin_bit_cast(qbf16x8, h);
  p.m = __builtin_bit_cast(qbf16x8, m);
  p.l = __builtin_bit_cast(qbf16x8, l);
  return p;
}

// pass K = levels 2K, 2K + 1 of the groups gmax(K) = G0 - 2K .. 0, G0 = (n - 2) / 64.  Blocks of a pass in walk order:
// (g, 2K), (g, 2K + 1) for g = gmax .. 0: 2 (gmax + 1) blocks.  Level 2K + 1 does not exist for g = gmax: that block is
// kept as an IDENTITY block (no reflector: all-zero image) so that every step of the walk has the same two blocks and the
// kernel's loop has no branch between them.  Blocks in front of pass K counted from pass K0:
__host__ __device__ inline int64_t qs_pass_offset(int G0, int K0, int K) {
  const int64_t d = K - K0;
  return d * (2 * (int64_t)G0 + 2) - 4 * d * K0 - 2 * d * (d - 1);
}

struct QsPrep {
  const float *R2;
  int64_t ldr;
  const float *tau2;
  int nk, n, G0, K0;
  unsigned char *img;  // images of the passes K0 .., walk order
};

// ---- image of one block: T factor, T V, and the 78 operand fragments ------------------------------------------------
constexpr int QS_LDV = QS_WIN + 1;  // 129: odd row stride, column walks are conflict-free
constexpr int QS_LDT = QS_W + 1;
constexpr int QS_PREP_LDS = (2 * QS_W * QS_LDV + 2 * QS_W * QS_LDT + QS_W) * 4;

__global__ __launch_bounds__(256) void qs_prepare_kernel(QsPrep a) {
  extern __shared__ __attribute__((aligned(16))) float qsp_lds[];
  float *Vw = qsp_lds;                 // [64][129]  Vw[t][w] = v_t[w - 1]
  float *TV = Vw + QS_W * QS_LDV;      // [64][129]
  float *S = TV + QS_W * QS_LDV;       // [64][65]
  float *Ts = S + QS_W * QS_LDT;       // [64][65]
  float *taus = Ts + QS_W * QS_LDT;    // [64]
  const int tid = threadIdx.x;
  const int K = a.K0 + blockIdx.y;
  const int gmax = a.G0 - 2 * K;
  if (gmax < 0 || (int)blockIdx.x > 2 * gmax + 1) return;
  const int g = gmax - (int)blockIdx.x / 2, k = 2 * K + ((int)blockIdx.x & 1);   // (gmax, 2K + 1): no reflector is live
  unsigned char *img = a.img + (qs_pass_offset(a.G0, a.K0, K) + blockIdx.x) * (int64_t)QS_IMG;
  const int g0 = g * QS_W;
  const int c_start = g0 + 1 + k * QS_B;
  // reflectors: all loads of a thread in flight, masked afterwards
  {
    constexpr int NV = QS_W * QS_WIN / 256;  // 32
    float vv[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + 256 * u;
      const int t = idx / QS_WIN, w = idx - t * QS_WIN;
      const int s = g0 + t, c0 = c_start + t;
      const bool live = s <= a.n - 3 && c0 < a.n;
      const int L = (a.n - c0) < QS_B ? (a.n - c0) : QS_B;
      const int i = w - 1;
      const bool in = live && i >= t && i < t + L;
      vv[u] = a.R2[in ? (int64_t)s * a.ldr + c_start + i : 0];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + 256 * u;
      const int t = idx / QS_WIN, w = idx - t * QS_WIN;
      const int s = g0 + t, c0 = c_start + t;
      const bool live = s <= a.n - 3 && c0 < a.n;
      const int L = (a.n - c0) < QS_B ? (a.n - c0) : QS_B;
      const int i = w - 1;
      const bool in = live && i >= t && i < t + L;
      Vw[t * QS_LDV + w] = in ? vv[u] : 0.f;
    }
  }
  if (tid < QS_W) {
    const int s = g0 + tid;
    taus[tid] = (s <= a.n - 3 && c_start + tid < a.n) ? a.tau2[(int64_t)s * a.nk + k] : 0.f;
  }
  __syncthreads();
  {  // S = V V^T (64 x 64), 16 entries per thread; rows r and c overlap in w in [max(r, c) + 1, min(r, c) + 64]
    const int r = tid >> 2, cb = (tid & 3) * 16;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    for (int w = r + 1; w <= r + QS_B; ++w) {
      const float vr = Vw[r * QS_LDV + w];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] += vr * Vw[(cb + c) * QS_LDV + w];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) S[r * QS_LDT + cb + c] = acc[c];
  }
  __syncthreads();
  // T factor (device_utils.h:tfactor_column gives the same matrix with one thread per column: a dependent chain of 2016
  // multiply-adds for the last column, ~100 us per block and most of this kernel).  Blocked instead: the four 16 x 16
  // diagonal blocks by the column recurrence (chains of <= 120), then the merge T12 = -T1 (S12 T2) of compact-WY factors
  // for the block pairs of size 16 and 32, every product spread over the 256 threads.  X: scratch in the T V array.
  {
    float *X = TV;
    for (int idx = tid; idx < QS_W * QS_LDT; idx += 256) Ts[idx] = 0.f;
    __syncthreads();
    if (tid < QS_W) tfactor_column(S + (tid & ~15) * (QS_LDT + 1), taus + (tid & ~15), Ts + (tid & ~15) * (QS_LDT + 1), QS_LDT, 16, tid & 15);
    __syncthreads();
    for (int hb = 16; hb < QS_W; hb *= 2) {
      const int npair = QS_W / (2 * hb), hh = hb * hb;
      for (int idx = tid; idx < npair * hh; idx += 256) {   // X = S12 T2 (T2 upper triangular: k <= c)
        const int pr = idx / hh, rem = idx - pr * hh, r = rem / hb, c = rem - r * hb, o = 2 * hb * pr;
        float acc = 0.f;
        for (int k = 0; k <= c; ++k) acc += S[(o + r) * QS_LDT + o + hb + k] * Ts[(o + hb + k) * QS_LDT + o + hb + c];
        X[idx] = acc;
      }
      __syncthreads();
      for (int idx = tid; idx < npair * hh; idx += 256) {   // T12 = -T1 X (T1 upper triangular: k >= r)
        const int pr = idx / hh, rem = idx - pr * hh, r = rem / hb, c = rem - r * hb, o = 2 * hb * pr;
        float acc = 0.f;
        for (int k = r; k < hb; ++k) acc += Ts[(o + r) * QS_LDT + o + k] * X[pr * hh + k * hb + c];
        Ts[(o + r) * QS_LDT + o + hb + c] = -acc;
      }
      __syncthreads();
    }
  }
  // T V (T upper triangular): (S V^T) T^T = S (T V)^T, one product instead of two in the apply kernel.  Thread = window
  // column w and half of the rows t': V[t][w] is read once per t, T[t'][t] is the same word for the whole wave
  {
    const int w = tid & (QS_WIN - 1), tp0 = (tid >> 7) * 32;
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = 0.f;
    // V[t][w] != 0 only for t in [w - 64, w - 1]; rows t' <= t
    const int tlo = w - QS_B > tp0 ? w - QS_B : tp0, thi = w - 1 < QS_W - 1 ? w - 1 : QS_W - 1;
    for (int t = tlo; t <= thi; ++t) {
      const float v = Vw[t * QS_LDV + w];
#pragma unroll
      for (int j = 0; j < 32; ++j) acc[j] += Ts[(tp0 + j) * QS_LDT + t] * v;   // (zero below the diagonal)
    }
    __syncthreads();   // (X aliased the T V array)
#pragma unroll
    for (int j = 0; j < 32; ++j) TV[(tp0 + j) * QS_LDV + w] = acc[j];
  }
  __syncthreads();
  // fragments: (fragment triple ft, lane) items; a thread writes the three 16-byte pieces of its lane
  for (int idx = tid; idx < (QS_NW2 + QS_NU) * 64; idx += 256) {
    const int ft = idx >> 6, ln = idx & 63;
    float v[8];
    {
      const int m16 = ln & 15, kq = ln >> 4;
      if (ft < QS_NW2) {
        // walk order of the W2 steps: ks outer, ta inner over the needed tiles: ks = 0: ta 0, 1; ks >= 1: ta 0..3
        const int ks = ft < 2 ? 0 : 1 + (ft - 2) / 4, ta = ft < 2 ? ft : (ft - 2) & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = TV[(16 * ta + m16) * QS_LDV + qs_kcol(ks, kq, j)];
      } else {
        // walk order of the U steps: wt outer, kt inner: (0,0) (1,0) (2,0) (2,1) ... (5,0) (5,1) (6,1) (7,1)
        const int fu = ft - QS_NW2;
        const int wt = fu < 2 ? fu : (fu < 10 ? 2 + (fu - 2) / 2 : fu - 4), kt = fu < 2 ? 0 : (fu < 10 ? (fu - 2) & 1 : 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = Vw[qs_kcol(kt, kq, j) * QS_LDV + 16 * wt + m16];
      }
    }
    qu32x4 h, m, l;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      unsigned x, y, z;
      qs_split2(v[2 * jj], v[2 * jj + 1], x, y, z);
      h[jj] = x; m[jj] = y; l[jj] = z;
    }
    unsigned char *dst = img + (int64_t)(3 * ft) * 1024 + ln * 16;
    *reinterpret_cast<qu32x4 *>(dst) = h;
    *reinterpret_cast<qu32x4 *>(dst + 1024) = m;
    *reinterpret_cast<qu32x4 *>(dst + 2048) = l;
  }
}

// ---- apply ---------------------------------------------------------------------------------------------------------
struct QsArgs {
  const unsigned char *img;  // images of the passes K0 .. K1 - 1 in walk order
  float *Zt;
  int64_t ldz;
  int nrows, n, G0, K0, K1;
  int nload;   // loader waves behind the compute waves of a workgroup (0: the compute waves request the images themselves)
  // per-XCD progress board (round 6, VIVIT_Q2_LOCKSTEP = window in blocks, 0 = off): [8 XCDs][64 ints] = {arrival counter, ..,
  // [32 + slot]: the block a workgroup's loader is about to request + 1 (0: not started, INT_MAX: done)}.  A loader that is more
  // than `window` blocks ahead of the slowest running workgroup of ITS XCD waits (bounded) before requesting: the 32 CUs of an
  // XCD then pull a block image while it is still in their 4 MiB L2 instead of once per CU from the fabric.
  int *board;
  int window;
};

// six partial products, smallest first (A pieces ah/am/al, B pieces bh/bm/bl)
#define QS_MFMA6(acc, ah, am, al, bp)                                                   \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, (bp).h, acc, 0, 0, 0);              \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, (bp).l, acc, 0, 0, 0);              \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, (bp).m, acc, 0, 0, 0);              \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, (bp).h, acc, 0, 0, 0);              \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, (bp).m, acc, 0, 0, 0);              \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, (bp).h, acc, 0, 0, 0);

// one block on the float4s Q0 .. Q0 + 7 of the window; frag: this lane's 16 bytes of fragment 0 of the block's image
typedef const __attribute__((address_space(3))) unsigned char *qs_lds_ptr;
typedef const __attribute__((address_space(3))) qbf16x8 *qs_lds_frag;


// the 26 tile steps of a block in walk order: steps 0..13 form W2^T (ks outer, t'-tile inner), 14..25 form U^T (w-tile
// outer, kt inner).  Compile-time tables (every index below is a constant after unrolling).
struct QsStep {
  int ks, ta;   // W2 step: k step of the window, t' tile
  int wt, kt;   // U step: w tile, k step over t'
};
__host__ __device__ constexpr QsStep qs_step(int i) {
  int c = 0;
  for (int ks = 0; ks < 4; ++ks)
    for (int ta = 0; ta < 4; ++ta)
      if (qs_w2_need(ks, ta)) {
        if (c == i) return QsStep{ks, ta, -1, -1};
        ++c;
      }
  for (int wt = 0; wt < 8; ++wt)
    for (int kt = 0; kt < 2; ++kt)
      if (qs_u_need(wt, kt)) {
        if (c == i) return QsStep{-1, -1, wt, kt};
        ++c;
      }
  return QsStep{-1, -1, -1, -1};
}
constexpr int QS_NSTEP = QS_NW2 + QS_NU;  // 26

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <int N, int I = 0, class F>
__device__ __forceinline__ void qs_static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    qs_static_for<N, I + 1>(f);
  }
}

struct QsFrag {
  qbf16x8 h, m, l;
};
__device__ __forceinline__ QsFrag qs_frag(qs_lds_ptr frag, int i) {
  QsFrag f;
#if QS_VAR == 6 || QS_VAR == 8   // timing only: no LDS reads
  {
    qu32x4 a = {(unsigned)(uintptr_t)frag, 0x3f803f80u, (unsigned)i, 0x3f803f80u};
    f.h = __builtin_bit_cast(qbf16x8, a);
    f.m = f.h;
    f.l = f.h;
    return f;
  }
#endif
  f.h = *reinterpret_cast<qs_lds_frag>(frag + (3 * i + 0) * 1024);
  f.m = *reinterpret_cast<qs_lds_frag>(frag + (3 * i + 1) * 1024);
  f.l = *reinterpret_cast<qs_lds_frag>(frag + (3 * i + 2) * 1024);
  return f;
}

// One block on the float4s Q0 .. Q0 + 7 of the window; frag: this lane's 16 bytes of fragment 0 of the block's image.
// The A fragments of step i + 1 are read from LDS before the six MFMAs of step i are issued (a wave then waits for an LDS
// read only when the matrix pipe is already ahead of it); `mid(i)` is called once per step between the reads and the MFMAs
// (the caller spreads its image DMA requests over the block with it).  `bp0`: the split of the block's first B operand
// (sw[Q0], sw[Q0 + 1]), which the caller computes BEFORE the workgroup barrier in front of the block: the block then opens
// with MFMAs (the first ~250 cycles behind a barrier release are the expensive place for vector instructions -- MI355X guide,
// "start-of-segment VALU penalty" -- and a wave that reaches the barrier early splits while it would otherwise wait).
template <int Q0, class Mid>
__device__ __forceinline__ void qs_apply_block(float4 (&sw)[12], qs_lds_ptr frag, Mid mid, const QsPieces &bp0) {
#if QS_VAR == 1
  __asm__ volatile("" : "+v"(sw[Q0].x), "+v"(sw[Q0 + 7].w));
  (void)bp0;
  qs_static_for<QS_NSTEP>([&](auto itag) __attribute__((always_inline)) { mid(decltype(itag)::value); });
  return;
#endif
  f32x4 acc2[4];
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc2[ta][e] = 0.f;
  QsPieces bp, wp[2];
  f32x4 u;
  QsFrag cur = qs_frag(frag, 0);
  qs_static_for<QS_NSTEP>([&](auto itag) __attribute__((always_inline)) {
    constexpr int i = decltype(itag)::value;
    constexpr QsStep st = qs_step(i), prev = qs_step(i > 0 ? i - 1 : 0), next = qs_step(i + 1 < QS_NSTEP ? i + 1 : i);
    QsFrag nxt = cur;
    if constexpr (i + 1 < QS_NSTEP) nxt = qs_frag(frag, i + 1);
    mid(i);
    if constexpr (st.ks >= 0) {
      if constexpr (i == 0) bp = bp0;
      else if constexpr (st.ks != prev.ks) bp = qs_split8(sw[Q0 + 2 * st.ks], sw[Q0 + 2 * st.ks + 1]);
#if defined(QS_PRIO)
      __builtin_amdgcn_s_setprio(1);
#endif
      QS_MFMA6(acc2[st.ta], cur.h, cur.m, cur.l, bp)
#if defined(QS_PRIO)
      __builtin_amdgcn_s_setprio(0);
#endif
    } else {
      if constexpr (i == QS_NW2) {
        wp[0] = qs_split8(make_float4(acc2[0][0], acc2[0][1], acc2[0][2], acc2[0][3]),
                          make_float4(acc2[1][0], acc2[1][1], acc2[1][2], acc2[1][3]));
        wp[1] = qs_split8(make_float4(acc2[2][0], acc2[2][1], acc2[2][2], acc2[2][3]),
                          make_float4(acc2[3][0], acc2[3][1], acc2[3][2], acc2[3][3]));
      }
      if constexpr (prev.wt != st.wt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = 0.f;
      }
#if defined(QS_PRIO)
      __builtin_amdgcn_s_setprio(1);
#endif
      QS_MFMA6(u, cur.h, cur.m, cur.l, wp[st.kt])
#if defined(QS_PRIO)
      __builtin_amdgcn_s_setprio(0);
#endif
      if constexpr (i + 1 == QS_NSTEP || next.wt != st.wt) {
        float4 &x = sw[Q0 + st.wt];
        x.x -= u[0]; x.y -= u[1]; x.z -= u[2]; x.w -= u[3];
      }
    }
    cur = nxt;
    // nothing moves across a step boundary: the scheduler otherwise hoists the fragment reads of several steps and the
    // next split to the front (three more registers than the 168 that ten to twelve waves per workgroup allow)
    __builtin_amdgcn_sched_barrier(0);
  });
}

// MAXW: most waves per workgroup of the instantiation (register budget 512 / ceil(MAXW / 4) per lane)
template <int MAXW>
__global__ __launch_bounds__(64 * MAXW) void qs_apply_kernel(QsArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char qs_lds[];  // two images
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwt = (int)(blockDim.x >> 6);
  const int nw = nwt - a.nload;     // compute waves (16 rows each)
  const int n16 = lane & 15, kq = lane >> 4;
  const int64_t nseq = qs_pass_offset(a.G0, a.K0, a.K1);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)qs_lds);
  const unsigned lane16 = (unsigned)lane * 16u;
  if (wave >= nw) {
    // ---- loader wave: requests its share of every image, one barrier per block as the compute waves.  A global -> LDS
    // request costs the issuing wave 60 - 180 cycles (MI355X guide: LDS-DMA piece issue cost); 78 of them per block spread
    // over the compute waves were ~8 % of their time.  Two loader waves sit on the two SIMDs that carry two compute waves
    // when ten compute waves share four SIMDs.
    const int lw = wave - nw, nl = a.nload;
    // (The asm below writes M0.  clang reserves M0 in inline asm -- it cannot be named as a clobber ("inline asm clobber list
    // contains reserved registers: m0 ... may not be preserved") -- and hipcc writes M0 itself only for its own LDS-DMA
    // builtins, s_movrel and s_sendmsg, none of which occurs in this kernel: every use of M0 here is set up by the same asm
    // statement that consumes it.)
    // Every workgroup starts its share at another piece: the CUs of an XCD otherwise ask the L2 for the same line at the same
    // time (the image stream alone, without arithmetic, runs 6-11 % faster rotated; the complete kernel 0.5-1 %:
    // profiles/r05_q2_experiments.md).  The pieces land at their own LDS addresses whatever the order.
    const int cnt = (QS_NFRAG - lw + nl - 1) / nl;
    const int rot = (int)(blockIdx.x % (unsigned)cnt);
    auto request = [&](int64_t seq) __attribute__((always_inline)) {
      for (int i = 0; i < cnt; ++i) {
        int ii = i + rot;
        if (ii >= cnt) ii -= cnt;
        const int f = lw + nl * ii;
        const unsigned char *src = a.img + seq * (int64_t)QS_IMG + (int64_t)f * 1024;
        const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(seq & 1) * (unsigned)QS_IMG + (unsigned)f * 1024u);
        __asm__ volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(d), "v"(lane16), "s"(src) : "memory");
      }
    };
    // lock-step of an XCD's loaders (loader wave 0 of every workgroup; see QsArgs::board)
    const bool pace = a.window > 0 && a.board != nullptr && lw == 0;
    int *mine = nullptr, *peers = nullptr;
    if (pace) {
      int xcc;
      __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      int *xb = a.board + (xcc & 7) * 64;
      int slot = 0;
      if (lane == 0) slot = __hip_atomic_fetch_add(xb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 31;
      slot = __builtin_amdgcn_readfirstlane(slot);
      mine = xb + 32 + slot;
      peers = xb + 32 + (lane & 31);
    }
    if (nseq > 0) request(0);
    for (int64_t seq = 0; seq < nseq; ++seq) {
      int slowest = 0x7fffffff;
      if (pace) {
        // publish "about to request block seq + 1" and look at the peers -- the load is in flight while this wave waits for
        // its image and for the compute waves: nothing is added to the block's critical path
        if (lane == 0) __hip_atomic_store(mine, (int)seq + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int v = __hip_atomic_load(peers, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        slowest = v == 0 ? 0x7fffffff : v;
      }
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of image seq have landed
#if !QS_NO_SYNC
      __builtin_amdgcn_s_barrier();                           // ... everybody's; and every wave is done with image seq - 1
#endif
      if (pace) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          const int other = __shfl_xor(slowest, o, 64);
          slowest = other < slowest ? other : slowest;
        }
        // ahead of the slowest running workgroup of this XCD by more than the window: wait for it, at most ~16 us per block
        // (a peer that is descheduled or stuck must not stop this slab: the throttle is best effort)
        for (int tries = 0; tries < 16 && (int)seq + 2 - slowest > a.window; ++tries) {
          __builtin_amdgcn_s_sleep(100);
          const int v = __hip_atomic_load(peers, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          slowest = v == 0 ? 0x7fffffff : v;
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) {
            const int other = __shfl_xor(slowest, o, 64);
            slowest = other < slowest ? other : slowest;
          }
        }
      }
      if (seq + 1 < nseq) request(seq + 1);
    }
    if (pace && lane == 0) __hip_atomic_store(mine, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const bool self_load = a.nload == 0;
  // wave-uniform row base (scalar registers) + a 32-bit lane offset: one VGPR of address instead of a 64-bit pointer
  const int64_t row0 = (int64_t)blockIdx.x * (16 * nw) + wave * 16;
  const bool wave_valid = row0 < a.nrows;                  // some row of the wave exists
  const bool rok = row0 + n16 < a.nrows;
  float *zbase = a.Zt + (wave_valid ? row0 : 0) * a.ldz;   // rows beyond nrows read the wave's first row and are never stored
  const unsigned zoff = (unsigned)((rok ? n16 : 0) * a.ldz + 4 * kq);   // + 64 unit + 16 q
  const int n = a.n;
  qs_lds_ptr myfrag = (qs_lds_ptr)qs_lds + lane * 16;

  // image `seq` -> LDS buffer seq & 1: this wave's share of the 78 one-KB pieces is wave, wave + nw, ... (lane i's 16
  // bytes land at M0 + 16 i); piece j of the share is requested at tile step QS_DMA_AT(j) of the block that computes
  // while the image lands, so that the requests do not all queue up behind the barrier
  auto dma_piece = [&](int64_t seq, int j) __attribute__((always_inline)) {
    const int f = wave + nw * j;
    if (f < QS_NFRAG) {
      // scalar base + 32-bit lane offset: no per-piece address registers (eight 64-bit addresses kept across the block
      // were spilled by the 168-register instantiation)
      const unsigned char *src = a.img + seq * (int64_t)QS_IMG + (int64_t)f * 1024;
      const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(seq & 1) * (unsigned)QS_IMG + (unsigned)f * 1024u);
      __asm__ volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(d), "v"(lane16), "s"(src) : "memory");
    }
  };
  const int npiece = (QS_NFRAG + nw - 1) / nw;   // pieces per wave (the last one only for some waves)
  auto dma = [&](int64_t seq) __attribute__((always_inline)) {
    for (int j = 0; j < npiece; ++j) dma_piece(seq, j);
  };
  // pieces requested at tile step i of the block that computes while the image lands: piece i at steps 0 .. 12 (one
  // request per step: they do not queue up behind the barrier, and all of them are out in the first half of the block so
  // that the image has the second half to land), whatever is left (fewer than six waves) at step 0
  constexpr int QS_DMA_STEPS = QS_NSTEP / 2;   // 13
  auto dma_step = [&](int64_t seq, int i) __attribute__((always_inline)) {
    if (!self_load) return;
    if (i == 0)
      for (int j = QS_DMA_STEPS; j < npiece; ++j) dma_piece(seq, j);
    if (i < QS_DMA_STEPS) dma_piece(seq, i);
  };
  // unit u entirely inside the matrix (64 u + 63 < n): unguarded loads (rows beyond nrows read row 0, never stored)
  auto load_unit = [&](int u, float4 (&dst)[4]) __attribute__((always_inline)) {
    const float *src = zbase + (int64_t)64 * u;   // (uniform)
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const float4 *>(src + zoff + 16 * q);
  };
  // the last unit of the matrix (u = G0: columns up to n - 1 exist; n % 4 == 0, so a float4 is all in or all out)
  auto load_unit_edge = [&](int u, float4 (&dst)[4]) __attribute__((always_inline)) {
    const int c0 = 64 * u + 4 * kq;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = c0 + 16 * q < n;
      dst[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) dst[q] = *reinterpret_cast<const float4 *>(zbase + (int64_t)64 * u + zoff + 16 * q);
    }
  };
  auto store_unit = [&](int u, const float4 *src) __attribute__((always_inline)) {
    const int c0 = 64 * u + 4 * kq;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (rok && c0 + 16 * q < n) {
#if defined(QS_NT_STORE)
        __builtin_nontemporal_store(src[q].x, zbase + (int64_t)64 * u + zoff + 16 * q);
        __builtin_nontemporal_store(src[q].y, zbase + (int64_t)64 * u + zoff + 16 * q + 1);
        __builtin_nontemporal_store(src[q].z, zbase + (int64_t)64 * u + zoff + 16 * q + 2);
        __builtin_nontemporal_store(src[q].w, zbase + (int64_t)64 * u + zoff + 16 * q + 3);
#else
        *reinterpret_cast<float4 *>(zbase + (int64_t)64 * u + zoff + 16 * q) = src[q];
#endif
      }
  };

  // The Zt loads and stores are plain C++ (hipcc counts them), the image DMAs are asm (hipcc does not): a wait that hipcc
  // places for one of ITS loads also drains every DMA issued before that point.  The loop is arranged so that hipcc's
  // waits fall where that is wanted anyway, and the hand-placed waits count the operations that may stay in flight:
  //   * nothing of hipcc's is pending at the loop entry (compiler-visible vmcnt(0) behind the pass-start loads);
  //   * the next group's unit `pre` is requested in the MIDDLE of the first block, behind that block's image requests: it is
  //     the four youngest operations at the top of the second block (vmcnt(4) there waits for the image only) and has
  //     one and a half blocks to arrive;
  //   * `pre` is "used" by an empty asm at the end of the second block, in front of the stores: hipcc's wait for it also
  //     waits for the image requests of the second block, which the next first block needs in any case;
  //   * the stores of the retired unit need no wait; the wait at the top of the next first block leaves them in flight
  //     (vmcnt(4): they are the four youngest operations) when all four were issued (`steady`).
#define QS_USE4(a) "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)
  int64_t seq = 0;
  if (nseq > 0 && self_load) dma(0);
  float4 sw[12];
  for (int K = a.K0; K < a.K1; ++K) {
    const int gmax = a.G0 - 2 * K;
    {
      // every pass starts at the right edge: gmax + 2K = G0, the unit that holds column n - 1 (64 G0 <= n - 2); the two
      // units to its right lie outside the matrix (their V entries are zero, they are never stored)
      float4 t0[4];
      load_unit_edge(gmax + 2 * K, t0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        sw[q] = t0[q];
        sw[4 + q] = make_float4(0.f, 0.f, 0.f, 0.f);
        sw[8 + q] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible
      __asm__ volatile("" : QS_USE4(sw[0]), QS_USE4(sw[1]), QS_USE4(sw[2]), QS_USE4(sw[3]));
    }
    bool steady = false;   // the previous step issued exactly four stores behind the image requests
    QsPieces bpa = qs_split8(sw[0], sw[1]);   // first operand of block (gmax, 2K), in front of its barrier
    for (int g = gmax; g >= 0; --g) {
      // ---- block (g, 2K): its image has landed once every wave is past this wait and the barrier
      // (with loader waves a compute wave has no image request of its own to wait for: its stores stay in flight for two
      // blocks -- a store needs more than one block (5 us) to complete often enough to cost 0.3 s of 1.2 s when the wait at
      // the top of the second block covered it: timing-only builds without the stores / without the loads, 913 / 997 ms)
      if (self_load) {
        if (steady) __asm__ volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
#if !QS_NO_SYNC
      __builtin_amdgcn_s_barrier();
#endif
      const int64_t seq_a = seq + 1 < nseq ? seq + 1 : 0;   // (the very last block re-requests image 0: never read)
      // next group's left unit (g - 1 + 2K < G0: inside the matrix).  Unconditional (the last group of a pass fetches a
      // unit it does not use): a branch around the loads makes hipcc wait for them at the join
      float4 pre[4];
      const int up = g - 1 + 2 * K;
      qs_apply_block<0>(sw, myfrag + (seq & 1) * QS_IMG, [&](int i) __attribute__((always_inline)) {
#if QS_VAR != 2 && !QS_NO_SYNC
        dma_step(seq_a, i);
#endif
        if (i == QS_DMA_STEPS) {
#if QS_NO_ZT || defined(QS_NO_ZLOAD)
          for (int q = 0; q < 4; ++q) pre[q] = sw[8 + q];
#else
          load_unit(up > 0 ? up : 0, pre);
#endif
          __builtin_amdgcn_sched_barrier(0);   // (the loads stay here: behind the image requests, in front of the rest)
        }
      }, bpa);
      ++seq;
      {
        // ---- block (g, 2K + 1) (g = gmax: the identity block); its first operand is split in front of the barrier
        const QsPieces bpb = qs_split8(sw[4], sw[5]);
        if (self_load) {
#if QS_NO_ZT
          __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
          __asm__ volatile("s_waitcnt vmcnt(4)" ::: "memory");   // the image; `pre` stays in flight
#endif
        }
#if !QS_NO_SYNC
        __builtin_amdgcn_s_barrier();
#endif
        const int64_t seq_b = seq + 1 < nseq ? seq + 1 : 0;
        qs_apply_block<4>(sw, myfrag + (seq & 1) * QS_IMG, [&](int i) __attribute__((always_inline)) {
#if QS_VAR != 2 && !QS_NO_SYNC
          dma_step(seq_b, i);
#endif
        }, bpb);
        ++seq;
      }
      // hipcc waits for `pre` here (and with it for the image requests of the block just computed)
      __asm__ volatile("" : QS_USE4(pre[0]), QS_USE4(pre[1]), QS_USE4(pre[2]), QS_USE4(pre[3]));
      // the right unit is final: store it, slide the window
      const int ur = g + 2 * K + 2;
#if !QS_NO_ZT && !defined(QS_NO_ZSTORE)
      store_unit(ur, &sw[8]);
#endif
      steady = wave_valid && 64 * ur + 63 < n;
#pragma unroll
      for (int q = 0; q < 4; ++q) { sw[8 + q] = sw[4 + q]; sw[4 + q] = sw[q]; sw[q] = pre[q]; }
      bpa = qs_split8(sw[0], sw[1]);   // first operand of the next group's block (g - 1, 2K), in front of its barrier
    }
    // after g = 0: units 2K (slot 1) and 2K + 1 (slot 2) are still in registers
    store_unit(2 * K, &sw[4]);
    store_unit(2 * K + 1, &sw[8]);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // the next pass reads what this one stored
  }
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last image request
#undef QS_USE4
}

constexpr int QS_APPLY_LDS = 2 * QS_IMG;  // 159 744 bytes

// chunk of passes whose images fit `bytes`
static int qs_chunk_passes(int G0, int K0, int Kend, size_t bytes) {
  int K1 = K0;
  while (K1 < Kend && (size_t)qs_pass_offset(G0, K0, K1 + 1) * QS_IMG <= bytes) ++K1;
  return K1;
}

constexpr size_t QS_WS_TARGET = (size_t)2 << 30;   // images of ~2 GB per chunk of passes
// default window of the loaders' lock-step (QsArgs::board).  Measured at n = 40 960 (scripts/r06_q2_lockstep_pmc.sh,
// profiles/r06_q2_lockstep_pmc.log), window -> ms / FETCH_SIZE per two solves: off 1224.7 / 2.518e9 KiB; 2: 1200.6 / 1.201e9; 4: 1207.5 /
// 1.232e9; 8: 1210.2 / 1.289e9; 16: 1204.5 / 1.369e9; 128: 1266 -- half of the kernel's fabric traffic was block images fetched once
// per CU instead of once per XCD; the time follows only by 1.5-2 % (the kernel is bound by its MFMA chains and the L2 -> LDS stream).
#ifndef QS_LOCKSTEP_DEFAULT
#define QS_LOCKSTEP_DEFAULT 4
#endif

size_t q2_slide_workspace_bytes(int64_t n) {
  if (n < 3) return 0;
  const int G0 = (int)((n - 2) / 64);
  // all images, at most ~2 GB of them at a time, at least the first pass (the longest): 2 G0 + 2 blocks
  const size_t all = (size_t)qs_pass_offset(G0, 0, G0 / 2 + 1) * QS_IMG, one = (size_t)(2 * G0 + 2) * QS_IMG;
  const size_t cap = all < QS_WS_TARGET ? all : QS_WS_TARGET;
  return (one > cap ? one : cap) + 2048 + 4096;   // (+ the loaders' progress board)
}

static int qs_env(const char *name, int dflt) {
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}

// the sliding-window form pays when every CU gets a slab of a few waves: rows >= QS_MIN_ROWS (VIVIT_Q2_SLIDE_MIN_ROWS)
// (shape part of the decision: what the workspace queries can know)
bool q2_slide_possible(int64_t nrows, int64_t n) {
  static int on = -1, min_rows = 0;
  if (on < 0) {
    on = qs_env("VIVIT_Q2_SLIDE", 1);
    min_rows = qs_env("VIVIT_Q2_SLIDE_MIN_ROWS", 14336);
  }
  return on != 0 && n % 4 == 0 && n >= 192 && nrows >= min_rows;
}
bool q2_slide_ok(int64_t nrows, int64_t n, const float *Zt, int64_t ldz) {
  const bool vec = ((reinterpret_cast<uintptr_t>(Zt) & 15) == 0) && (ldz % 4 == 0);
  return q2_slide_possible(nrows, n) && vec && device_cu_count() > 0;
}

int q2_slide_launch(float *Zt, int64_t ldz, int64_t nrows, int64_t n, const float *R2, int64_t ldr, const float *tau2, void *ws,
                    size_t ws_bytes, hipStream_t stream) {
  if (n < 3 || nrows <= 0) return VIVIT_OK;
  if (!ws || ws_bytes < q2_slide_workspace_bytes(n)) return VIVIT_E_WORKSPACE;
  static unsigned long long attr_done = 0;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return VIVIT_E_LAUNCH;
    if (!(attr_done & (1ull << (dev & 63)))) {
      if (!ensure_dynamic_lds(reinterpret_cast<const void *>(qs_apply_kernel<8>), QS_APPLY_LDS, attr_done) ||
          !ensure_dynamic_lds(reinterpret_cast<const void *>(qs_apply_kernel<12>), QS_APPLY_LDS, attr_done) ||
          !ensure_dynamic_lds(reinterpret_cast<const void *>(qs_prepare_kernel), QS_PREP_LDS, attr_done))
        return VIVIT_E_LAUNCH;
      attr_done |= 1ull << (dev & 63);
    }
  }
  unsigned char *img = reinterpret_cast<unsigned char *>(align_up(reinterpret_cast<uintptr_t>(ws), 1024));
  // the last 2 KB (8 XCDs x 64 ints) of the workspace: the loaders' progress board, zeroed in front of every launch
  int *board = reinterpret_cast<int *>((reinterpret_cast<uintptr_t>(ws) + ws_bytes - 2048) & ~(uintptr_t)255);
  const size_t img_bytes = (size_t)(reinterpret_cast<unsigned char *>(board) - img);
  static int window = -1;
  if (window < 0) window = qs_env("VIVIT_Q2_LOCKSTEP", QS_LOCKSTEP_DEFAULT);
  const int G0 = (int)((n - 2) / 64);
  const int Kend = G0 / 2 + 1;   // passes K with gmax(K) = G0 - 2K >= 0
  // waves per workgroup (16 rows each): one slab per CU when the rows allow it, at most 12 waves (168 registers)
  static int force_nw = -2;
  if (force_nw == -2) force_nw = qs_env("VIVIT_Q2_SLIDE_WAVES", -1);
  const int cus = device_cu_count() > 0 ? device_cu_count() : 256;
  int nw = (int)cdiv(cdiv(nrows, cus), 16);
  if (force_nw > 0) nw = force_nw;
  if (nw < 1) nw = 1;
  static int nload_env = -2;
  if (nload_env == -2) nload_env = qs_env("VIVIT_Q2_SLIDE_LOADERS", 2);
  // loader waves (they only request images): behind at most ten compute waves, so that the workgroup stays within twelve
  const int nload = (nload_env > 0 && nw <= 10) ? (nload_env < 12 - nw ? nload_env : 12 - nw) : 0;
  if (nw > 12) nw = 12;
  const int nwt = nw + nload;
  const unsigned nslab = (unsigned)cdiv(nrows, 16 * nw);
  QsPrep pa;
  pa.R2 = R2; pa.ldr = ldr; pa.tau2 = tau2; pa.nk = sb2st_num_levels(n); pa.n = (int)n; pa.G0 = G0; pa.img = img;
  QsArgs aa;
  aa.img = img; aa.Zt = Zt; aa.ldz = ldz; aa.nrows = (int)nrows; aa.n = (int)n; aa.G0 = G0; aa.nload = nload;
  aa.board = (window > 0 && nload > 0) ? board : nullptr;
  aa.window = window;
  for (int K0 = 0; K0 < Kend;) {
    const int K1 = qs_chunk_passes(G0, K0, Kend, img_bytes);
    if (K1 == K0) return VIVIT_E_WORKSPACE;
    pa.K0 = K0;
    const dim3 pgrid((unsigned)(2 * (G0 - 2 * K0) + 2), (unsigned)(K1 - K0));
    qs_prepare_kernel<<<pgrid, 256, QS_PREP_LDS, stream>>>(pa);
    aa.K0 = K0; aa.K1 = K1;
    if (aa.board && hipMemsetAsync(aa.board, 0, 2048, stream) != hipSuccess) return VIVIT_E_LAUNCH;
    if (nwt <= 8) qs_apply_kernel<8><<<nslab, 64 * nwt, QS_APPLY_LDS, stream>>>(aa);
    else qs_apply_kernel<12><<<nslab, 64 * nwt, QS_APPLY_LDS, stream>>>(aa);
    K0 = K1;
  }
  return launch_status();
}

} // namespace vivit
