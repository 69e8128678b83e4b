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
// and of the trapezoid T V are skipped: 26 tile steps x 6 = 156 MFMAs per block and wave.
#include <cstdlib>

#include "common.h"
#include "device_utils.h"
#include "eig_internal.h"

namespace vivit {

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
__device__ __forceinline__ void qs_split2(float a, float b, unsigned &hi, unsigned &mid, unsigned &lo) {
  const qbf16x2 h = {(__bf16)a, (__bf16)b};
  hi = __builtin_bit_cast(unsigned, h);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  const qbf16x2 m = {(__bf16)ra, (__bf16)rb};
  mid = __builtin_bit_cast(unsigned, m);
  const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xffff0000u);
  const qbf16x2 l = {(__bf16)sa, (__bf16)sb};
  lo = __builtin_bit_cast(unsigned, l);
}

struct QsPieces {
  qbf16x8 h, m, l;
};

// eight fp32 values (element j of an MFMA operand: x = j 0..3, y = j 4..7) -> three bf16x8 operands
__device__ __forceinline__ QsPieces qs_split8(const float4 x, const float4 y) {
  qu32x4 h, m, l;
  unsigned a, b, c;
  qs_split2(x.x, x.y, a, b, c); h[0] = a; m[0] = b; l[0] = c;
  qs_split2(x.z, x.w, a, b, c); h[1] = a; m[1] = b; l[1] = c;
  qs_split2(y.x, y.y, a, b, c); h[2] = a; m[2] = b; l[2] = c;
  qs_split2(y.z, y.w, a, b, c); h[3] = a; m[3] = b; l[3] = c;
  QsPieces p;
  p.h = __builtin_bit_cast(qbf16x8, h);
  p.m = __builtin_bit_cast(qbf16x8, m);
  p.l = __builtin_bit_cast(qbf16x8, l);
  return p;
}

// pass K = levels 2K, 2K + 1 of the groups gmax(K) = G0 - 2K .. 0, G0 = (n - 2) / 64; level 2K + 1 exists for
// g < gmax(K) only.  Blocks of a pass in walk order: (gmax, 2K), then (g, 2K), (g, 2K + 1) for g = gmax - 1 .. 0:
// 2 gmax + 1 blocks.  Blocks in front of pass K counted from pass K0:
__host__ __device__ inline int64_t qs_pass_offset(int G0, int K0, int K) {
  const int64_t d = K - K0;
  return d * (2 * (int64_t)G0 + 1) - 4 * d * K0 - 2 * d * (d - 1);
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
  if (gmax < 0 || (int)blockIdx.x > 2 * gmax) return;
  int g, k;
  if (blockIdx.x == 0) { g = gmax; k = 2 * K; }
  else { g = gmax - 1 - ((int)blockIdx.x - 1) / 2; k = 2 * K + (((int)blockIdx.x - 1) & 1); }
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
  if (tid < QS_W) tfactor_column(S, taus, Ts, QS_LDT, QS_W, tid);
  __syncthreads();
  // T V (T upper triangular): (S V^T) T^T = S (T V)^T, one product instead of two in the apply kernel
  for (int idx = tid; idx < QS_W * QS_WIN; idx += 256) {
    const int tp = idx / QS_WIN, w = idx - tp * QS_WIN;
    float acc = 0.f;
    // V[t][w] != 0 only for t in [w - 64, w - 1]
    const int tlo = tp > w - QS_B ? tp : w - QS_B, thi = w - 1 < QS_W - 1 ? w - 1 : QS_W - 1;
    for (int t = tlo; t <= thi; ++t) acc += Ts[tp * QS_LDT + t] * Vw[t * QS_LDV + w];
    TV[tp * QS_LDV + w] = acc;
  }
  __syncthreads();
  // fragments: (fragment triple ft, lane) items; a thread writes the three 16-byte pieces of its lane
  for (int idx = tid; idx < (QS_NW2 + QS_NU) * 64; idx += 256) {
    const int ft = idx >> 6, ln = idx & 63;
    const int m16 = ln & 15, kq = ln >> 4;
    float v[8];
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

template <int Q0>
__device__ __forceinline__ void qs_apply_block(float4 (&sw)[12], qs_lds_ptr frag) {
  f32x4 acc2[4];
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc2[ta][e] = 0.f;
  int f = 0;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const QsPieces bp = qs_split8(sw[Q0 + 2 * ks], sw[Q0 + 2 * ks + 1]);
#pragma unroll
    for (int ta = 0; ta < 4; ++ta) {
      if (qs_w2_need(ks, ta)) {
        const qbf16x8 ah = *reinterpret_cast<qs_lds_frag>(frag + (f + 0) * 1024);
        const qbf16x8 am = *reinterpret_cast<qs_lds_frag>(frag + (f + 1) * 1024);
        const qbf16x8 al = *reinterpret_cast<qs_lds_frag>(frag + (f + 2) * 1024);
        QS_MFMA6(acc2[ta], ah, am, al, bp)
        f += 3;
      }
    }
  }
  QsPieces wp[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
    wp[kt] = qs_split8(make_float4(acc2[2 * kt][0], acc2[2 * kt][1], acc2[2 * kt][2], acc2[2 * kt][3]),
                       make_float4(acc2[2 * kt + 1][0], acc2[2 * kt + 1][1], acc2[2 * kt + 1][2], acc2[2 * kt + 1][3]));
#pragma unroll
  for (int wt = 0; wt < 8; ++wt) {
    f32x4 u;
#pragma unroll
    for (int e = 0; e < 4; ++e) u[e] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (qs_u_need(wt, kt)) {
        const qbf16x8 ah = *reinterpret_cast<qs_lds_frag>(frag + (f + 0) * 1024);
        const qbf16x8 am = *reinterpret_cast<qs_lds_frag>(frag + (f + 1) * 1024);
        const qbf16x8 al = *reinterpret_cast<qs_lds_frag>(frag + (f + 2) * 1024);
        QS_MFMA6(u, ah, am, al, wp[kt])
        f += 3;
      }
    }
    float4 &x = sw[Q0 + wt];
    x.x -= u[0]; x.y -= u[1]; x.z -= u[2]; x.w -= u[3];
  }
}

// MAXW: most waves per workgroup of the instantiation (register budget 512 / ceil(MAXW / 4) per lane)
template <int MAXW>
__global__ __launch_bounds__(64 * MAXW) void qs_apply_kernel(QsArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char qs_lds[];  // two images
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = (int)(blockDim.x >> 6);
  const int n16 = lane & 15, kq = lane >> 4;
  const int64_t row = (int64_t)blockIdx.x * (16 * nw) + wave * 16 + n16;
  const bool rok = row < a.nrows;
  float *zrow = a.Zt + (rok ? row : 0) * a.ldz + 4 * kq;   // + 64 unit + 16 q
  const int n = a.n;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)qs_lds);
  qs_lds_ptr myfrag = (qs_lds_ptr)qs_lds + lane * 16;

  const int64_t nseq = qs_pass_offset(a.G0, a.K0, a.K1);
  // image `seq` -> LDS buffer seq & 1: this wave's share of the 78 one-KB pieces (lane i's 16 bytes land at M0 + 16 i)
  auto dma = [&](int64_t seq) __attribute__((always_inline)) {
    const unsigned char *src = a.img + seq * (int64_t)QS_IMG + lane * 16;
    const unsigned dst = lds0 + (unsigned)(seq & 1) * (unsigned)QS_IMG;
    for (int f = wave; f < QS_NFRAG; f += nw) {
      const unsigned d = __builtin_amdgcn_readfirstlane(dst + (unsigned)f * 1024u);
      __asm__ volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(d), "v"(src + (int64_t)f * 1024) : "memory");
    }
  };
  // unit u entirely inside the matrix (64 u + 63 < n): unguarded loads (rows beyond nrows read row 0, never stored)
  auto load_unit = [&](int u, float4 (&dst)[4]) __attribute__((always_inline)) {
    const float4 *src = reinterpret_cast<const float4 *>(zrow + (int64_t)64 * u);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = src[4 * q];
  };
  // the last unit of the matrix (u = G0: columns up to n - 1 exist; n % 4 == 0, so a float4 is all in or all out)
  auto load_unit_edge = [&](int u, float4 (&dst)[4]) __attribute__((always_inline)) {
    const int c0 = 64 * u + 4 * kq;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = c0 + 16 * q < n;
      dst[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) dst[q] = *reinterpret_cast<const float4 *>(zrow + (int64_t)64 * u + 16 * q);
    }
  };
  auto store_unit = [&](int u, const float4 *src) __attribute__((always_inline)) {
    const int c0 = 64 * u + 4 * kq;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (rok && c0 + 16 * q < n) *reinterpret_cast<float4 *>(zrow + (int64_t)64 * u + 16 * q) = src[q];
  };

  // The Zt loads and stores are plain C++ (hipcc counts them), the image DMAs are asm (hipcc does not): a wait that hipcc
  // places for one of ITS loads also drains every DMA issued before that point.  The loop is therefore arranged so that
  // hipcc's waits fall where nothing of ours is in flight:
  //   * nothing of hipcc's is pending at the loop entry (compiler-visible vmcnt(0) behind the pass-start loads);
  //   * the prefetched unit `pre` is "used" by an empty asm right behind the wait at the top of the second block, before
  //     the next image is requested, so hipcc's wait for it is a no-op there;
  //   * the stores of the retired unit need no wait at all; the wait at the top of the next first block leaves them in
  //     flight (vmcnt(4): they are the four youngest operations) when all four were issued (`steady`).
#define QS_USE4(a) "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)
  const bool wave_valid = (int64_t)blockIdx.x * (16 * nw) + wave * 16 < a.nrows;   // wave-uniform: some row of the wave exists
  int64_t seq = 0;
  if (nseq > 0) dma(0);
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
    bool steady = false;   // the previous step issued exactly four stores behind the image request
    for (int g = gmax; g >= 0; --g) {
      // ---- block (g, 2K): its image has landed once every wave is past this wait and the barrier
      if (steady) __asm__ volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (seq + 1 < nseq) dma(seq + 1);
      // next group's left unit (g - 1 + 2K < G0: inside the matrix), in flight during the block.  Unconditional (the last
      // group of a pass fetches a unit it does not use): a branch around the loads makes hipcc wait for them at the join
      float4 pre[4];
      {
        const int up = g - 1 + 2 * K;
        load_unit(up > 0 ? up : 0, pre);
      }
      __builtin_amdgcn_sched_barrier(0);   // (the loads stay up here)
      qs_apply_block<0>(sw, myfrag + (seq & 1) * QS_IMG);
      ++seq;
      if (g < gmax) {
        // ---- block (g, 2K + 1)
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __asm__ volatile("" : QS_USE4(pre[0]), QS_USE4(pre[1]), QS_USE4(pre[2]), QS_USE4(pre[3]));
        if (seq + 1 < nseq) dma(seq + 1);
        qs_apply_block<4>(sw, myfrag + (seq & 1) * QS_IMG);
        ++seq;
      } else {
        // first group of a pass (no second block): settle `pre` here, so that hipcc has nothing pending where the paths join
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __asm__ volatile("" : QS_USE4(pre[0]), QS_USE4(pre[1]), QS_USE4(pre[2]), QS_USE4(pre[3]));
      }
      // the right unit is final: store it, slide the window
      const int ur = g + 2 * K + 2;
      store_unit(ur, &sw[8]);
      steady = wave_valid && g < gmax && 64 * ur + 63 < n;
#pragma unroll
      for (int q = 0; q < 4; ++q) { sw[8 + q] = sw[4 + q]; sw[4 + q] = sw[q]; sw[q] = pre[q]; }
    }
    // after g = 0: units 2K (slot 1) and 2K + 1 (slot 2) are still in registers
    store_unit(2 * K, &sw[4]);
    store_unit(2 * K + 1, &sw[8]);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // the next pass reads what this one stored
  }
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

size_t q2_slide_workspace_bytes(int64_t n) {
  if (n < 3) return 0;
  const int G0 = (int)((n - 2) / 64);
  // all images, at most ~2 GB of them at a time, at least the first pass (the longest): 2 G0 + 1 blocks
  const size_t all = (size_t)qs_pass_offset(G0, 0, G0 / 2 + 1) * QS_IMG, one = (size_t)(2 * G0 + 1) * QS_IMG;
  const size_t cap = all < QS_WS_TARGET ? all : QS_WS_TARGET;
  return (one > cap ? one : cap) + 2048;
}

static int qs_env(const char *name, int dflt) {
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}

// the sliding-window form pays when every CU gets a slab of a few waves: rows >= QS_MIN_ROWS (VIVIT_Q2_SLIDE_MIN_ROWS)
bool q2_slide_ok(int64_t nrows, int64_t n, const float *Zt, int64_t ldz) {
  static int on = -1, min_rows = 0;
  if (on < 0) {
    on = qs_env("VIVIT_Q2_SLIDE", 1);
    min_rows = qs_env("VIVIT_Q2_SLIDE_MIN_ROWS", 12288);
  }
  const bool vec = ((reinterpret_cast<uintptr_t>(Zt) & 15) == 0) && (ldz % 4 == 0) && (n % 4 == 0);
  return on != 0 && vec && n >= 192 && nrows >= min_rows && device_cu_count() > 0;
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
  const size_t img_bytes = ws_bytes - (size_t)(img - reinterpret_cast<unsigned char *>(ws));
  const int G0 = (int)((n - 2) / 64);
  const int Kend = G0 / 2 + 1;   // passes K with gmax(K) = G0 - 2K >= 0
  // waves per workgroup: one slab per CU when the rows allow it (16 rows per wave), at most 12 waves (168 registers)
  const int cus = device_cu_count() > 0 ? device_cu_count() : 256;
  int nw = (int)cdiv(cdiv(nrows, cus), 16);
  static int force_nw = -2;
  if (force_nw == -2) force_nw = qs_env("VIVIT_Q2_SLIDE_WAVES", -1);
  if (force_nw > 0) nw = force_nw;
  if (nw < 1) nw = 1;
  if (nw > 12) nw = 12;
  const unsigned nslab = (unsigned)cdiv(nrows, 16 * nw);
  QsPrep pa;
  pa.R2 = R2; pa.ldr = ldr; pa.tau2 = tau2; pa.nk = sb2st_num_levels(n); pa.n = (int)n; pa.G0 = G0; pa.img = img;
  QsArgs aa;
  aa.img = img; aa.Zt = Zt; aa.ldz = ldz; aa.nrows = (int)nrows; aa.n = (int)n; aa.G0 = G0;
  for (int K0 = 0; K0 < Kend;) {
    const int K1 = qs_chunk_passes(G0, K0, Kend, img_bytes);
    if (K1 == K0) return VIVIT_E_WORKSPACE;
    pa.K0 = K0;
    qs_prepare_kernel<<<dim3((unsigned)(2 * (G0 - 2 * K0) + 1), (unsigned)(K1 - K0)), 256, QS_PREP_LDS, stream>>>(pa);
    aa.K0 = K0; aa.K1 = K1;
    if (nw <= 8) qs_apply_kernel<8><<<nslab, 64 * nw, QS_APPLY_LDS, stream>>>(aa);
    else qs_apply_kernel<12><<<nslab, 64 * nw, QS_APPLY_LDS, stream>>>(aa);
    K0 = K1;
  }
  return launch_status();
}

} // namespace vivit
