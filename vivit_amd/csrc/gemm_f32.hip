// fp32 MFMA GEMM / SYRK for gfx950 (MI355X).
//
//   C[M,N] = alpha * op(A) op(B)^T + beta * C,   exact fp32 (v_mfma_f32_32x32x2_f32 is a
//   k-ordered fmaf chain), accumulators flushed into a second accumulator every 2048 k so that
//   long contractions (P up to ~4e5 for the Gram build) keep pairwise-like rounding error.
//
// Tile: 128x128x16 per 256-thread workgroup; 4 waves in a 2x2 grid, each wave owns a 64x64
// block = 2x2 MFMA 32x32 tiles (64 accumulator VGPRs + 64 for the second level).  Operands are
// register-staged into double-buffered LDS (one barrier per K tile).  fp32 MFMA runs at the
// vector rate (64 flop/clk/SIMD), so one wave spends 2048 cycles of matrix work per K tile and
// the 4 global float4 loads + 8 ds_read_b128 per wave per K tile hide completely behind it.
//
// blockIdx -> tile mapping is XCD-aware: the grid is cut into 16x16-tile super-blocks; inside a
// super-block the 8 workgroups that share an XCD (blockIdx % 8, round-robin dispatch) own one
// compact 8x4-tile sub-block, so the A/B row panels they stream are shared through that XCD's
// L2 (speed only, never correctness).
//
// SYRK mode (the Gram build, K1): B == A, only super-blocks/tiles with tile_i >= tile_j are
// computed (n(n+1)p flops instead of 2n^2p) and off-diagonal tiles are stored twice, the mirror
// image transposed through LDS so that both stores are coalesced.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace vivit {

constexpr int BM = 128, BN = 128, BK = 16;     // default tile (2 x 2 waves); the WM = 1 variant is 64 x 256
constexpr int SK = BK + 4;                     // LDS row stride (floats) of a LAY_K tile [rows][20]
__host__ __device__ constexpr int tile_floats(int rows) {  // one operand tile of `rows` rows in either layout
  return rows * SK > BK * (rows + 4) ? rows * SK : BK * (rows + 4);
}
constexpr int FLUSH_TILES = 2048 / BK;         // second-level accumulation period
constexpr int SB = 16;                         // super-block edge in tiles

// Operand pointers are re-read from a device-resident descriptor in batched mode, which makes
// hipcc lose their address space and emit flat_load (slower, and waited for with vmcnt(0) +
// lgkmcnt(0)).  All global accesses therefore go through explicitly global-address-space pointers.
typedef const float __attribute__((address_space(1))) *gcptr;
typedef float __attribute__((address_space(1))) *gptr;
typedef const f32x4 __attribute__((address_space(1))) *gcptr4;
__device__ __forceinline__ float4 ldg4(gcptr q) {
  const f32x4 v = *(gcptr4)q;
  return make_float4(v.x, v.y, v.z, v.w);
}

// Guarded element load, BRANCH-FREE: out-of-range elements read a clamped in-range address and
// select zero.  (A branch per load makes hipcc put s_waitcnt vmcnt(0) in front of every load and
// in front of the MFMA block, which serialises the stream and exposes the full HBM latency.)
__device__ __forceinline__ float ld1_sel(gcptr P, int64_t idx_major, int64_t n_major,
                                         int64_t idx_minor, int64_t n_minor, int64_t ld) {
  const bool ok = idx_major < n_major && idx_minor < n_minor;
  const int64_t a = idx_major < n_major ? idx_major : n_major - 1;
  const int64_t b = idx_minor < n_minor ? idx_minor : n_minor - 1;
  const float x = P[a * ld + b];
  return ok ? x : 0.f;
}

// Global -> registers for one ROWS x 16 operand tile (ROWS / 64 float4 per thread).  MODE:
//   0  tile completely in range, operand 16-byte aligned: unconditional float4
//   1  aligned operand, ragged rows (and, for LAY_M, row count % 4 == 0): float4 from a clamped
//      row + zero select -- still one vector load per thread and no branch
//   2  anything else: clamped scalar loads
// Full K tiles only for modes 0/1 (a ragged last K tile is loaded with mode 2).
template <int LAY, int MODE, int ROWS>
__device__ __forceinline__ void tile_load(gcptr P, int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                          int64_t kend, int tid, float4 (&st)[ROWS / 64]) {
#pragma unroll
  for (int q = 0; q < ROWS / 64; ++q) {
    const int f = tid + 256 * q;
    if (LAY == LAY_K) {
      const int64_t row = row0 + (f >> 2), k = k0 + 4 * (f & 3);
      if constexpr (MODE == 0) {
        st[q] = ldg4(P + row * ld + k);
      } else if constexpr (MODE == 1) {
        const bool ok = row < nrows;
        const float4 v = ldg4(P + (ok ? row : nrows - 1) * ld + k);
        st[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        st[q] = make_float4(ld1_sel(P, row, nrows, k, kend, ld), ld1_sel(P, row, nrows, k + 1, kend, ld),
                            ld1_sel(P, row, nrows, k + 2, kend, ld), ld1_sel(P, row, nrows, k + 3, kend, ld));
      }
    } else {
      const int64_t row = row0 + 4 * (f & (ROWS / 4 - 1)), k = k0 + f / (ROWS / 4);
      if constexpr (MODE == 0) {
        st[q] = ldg4(P + k * ld + row);
      } else if constexpr (MODE == 1) {
        const bool ok = row < nrows;  // nrows % 4 == 0: the whole float4 is in or out
        const float4 v = ldg4(P + k * ld + (ok ? row : 0));
        st[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        st[q] = make_float4(ld1_sel(P, k, kend, row, nrows, ld), ld1_sel(P, k, kend, row + 1, nrows, ld),
                            ld1_sel(P, k, kend, row + 2, nrows, ld), ld1_sel(P, k, kend, row + 3, nrows, ld));
      }
    }
  }
}

// Registers -> LDS.
template <int LAY, int ROWS>
__device__ __forceinline__ void tile_store(float *__restrict__ s, int tid, const float4 (&st)[ROWS / 64]) {
#pragma unroll
  for (int q = 0; q < ROWS / 64; ++q) {
    const int f = tid + 256 * q;
    if (LAY == LAY_K) {
      *reinterpret_cast<float4 *>(s + (f >> 2) * SK + 4 * (f & 3)) = st[q];
    } else {
      *reinterpret_cast<float4 *>(s + (f / (ROWS / 4)) * (ROWS + 4) + 4 * (f & (ROWS / 4 - 1))) = st[q];
    }
  }
}

// MFMA operand fragments for the 8 k-pairs of one K tile.  MFMA u = 4q + t (q in 0..1,
// t in 0..3) consumes k = 8q + 4h + t from lane half h = lane >> 5; both layouts use that same
// assignment so any A layout pairs with any B layout.
//   frag[q][t] for rows r0 + (lane & 31).
template <int LAY, int ROWS>
__device__ __forceinline__ void frag_load(const float *__restrict__ s, int r, int h, float (&fr)[2][4]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (LAY == LAY_K) {
      const float4 v = *reinterpret_cast<const float4 *>(s + r * SK + 4 * (2 * q + h));
      fr[q][0] = v.x; fr[q][1] = v.y; fr[q][2] = v.z; fr[q][3] = v.w;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) fr[q][t] = s[(8 * q + 4 * h + t) * (ROWS + 4) + r];
    }
  }
}

// Super-block geometry: 256 tiles per super-block, SBH x SBW tiles (16 x 16 for large outputs; for
// skinny outputs the short side shrinks to the next power of two >= its tile count, so that a
// one-tile-wide or one-tile-tall GEMM still spreads over all 8 XCDs), each XCD owning one compact
// sub-block of 32 tiles (xh x xw).
__host__ __device__ inline int sb_width(int tiles_m, int tiles_n) {
  int w = 1;
  while (w < 16 && w < tiles_n) w <<= 1;
  if (w == 16 && tiles_m < 16) {  // short in M instead: widen the super-block
    int hgt = 1;
    while (hgt < 16 && hgt < tiles_m) hgt <<= 1;
    w = 256 / hgt;
  }
  return w;
}

// blockIdx.x -> (tile_i, tile_j); returns false for padding slots.
__device__ __forceinline__ bool map_tile(int syrk, int SBW, int tiles_m, int tiles_n, int &ti, int &tj) {
  const int sb = blockIdx.x >> 8;
  // split-K: rotate the slots with the split index.  Workgroups go to XCD (linear id % 8) and every split's
  // grid row starts at a multiple of 256, so without the rotation slot 0 of EVERY split - the only valid one of
  // a single-tile output - lands on XCD 0 and the whole product runs on one eighth of the chip.
  const int slot = (blockIdx.x - blockIdx.y) & 255;
  const int SBH = 256 / SBW;
  int I, J;
  if (syrk) {
    I = (int)((sqrtf(8.f * (float)sb + 1.f) - 1.f) * 0.5f);
    while ((I + 1) * (I + 2) / 2 <= sb) ++I;
    while (I * (I + 1) / 2 > sb) --I;
    // row I of the super-block triangle from its diagonal block leftwards: the LAST super-block of the grid is then an
    // off-diagonal one (diagonal tiles flush their accumulators more often - bx_flush_tiles - and a slower tail block
    // delayed every chunk launch of the Gram SYRK)
    J = I - (sb - I * (I + 1) / 2);
  } else {
    const int sbn = (tiles_n + SBW - 1) / SBW;
    I = sb / sbn;
    J = sb - I * sbn;
  }
  const int xcd = slot & 7, w = slot >> 3;
  // Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), so every super-block must hand each XCD
  // the same number of tiles or the busiest XCD sets the kernel time.  Full rectangular super-blocks do
  // (32 tiles per XCD in a compact sub-block).  The others - the lower triangles on the diagonal of a SYRK
  // and the partial super-blocks at the bottom/right edge - are dealt out evenly instead: XCD x takes the
  // c = ceil(v / 8) consecutive valid tiles [c x, c x + c) in row-major order.  (Config-2 Gram matrix before
  // this: 26/10/0/0/32/32/26/10 tiles of every diagonal super-block per XCD, busiest XCD 1760 tiles against
  // a mean of 1610, kernel 9 % over its MFMA time.)
  const int vr = tiles_m - I * SBH < SBH ? tiles_m - I * SBH : SBH;  // valid rows / columns of this super-block
  const int vc = tiles_n - J * SBW < SBW ? tiles_n - J * SBW : SBW;
  if (syrk && I == J) {
    const int d = vr < vc ? vr : vc;  // triangle edge
    const int v = d * (d + 1) / 2, c = (v + 7) >> 3;
    const int idx = c * xcd + w;
    if (w >= c || idx >= v) return false;
    int a = (int)((sqrtf(8.f * (float)idx + 1.f) - 1.f) * 0.5f);
    while ((a + 1) * (a + 2) / 2 <= idx) ++a;
    while (a * (a + 1) / 2 > idx) --a;
    ti = I * SBH + a;
    tj = J * SBW + (idx - a * (a + 1) / 2);
    return true;
  }
  if (vr < SBH || vc < SBW) {
    if (vr <= 0 || vc <= 0) return false;
    const int v = vr * vc, c = (v + 7) >> 3;
    const int idx = c * xcd + w;
    if (w >= c || idx >= v) return false;
    ti = I * SBH + idx / vc;
    tj = J * SBW + idx % vc;
    return true;
  }
  int xw = SBW < 4 ? SBW : 4, xh = 32 / xw;          // XCD sub-block: xh x xw tiles
  if (xh > SBH) { xh = SBH; xw = 32 / xh; }
  const int xcols = SBW / xw;                         // XCD sub-blocks per super-block row
  ti = I * SBH + (xcd / xcols) * xh + w / xw;
  tj = J * SBW + (xcd % xcols) * xw + w % xw;
  if (ti >= tiles_m || tj >= tiles_n) return false;
  if (syrk && tj > ti) return false;
  return true;
}

// map_tile + the K split of the workgroup.  SBW >= 0: the split is blockIdx.y.  SBW < 0 (the split-K launches of SMALL outputs: one
// partial super-block of v valid tiles): a COMPACT one-dimensional grid of 8 v ceil(nsplit / 8) workgroups, workgroup L -> tile
// (L / 8) % v of split 8 (L / 8 / v) + L % 8.  Workgroups go to XCD L % 8, so ALL tiles of a split run on ONE XCD at about the same
// time and share their operand panels in its L2 -- the padded grid (256 slots per split row, slots rotated over the XCDs) gave
// every XCD tiles of many splits that share nothing, so the product ran at the HBM rate of 48 KB per K tile and workgroup, and its
// 8 200 padding workgroups (each asking for a whole CU's LDS) queued for whatever CU was free: the config-1 Gram SYRK (n = 1280,
// K = 401 408: 15 tiles x 34 splits) took 7.1 ms for 2.4 ms of tile time (profiles/r06_splitk_compact.log).
__device__ __forceinline__ bool map_tile_z(int syrk, int SBW, int tiles_m, int tiles_n, int nsplit, int &ti, int &tj, int &zsplit) {
  if (SBW >= 0) {
    zsplit = (int)blockIdx.y;
    return map_tile(syrk, SBW, tiles_m, tiles_n, ti, tj);
  }
  const int d = tiles_m < tiles_n ? tiles_m : tiles_n;
  const int v = syrk ? d * (d + 1) / 2 : tiles_m * tiles_n;
  const int L = (int)blockIdx.x, r = L >> 3;
  const int idx = r % v;
  zsplit = 8 * (r / v) + (L & 7);
  if (zsplit >= nsplit) return false;
  if (syrk) {
    int a = (int)((sqrtf(8.f * (float)idx + 1.f) - 1.f) * 0.5f);
    while ((a + 1) * (a + 2) / 2 <= idx) ++a;
    while (a * (a + 1) / 2 > idx) --a;
    ti = a;
    tj = idx - a * (a + 1) / 2;
  } else {
    ti = idx / tiles_n;
    tj = idx % tiles_n;
  }
  return true;
}

// Number of valid tiles map_tile hands XCD `blockIdx.x & 7` in this workgroup's super-block (the members of its XCD
// group: they share their operand panels through that XCD's L2), and the group's index.  Same geometry as map_tile.
__device__ __forceinline__ int xcd_group(int syrk, int SBW, int tiles_m, int tiles_n, int &group) {
  const int sb = blockIdx.x >> 8;
  const int xcd = (blockIdx.x - blockIdx.y) & 7;
  const int SBH = 256 / SBW;
  int I, J;
  if (syrk) {
    I = (int)((sqrtf(8.f * (float)sb + 1.f) - 1.f) * 0.5f);
    while ((I + 1) * (I + 2) / 2 <= sb) ++I;
    while (I * (I + 1) / 2 > sb) --I;
    J = I - (sb - I * (I + 1) / 2);
  } else {
    const int sbn = (tiles_n + SBW - 1) / SBW;
    I = sb / sbn;
    J = sb - I * sbn;
  }
  group = sb * 8 + xcd;
  const int vr = tiles_m - I * SBH < SBH ? tiles_m - I * SBH : SBH;
  const int vc = tiles_n - J * SBW < SBW ? tiles_n - J * SBW : SBW;
  int v;
  if (syrk && I == J) {
    const int d = vr < vc ? vr : vc;
    v = d * (d + 1) / 2;
  } else if (vr < SBH || vc < SBW) {
    v = vr > 0 && vc > 0 ? vr * vc : 0;
  } else {
    return 32;
  }
  const int c = (v + 7) >> 3;
  const int left = v - c * xcd;
  return left < 0 ? 0 : (left < c ? left : c);
}

// WM = waves along M: 2 -> 128 x 128 tile (2 x 2 waves), 1 -> 64 x 256 tile (1 x 4 waves) for outputs
// with at most 64 rows (the panel products of the band reduction), where the square tile would
// spend half of its MFMAs on padding.
template <int ALAY, int BLAY, int WM = 2>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs p) {
  constexpr int WN = 4 / WM, BM = 64 * WM, BN = 64 * WN;
  constexpr int TA = tile_floats(BM), TB = tile_floats(BN);
  static_assert(WM == 1 || WM == 2, "wave grid");
  __shared__ __attribute__((aligned(16))) float smem[2 * TA + 2 * TB];
  if (p.desc) {  // batched mode: this problem's pointers and sizes come from device memory
    const GemmDesc ds = p.desc[blockIdx.z];
    p.A = ds.A; p.B = ds.B; p.C = ds.C;
    p.M = ds.M; p.N = ds.N; p.K = ds.K; p.lda = ds.lda; p.ldb = ds.ldb; p.ldc = ds.ldc;
    p.tiles_m = (int)((ds.M + BM - 1) / BM);
    p.tiles_n = (int)((ds.N + BN - 1) / BN);
    p.kchunk = ((ds.K + BK - 1) / BK) * BK;
    p.a_vec = ((reinterpret_cast<uintptr_t>(ds.A) & 15) == 0 && (ds.lda & 3) == 0) ? 1 : 0;
    p.b_vec = ((reinterpret_cast<uintptr_t>(ds.B) & 15) == 0 && (ds.ldb & 3) == 0) ? 1 : 0;
  }
  int ti, tj;
  if (!map_tile(p.syrk, p.sbw, p.tiles_m, p.tiles_n, ti, tj)) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  const int64_t row0 = (int64_t)ti * BM, col0 = (int64_t)tj * BN;
  const int64_t kbeg = (int64_t)blockIdx.y * p.kchunk;
  const int64_t kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;
  const int nt = (int)((kend - kbeg + BK - 1) / BK);

#define SA(b) (smem + (b) * TA)
#define SB_(b) (smem + 2 * TA + (b) * TB)

  f32x16 acc[2][2], tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; tot[i][j][e] = 0.f; }
  // C prefetch (accumulating GEMMs on interior tiles): the second-level accumulator starts as (beta/alpha) C,
  // read while the K loop runs, instead of reading C after it with nothing left to hide the latency
  // (the rank-128 updates of the band reduction and the back-transformation have K = 128: 8 K tiles).
  const bool prefetch_c = p.ksplit <= 1 && p.beta != 0.f && p.alpha != 0.f && row0 + BM <= p.M && col0 + BN <= p.N;
  if (prefetch_c) {
    const float ba = p.beta / p.alpha;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        gcptr cbase = (gcptr)p.C + (row0 + wm * 64 + i * 32 + 4 * h) * p.ldc + col0 + wn * 64 + j * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) tot[i][j][e] = ba * cbase[(int64_t)((e & 3) + 8 * (e >> 2)) * p.ldc];
      }
  }

  // One K tile of MFMA work from LDS buffer `cur`.
  auto compute = [&](int cur) {
    float fa[2][2][4], fb[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) frag_load<ALAY, BM>(SA(cur), wm * 64 + i * 32 + r, h, fa[i]);
#pragma unroll
    for (int j = 0; j < 2; ++j) frag_load<BLAY, BN>(SB_(cur), wn * 64 + j * 32 + r, h, fb[j]);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q][tt], fb[j][q][tt], acc[i][j], 0, 0, 0);
  };
  auto flush = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        tot[i][j] += acc[i][j];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
      }
  };

  // Register-staged double buffering: the loads of tile t+1 are issued BEFORE the MFMAs of tile t
  // and written to the other LDS buffer after them; one barrier per K tile.
  auto mainloop = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool FAST = MODE < 2;
    // modes 0/1 cover the full K tiles; a ragged last K tile goes through the scalar loader.
    const int nt_fast = FAST ? (int)((kend - kbeg) / BK) : 0;
    if constexpr (WM == 1) {
      // 64-row outputs stream their big operand once from HBM (the panel products of the band reduction):
      // two register sets keep the global loads TWO K tiles ahead (2 x 20 KB per workgroup in flight; with
      // one tile ahead the kernel was latency-bound at 1.4 TB/s)
      const int nt_fast = FAST ? (int)((kend - kbeg) / BK) : 0;
      float4 sA[2][BM / 64], sB[2][BN / 64];
      auto load2 = [&](int t, auto par) __attribute__((always_inline)) {
        constexpr int P = decltype(par)::value;
        const int64_t k0 = kbeg + (int64_t)t * BK;
        if (FAST && t < nt_fast) {
          tile_load<ALAY, MODE, BM>((gcptr)p.A, p.lda, row0, p.M, k0, kend, tid, sA[P]);
          tile_load<BLAY, MODE, BN>((gcptr)p.B, p.ldb, col0, p.N, k0, kend, tid, sB[P]);
        } else {
          tile_load<ALAY, 2, BM>((gcptr)p.A, p.lda, row0, p.M, k0, kend, tid, sA[P]);
          tile_load<BLAY, 2, BN>((gcptr)p.B, p.ldb, col0, p.N, k0, kend, tid, sB[P]);
        }
      };
      using P0 = std::integral_constant<int, 0>;
      using P1 = std::integral_constant<int, 1>;
      if (nt > 0) {
        load2(0, P0{});
        tile_store<ALAY, BM>(SA(0), tid, sA[0]);
        tile_store<BLAY, BN>(SB_(0), tid, sB[0]);
        if (nt > 1) load2(1, P1{});
      }
      __syncthreads();
      int since_flush = 0;
      // tile t: LDS buffer t & 1; register set t & 1 is free (tile t is in LDS) and receives tile t + 2;
      // tile t + 1 waits in set (t + 1) & 1 and is written to LDS after the MFMAs
      auto step = [&](int t, auto par) __attribute__((always_inline)) {
        constexpr int P = decltype(par)::value;
        if (t + 2 < nt) load2(t + 2, par);
        compute(P);
        if (++since_flush == FLUSH_TILES) { since_flush = 0; flush(); }
        if (t + 1 < nt) {
          tile_store<ALAY, BM>(SA(P ^ 1), tid, sA[P ^ 1]);
          tile_store<BLAY, BN>(SB_(P ^ 1), tid, sB[P ^ 1]);
        }
        __syncthreads();
      };
      int t = 0;
      for (; t + 1 < nt; t += 2) {
        step(t, P0{});
        step(t + 1, P1{});
      }
      if (t < nt) step(t, P0{});
      return;
    }
    float4 stA[BM / 64], stB[BN / 64];
    auto load = [&](int t) {
      const int64_t k0 = kbeg + (int64_t)t * BK;
      if (FAST && t < nt_fast) {
        tile_load<ALAY, MODE, BM>((gcptr)p.A, p.lda, row0, p.M, k0, kend, tid, stA);
        tile_load<BLAY, MODE, BN>((gcptr)p.B, p.ldb, col0, p.N, k0, kend, tid, stB);
      } else {
        tile_load<ALAY, 2, BM>((gcptr)p.A, p.lda, row0, p.M, k0, kend, tid, stA);
        tile_load<BLAY, 2, BN>((gcptr)p.B, p.ldb, col0, p.N, k0, kend, tid, stB);
      }
    };
    if (nt > 0) {
      load(0);
      tile_store<ALAY, BM>(SA(0), tid, stA);
      tile_store<BLAY, BN>(SB_(0), tid, stB);
    }
    __syncthreads();
    int since_flush = 0;
    for (int t = 0; t < nt; ++t) {
      const int cur = t & 1;
      if (t + 1 < nt) load(t + 1);
      compute(cur);
      if (++since_flush == FLUSH_TILES) { since_flush = 0; flush(); }
      if (t + 1 < nt) {
        tile_store<ALAY, BM>(SA(cur ^ 1), tid, stA);
        tile_store<BLAY, BN>(SB_(cur ^ 1), tid, stB);
      }
      __syncthreads();
    }
  };
  // a LAY_M operand needs its row count to be a multiple of 4 for the clamped vector mode
  const bool vec_ok = p.a_vec && p.b_vec && (ALAY == LAY_K || (p.M & 3) == 0) && (BLAY == LAY_K || (p.N & 3) == 0);
  const bool full = row0 + BM <= p.M && col0 + BN <= p.N;
  if (vec_ok && full) mainloop(std::integral_constant<int, 0>{});
  else if (vec_ok) mainloop(std::integral_constant<int, 1>{});
  else mainloop(std::integral_constant<int, 2>{});
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) tot[i][j] += acc[i][j];

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (e&3) + 8*(e>>2) + 4*h.
  const bool partial = p.ksplit > 1;
  gptr Cout = (gptr)(partial ? p.slab + (int64_t)blockIdx.y * p.M * p.N : p.C);
  const int64_t ldc = partial ? p.N : p.ldc;
  const float alpha = partial ? 1.f : p.alpha;
  const float beta = partial ? 0.f : p.beta;

  const bool full_tile = row0 + BM <= p.M && col0 + BN <= p.N;
  if (full_tile) {
    // unguarded epilogue: all loads (beta != 0) are issued before the first use
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        gptr cbase = Cout + (row0 + wm * 64 + i * 32 + 4 * h) * ldc + col0 + wn * 64 + j * 32 + r;
        float old[16];
        const bool rd = beta != 0.f && !prefetch_c;
        if (rd) {
#pragma unroll
          for (int e = 0; e < 16; ++e) old[e] = cbase[(int64_t)((e & 3) + 8 * (e >> 2)) * ldc];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = alpha * tot[i][j][e];
          if (rd) v += beta * old[e];
          cbase[(int64_t)((e & 3) + 8 * (e >> 2)) * ldc] = v;
          tot[i][j][e] = v;  // final value: the mirrored store below reuses it
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t col = col0 + wn * 64 + j * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t row = row0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          float v = alpha * tot[i][j][e];
          if (row < p.M && col < p.N) {
            gptr c = Cout + row * ldc + col;
            if (beta != 0.f) v += beta * *c;
            *c = v;
          }
          tot[i][j][e] = v;
        }
      }
  }

  if (WM == 2 && p.syrk == 1 && !partial && ti != tj) {
    // Mirror image: C[col][row] = same value, transposed through LDS (32x33 floats per wave)
    // so that the second store is also 128-B coalesced.
    // (each wave transposes through its own LDS patch: wave-local ordering suffices, no workgroup barrier
    // per block; all waves left the K loop through its final barrier)
    float *ts = smem + wave * (32 * 33);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int e = 0; e < 16; ++e) ts[r * 33 + (e & 3) + 8 * (e >> 2) + 4 * h] = tot[i][j][e];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int64_t mrow0 = col0 + wn * 64 + j * 32;  // rows of the mirrored block
        const int64_t mcol = row0 + wm * 64 + i * 32 + r;
#pragma unroll
        for (int rr = 0; rr < 32; rr += 2) {
          const int64_t mrow = mrow0 + rr + h;
          if (mrow < p.N && mcol < p.M) {
            // C is symmetric on entry (SYRK accumulate / symmetric rank-2k update), so the mirror
            // image equals the value just stored in the lower tile: no second read of C
            gptr c = (gptr)p.C + mrow * p.ldc + mcol;
            *c = ts[(rr + h) * 33 + r];
          }
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Large-output variant: 256 x 256 x 16 tile, 4 waves in a 2 x 2 grid, each wave a 128 x 128 block =
// 4 x 4 MFMA tiles (256 accumulator registers: one wave per SIMD, the unified 512-register file).
// Half the LDS operand traffic per flop of the 128 x 128 tile and 128 MFMAs (8192 cycles) per wave
// between barriers.  With a single wave per SIMD nothing else hides latency, so:
//  * operands go global -> LDS directly (`global_load_lds_dwordx4`: lane i's 16 bytes land at
//    M0 + 16 i, probed in scripts/probe/lds_probe.hip), no staging registers, three LDS stages: the
//    loads of tile t+2 are issued at the top of tile t (two K tiles ~ 16 000 cycles of latency cover;
//    a register-staged version with half a tile of cover stalled on HBM round trips).  The LDS image
//    of a wave instruction is one contiguous KB, so K-major tiles cannot be padded; instead each lane
//    fetches the 16-byte chunk `pos ^ ((row >> 2) & 3)` of its row (the lane -> global address map is
//    free), which makes the ds_read_b128 fragment reads conflict-free.  The DMA is issued through
//    inline asm with hand-placed `s_waitcnt vmcnt` (the compiler would wait for vmcnt(0) in front of
//    every LDS read that follows an LDS-DMA it knows about).
//  * the K loop is software-pipelined in half K tiles (8 k): the fragment reads of the next half are
//    issued in front of the 64 MFMAs of the current one; one barrier per K tile.
// Ragged edge tiles read clamped (duplicate) rows instead of zeros: those accumulators are never stored.
// The second accumulation level lives in C itself: every 8192 k the accumulators are added into the
// output tile (read-modify-write through L2, 128 KB per wave every ~10^6 cycles) and cleared - the same
// pairwise-like rounding as the register `tot` of the small tile, without the registers.
constexpr int B2 = 256;
constexpr int T2 = B2 * BK;                         // 4096 floats (16 KB) per operand tile, unpadded
constexpr int STG2 = 2 * T2;                        // one stage: A tile, B tile
constexpr int GEMM256_LDS_BYTES = 3 * STG2 * 4;     // 96 KB
constexpr int FLUSH2_TILES = 8192 / BK;             // second-level accumulation period of this kernel

template <int LAY, int ROWS = B2>
__device__ __forceinline__ void frag_half(const float *__restrict__ s, int row, int q, int h, float (&fr)[4]) {
  if (LAY == LAY_K) {
    const int c = (2 * q + h) ^ ((row >> 2) & 3);
    const float4 v = *reinterpret_cast<const float4 *>(s + row * BK + 4 * c);
    fr[0] = v.x; fr[1] = v.y; fr[2] = v.z; fr[3] = v.w;
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) fr[t] = s[(8 * q + 4 * h + t) * ROWS + row];
  }
}

// this lane's global source for 1 KB block `blk` (0..15) of an operand tile at K offset 0
template <int LAY, int ROWS = B2>
__device__ __forceinline__ gcptr dma_src(const float *P, int64_t ld, int64_t row0, int64_t nrows, int blk, int lane) {
  if (LAY == LAY_K) {  // block = 16 rows x 16 k; lane -> (row, swizzled 16-byte chunk)
    const int rl = lane >> 2, pos = lane & 3;
    int64_t row = row0 + 16 * blk + rl;
    row = row < nrows ? row : nrows - 1;
    return (gcptr)(P + row * ld + 4 * (pos ^ ((rl >> 2) & 3)));
  } else {             // tile [16 k][ROWS]; block = its floats [256 blk, 256 blk + 256); lane -> 4 of them
    const int f = 256 * blk + 4 * lane, kr = f / ROWS;
    int64_t row = row0 + (f - kr * ROWS);
    row = row + 4 <= nrows ? row : nrows - 4;  // nrows % 4 == 0 and nrows >= 4 (host)
    return (gcptr)(P + (int64_t)kr * ld + row);
  }
}

__device__ __forceinline__ void dma16(gcptr src, unsigned lds_byte_addr) {
  __asm__ volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory");
}

template <int ALAY, int BLAY>
__global__ __launch_bounds__(256, 1) void gemm256_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem2[];
  // stand-in for a bf16-pipe launch whose operand chunk is out of the split's range: runs only when the chunk is flagged
  if (p.gate && (*p.gate & p.gate_mask) == 0) return;
  int ti, tj, zsplit;
  if (!map_tile_z(p.syrk, p.sbw, p.tiles_m, p.tiles_n, p.ksplit, ti, tj, zsplit)) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)ti * B2, col0 = (int64_t)tj * B2;
  // split-K (few tiles, deep K: Gram matrices of small batches): split zsplit owns [kbeg, kend) and writes a slab
  const int64_t kbeg = (int64_t)zsplit * p.kchunk;
  const int64_t kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;
  const int nt = (int)((kend - kbeg) / BK);  // K and kchunk are multiples of 16 (host)
  const bool partial = p.ksplit > 1;

#define S2A(st) (smem2 + (st) * STG2)
#define S2B(st) (smem2 + (st) * STG2 + T2)

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  gptr Cout = (gptr)(partial ? p.slab + (int64_t)zsplit * p.M * p.N : p.C);
  const int64_t ldc = partial ? p.N : p.ldc;
  const float alpha_ = partial ? 1.f : p.alpha, beta_ = partial ? 0.f : p.beta;
  const bool full_tile = row0 + B2 <= p.M && col0 + B2 <= p.N;
  // C <- C' + alpha * acc with C' = beta * C on the first flush and C afterwards; acc <- final value
  auto flush_to_c = [&](bool first) __attribute__((always_inline)) {
    const float beta = first ? beta_ : 1.f;
    // the 256 output addresses are loop-invariant: without an opaque term LICM hoists them out of the
    // K loop (512 registers of addresses -> scratch spills in the hot loop)
    int opaque = 0;
    __asm__ volatile("" : "+v"(opaque));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __asm__ volatile("" ::: "memory");  // one tile (16 loads, 16 stores) at a time
        const int64_t rbase = row0 + wm * 128 + i * 32 + 4 * h + opaque, col = col0 + wn * 128 + j * 32 + r;
        if (full_tile) {
          gptr cbase = Cout + rbase * ldc + col;
          float old[16];
          if (beta != 0.f) {
#pragma unroll
            for (int e = 0; e < 16; ++e) old[e] = cbase[(int64_t)((e & 3) + 8 * (e >> 2)) * ldc];
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float v = alpha_ * acc[i][j][e];
            if (beta != 0.f) v += beta * old[e];
            cbase[(int64_t)((e & 3) + 8 * (e >> 2)) * ldc] = v;
            acc[i][j][e] = v;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int64_t row = rbase + (e & 3) + 8 * (e >> 2);
            float v = alpha_ * acc[i][j][e];
            if (row < p.M && col < p.N) {
              gptr c = Cout + row * ldc + col;
              if (beta != 0.f) v += beta * *c;
              *c = v;
            }
            acc[i][j][e] = v;
          }
        }
      }
  };
  auto clear_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };

  float fa[2][4][4], fb[2][4][4];  // [half parity][tile][k pair]
  auto frags = [&](int st, int q, int par) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) frag_half<ALAY>(S2A(st), wm * 128 + i * 32 + r, q, h, fa[par][i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) frag_half<BLAY>(S2B(st), wn * 128 + j * 32 + r, q, h, fb[par][j]);
  };
  // DMA sources: wave w moves blocks w, w+4, w+8, w+12 of each operand tile; pointers advance per K tile
  gcptr srcA[4], srcB[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    srcA[u] = dma_src<ALAY>(p.A, p.lda, row0, p.M, wave + 4 * u, lane) + (ALAY == LAY_K ? kbeg : kbeg * p.lda);
    srcB[u] = dma_src<BLAY>(p.B, p.ldb, col0, p.N, wave + 4 * u, lane) + (BLAY == LAY_K ? kbeg : kbeg * p.ldb);
  }
  const int64_t stepA = (ALAY == LAY_K) ? BK : (int64_t)BK * p.lda, stepB = (BLAY == LAY_K) ? BK : (int64_t)BK * p.ldb;
  // LDS byte address of this wave's first block (wave-uniform: SGPR for M0)
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem2 +
                                                       (unsigned)(wave * 256 * 4));
  // issue the 8 DMA instructions of the next not yet requested K tile into stage st
  auto issue = [&](int st) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      dma16(srcA[u], lds0 + (unsigned)((st * STG2 + 4 * u * 256) * 4));
      srcA[u] += stepA;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      dma16(srcB[u], lds0 + (unsigned)((st * STG2 + T2 + 4 * u * 256) * 4));
      srcB[u] += stepB;
    }
  };

  // One K tile at stage S.  LOAD: request tile t+2 into stage S+2; NEXT: tile t+1 exists.  Tile t+1's DMA
  // was requested one K tile (8192 MFMA cycles) ago; every wave waits for its own part at the top of the
  // tile, BEFORE requesting tile t+2, and the barrier between the two halves publishes it to the other
  // waves.  That wait is the compiler-visible `s_waitcnt` builtin on purpose: hipcc cannot see the asm DMA,
  // but it does see its own spill reloads / flush accesses in the loop preheader, and with those pending in
  // its scoreboard it would put a vmcnt(0) in front of the first MFMA of every trip - after the DMA request.
  // The LDS reads (8 per half) and DMA requests (8 per tile) are spread over the four 16-MFMA k-steps of a
  // half instead of being issued back to back (the LDS / VMEM issue queues are short: a burst stalls the
  // wave at issue and drains the MFMA pipe); sched_barrier(0) fences keep hipcc from regrouping them.
  auto frag1 = [&](int st, int q, int par, int u) __attribute__((always_inline)) {  // A tile u and B tile u
    frag_half<ALAY>(S2A(st), wm * 128 + u * 32 + r, q, h, fa[par][u]);
    frag_half<BLAY>(S2B(st), wn * 128 + u * 32 + r, q, h, fb[par][u]);
  };
  auto mfma_step = [&](int par, int tt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[par][i][tt], fb[par][j][tt], acc[i][j], 0, 0, 0);
  };
  auto issue2 = [&](int st, int u) __attribute__((always_inline)) {  // DMA blocks u of A and B
    dma16(srcA[u], lds0 + (unsigned)((st * STG2 + 4 * u * 256) * 4));
    srcA[u] += stepA;
    dma16(srcB[u], lds0 + (unsigned)((st * STG2 + T2 + 4 * u * 256) * 4));
    srcB[u] += stepB;
  };
  auto body = [&](auto stage, bool do_load, bool has_next) __attribute__((always_inline)) {
    constexpr int S = decltype(stage)::value, S1 = (S + 1) % 3, S2 = (S + 2) % 3;
    __builtin_amdgcn_sched_barrier(0);   // tile boundary: the wait below stays behind the previous tile's MFMAs
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      if (do_load) issue2(S2, tt);
      frag1(S, 1, 1, tt);
      mfma_step(0, tt);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (has_next) __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      if (has_next) frag1(S1, 0, 0, tt);
      mfma_step(1, tt);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  // K is processed in chunks of FLUSH2_TILES tiles: a clean software-pipelined loop per chunk (the
  // accumulators stay in AGPRs), then the chunk sum is added into C.
  auto chunk = [&](int t0, int t1) __attribute__((always_inline)) {
    __syncthreads();  // every wave is done with the LDS stages of the previous chunk
    issue(0);
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t0 + 1 < t1) issue(1);
    __syncthreads();
    frags(0, 0, 0);
    int t = t0;
    for (; t + 4 < t1; t += 3) {  // steady state: three tiles per trip, every one requests tile t+2
      body(I0{}, true, true);
      body(I1{}, true, true);
      body(I2{}, true, true);
    }
    // at most 4 tiles left
    if (t < t1) { body(I0{}, t + 2 < t1, t + 1 < t1); ++t; }
    if (t < t1) { body(I1{}, t + 2 < t1, t + 1 < t1); ++t; }
    if (t < t1) { body(I2{}, t + 2 < t1, t + 1 < t1); ++t; }
    if (t < t1) { body(I0{}, false, false); ++t; }
  };
  bool first_flush = true;
  for (int t0 = 0; t0 < nt; t0 += FLUSH2_TILES) {
    const int t1 = t0 + FLUSH2_TILES < nt ? t0 + FLUSH2_TILES : nt;
    if (t0 > 0) clear_acc();
    chunk(t0, t1);
    flush_to_c(first_flush);  // after the last chunk acc holds the final values of the tile
    first_flush = false;
  }

  if (p.syrk == 1 && ti != tj && !partial) {
    // Mirror image through LDS (32 x 33 floats per wave), as in the small-tile kernel
    __syncthreads();  // the last K tile has no barrier: every wave must be done reading the LDS stages
    float *ts = smem2 + wave * (32 * 33);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // own patch: wave-local ordering suffices
#pragma unroll
        for (int e = 0; e < 16; ++e) ts[r * 33 + (e & 3) + 8 * (e >> 2) + 4 * h] = acc[i][j][e];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int64_t mrow0 = col0 + wn * 128 + j * 32;
        const int64_t mcol = row0 + wm * 128 + i * 32 + r;
#pragma unroll
        for (int rr = 0; rr < 32; rr += 2) {
          const int64_t mrow = mrow0 + rr + h;
          if (mrow < p.N && mcol < p.M) {
            gptr c = (gptr)p.C + mrow * p.ldc + mcol;
            *c = ts[(rr + h) * 33 + r];
          }
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
// fp32 GEMM / SYRK on the bf16 matrix pipe ("bf16 x 6"): the fp32 MFMA (v_mfma_f32_32x32x2_f32) retires 64 flop per
// SIMD-cycle, v_mfma_f32_32x32x16_bf16 1024.  Every fp32 operand is split EXACTLY into three bf16 pieces,
//   a = a_hi + a_mid + a_lo    (a_hi = bf16(a), a_mid = bf16(a - a_hi), a_lo = bf16(a - a_hi - a_mid); 3 x 8 = 24
//                               significand bits, the two subtractions are exact in fp32),
// and a product a b is the sum of the partial products whose weight is at least 2^-16 of it,
//   hi hi + hi mid + mid hi + hi lo + lo hi + mid mid          (each EXACT in the MFMA: 8 x 8 -> 16 bits, fp32 accumulate);
// the three dropped ones (mid lo, lo mid, lo lo) are below 2^-24 |a b|, i.e. below the rounding error the fp32 MFMA
// commits on the product itself.  Six bf16 MFMAs replace eight fp32 MFMAs per 16 k: 2.67x the matrix-pipe rate for the
// same accumulation arithmetic (fp32 accumulators, the same two-level flush into C, tile map and epilogues of
// gemm256_kernel).  Operands are K-contiguous (LAY_K) on both sides: the Gram SYRK and the NT products.
// (The six-product kernel issues them as THREE v_mfma_f32_16x16x32_bf16 per 16 x 16 tile -- lo hi + hi lo, mid mid + mid hi,
// hi mid + hi hi, each instruction summing two partial products over the same 16 k --: same pipe rate, same exactness argument,
// a higher clock under the power limit; see gemm256_bx_kernel.)
//
// Two kernels: bx_split_kernel writes the three pieces of a column chunk of an operand as bf16 matrices (HBM-bound:
// 4 B read + 6 B written per element, 1-2 % of the product's time); gemm256_bx_kernel is then a pure bf16 GEMM on the
// data path of gemm256_kernel: global -> LDS DMA, three stages, requested two K tiles ahead, no VALU work per
// element at all.  (A first version split inside the GEMM, global -> registers -> 3 bf16 -> LDS: hipcc would not
// overlap the ~300 VALU operations per K tile with the 96 MFMAs of a one-wave-per-SIMD kernel, and kept the
// prefetched values in scratch: 207 TFLOP/s-equivalent at best against 149 for the fp32 MFMA kernel.)
// LDS: per stage and operand 3 pieces x 8 blocks of 1 KB; block b = rows 32 b .. 32 b + 31 as [k half][32 rows][8 bf16]:
// ONE DMA instruction fills a block (lane -> (row lane % 32, half lane / 32), 16 B each), ONE conflict-free
// ds_read_b128 per lane reads an MFMA operand (row r, the 8 k of half h).  A and B use the same assignment of k to
// (half, slot), which is all the MFMA needs (the sum over k is order independent).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef const unsigned short __attribute__((address_space(1))) *gcptr16;
constexpr int BX_PIECE = 8 * 1024;               // bytes of one piece of one operand tile (256 rows x 16 k bf16)
constexpr int BX_OPER = 3 * BX_PIECE;            // 24 KB
constexpr int BX_STAGE = 2 * BX_OPER;            // A and B: 48 KB
#if defined(BX_EXP) && BX_EXP == 3   // timing only: 144 KB of LDS as in round 2 (the flush patch then aliases the stages: wrong results)
constexpr int GEMM256BX_LDS_BYTES = 3 * BX_STAGE;
#define BX_PATCH_OFFSET (2 * BX_STAGE)
#else
#define BX_PATCH_OFFSET (3 * BX_STAGE)
constexpr int GEMM256BX_LDS_BYTES = 3 * BX_STAGE + 4 * 4096;
#endif
// (three stages of 48 KB + a 32 x 32 flush patch per wave = 160 KB)

__device__ __forceinline__ void bx_split2(float a, float b, unsigned &hi, unsigned &mid, unsigned &lo) {
  const bf16x2 h = {(__bf16)a, (__bf16)b};
  hi = __builtin_bit_cast(unsigned, h);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  const bf16x2 m = {(__bf16)ra, (__bf16)rb};
  mid = __builtin_bit_cast(unsigned, m);
  const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xffff0000u);
  const bf16x2 l = {(__bf16)sa, (__bf16)sb};
  lo = __builtin_bit_cast(unsigned, l);
}

// Pieces of the column chunk [k0, k0 + kc) of A (rows x ., row stride lda) in the BLOCKED layout the GEMM's DMA wants:
//   P[pc][k tile kt (16 k)][row block rb (32 rows)] = 1 KB = [k half][32 rows][8 bf16]
// so that one global_load_lds instruction of the GEMM reads 1 KB of consecutive bytes (a row-major piece matrix made
// every lane fetch 16 bytes from a different cache line: 4x the L2 traffic, the product ran at 105 TFLOP/s-equivalent).
// One workgroup converts 32 rows x 64 k: coalesced 32-byte reads per lane, transposition through LDS, 1 KB bursts out.
// Rows beyond `rows` are written as zeros.  kc % 16 == 0; grid.x = row blocks, grid.y = groups of 4 k tiles.
//
// Range gate: the three-way split is exact for finite values whose smallest piece is a NORMAL bf16 number.  A value that
// is non-finite or rounds to +-inf as bf16 (|a| >= 0x7F7F8000 = 3.3962e38) sets bit 0 of *flag, a non-zero value below
// 2^-100 (its `lo` piece could fall below 2^-126) sets bit 1; the bf16-pipe product of a flagged chunk returns at once
// and the fp32 MFMA kernel, launched behind it on the same columns, computes the chunk instead (BX_GATE below).
constexpr int BX_GATE_RANGE = 1, BX_GATE_TINY = 2;
template <int LAY>
__global__ __launch_bounds__(256) void bx_split_kernel(const float *__restrict__ A, int64_t rows, int64_t lda, int64_t k0,
                                                       int64_t kc, unsigned short *__restrict__ P, int64_t piece_stride,
                                                       int64_t nrb, int *__restrict__ flag) {
  __shared__ __attribute__((aligned(16))) unsigned char sp[3][4][1024];
  __shared__ float tr[LAY == LAY_M ? 64 * 33 : 1];
  const int tid = threadIdx.x;
  const int64_t rb = blockIdx.x;
  const int64_t kt0 = (int64_t)blockIdx.y * 4;
  const int64_t nkt = kc >> 4;
  const int rl = tid >> 3, seg = tid & 7;           // row in the block, 8-float segment of the 64 k
  float v[8];
  if (LAY == LAY_K) {
    const int64_t row = rb * 32 + rl;
    const int64_t k = kt0 * 16 + seg * 8;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (row < rows && k < kc) {
      const float4 *src = reinterpret_cast<const float4 *>(A + row * lda + k0 + k);
      v0 = src[0];
      v1 = src[1];
    }
    v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
  } else {
    // k-major source X[k][row]: coalesced reads along the rows (8 rows per thread), transposition through LDS
    const int kk = tid >> 2, rs = tid & 3;
    const int64_t k = kt0 * 16 + kk, row = rb * 32 + rs * 8;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (k < kc && row + 8 <= rows) {
      const float4 *src = reinterpret_cast<const float4 *>(A + (k0 + k) * lda + row);
      v0 = src[0];
      v1 = src[1];
    } else if (k < kc) {
      float e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = row + j < rows ? A[(k0 + k) * lda + row + j] : 0.f;
      v0 = make_float4(e[0], e[1], e[2], e[3]);
      v1 = make_float4(e[4], e[5], e[6], e[7]);
    }
    float *d = tr + kk * 33 + rs * 8;
    d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w; d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tr[(seg * 8 + j) * 33 + rl];
  }
  {
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = fabsf(v[j]);
      bad |= !(x < __uint_as_float(0x7F7F8000u)) ? BX_GATE_RANGE : 0;           // inf, NaN, rounds to inf as bf16
      bad |= (x < __uint_as_float(0x0D800000u) && x != 0.f) ? BX_GATE_TINY : 0;  // 0 < |a| < 2^-100
    }
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64((bad & BX_GATE_RANGE) != 0);
    const unsigned long long m1 = __builtin_amdgcn_ballot_w64((bad & BX_GATE_TINY) != 0);
    if ((m0 | m1) != 0 && (tid & 63) == 0) atomicOr(flag, (m0 ? BX_GATE_RANGE : 0) | (m1 ? BX_GATE_TINY : 0));  // rare
  }
  {
    unsigned h[4], m[4], l[4];
    bx_split2(v[0], v[1], h[0], m[0], l[0]);
    bx_split2(v[2], v[3], h[1], m[1], l[1]);
    bx_split2(v[4], v[5], h[2], m[2], l[2]);
    bx_split2(v[6], v[7], h[3], m[3], l[3]);
    const int kt = seg >> 1, half = seg & 1;
    const int o = half * 512 + rl * 16;
    *reinterpret_cast<uint4 *>(&sp[0][kt][o]) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4 *>(&sp[1][kt][o]) = make_uint4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<uint4 *>(&sp[2][kt][o]) = make_uint4(l[0], l[1], l[2], l[3]);
  }
  __syncthreads();
  // 12 KB out: thread t moves 16 bytes of (piece, k tile) = (j / 4, j % 4) for j = t / 64 + 4 i
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = (tid >> 6) + 4 * i, pc = j >> 2, kt = j & 3;
    if (kt0 + kt < nkt) {
      unsigned char *dst = reinterpret_cast<unsigned char *>(P + pc * piece_stride) + ((kt0 + kt) * nrb + rb) * 1024 + (tid & 63) * 16;
      *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(&sp[pc][kt][(tid & 63) * 16]);
    }
  }
}

// The partial products of a split product are accumulated in a fixed order of (A piece, B piece); element (r, c) and
// element (c, r) of a DIAGONAL tile of a SYRK see that order with the roles swapped and may differ in the last bit
// (off-diagonal tiles are mirrored exactly).  The lower triangle of every diagonal 256 x 256 tile is copied up.
__global__ __launch_bounds__(256) void bx_sym_diag_kernel(float *__restrict__ C, int64_t n, int64_t ldc) {
  const int64_t base = (int64_t)blockIdx.x * B2;
  for (int idx = threadIdx.x; idx < B2 * B2; idx += 256) {
    const int64_t rr = base + idx / B2, cc = base + idx % B2;
    if (rr < n && cc < n && cc > rr) C[rr * ldc + cc] = C[cc * ldc + rr];
  }
}

struct GemmBxArgs {
  const unsigned short *A, *B;   // piece 0 of each operand (blocked layout of bx_split_kernel); pieces 1, 2 at + strideA / strideB elements
  int64_t strideA, strideB;
  int64_t nrbA, nrbB;            // 32-row blocks per k tile of each operand
  float *C;
  int64_t M, N, K, ldc;          // K = contraction length of THIS launch (multiple of 16)
  float alpha, beta;
  int tiles_m, tiles_n, syrk, sbw;
  // split-K over blockIdx.y for small outputs with a deep contraction (kt_split > 0): split z takes k tiles
  // [z kt_split, (z + 1) kt_split) and writes its partial tile (alpha = 1, beta = 0) to slab[z][M][N]
  float *slab;
  int kt_split;
  const int *gate;               // range flag of this chunk (bx_split_kernel); the kernel returns when *gate & gate_mask
  int gate_mask;
  // K tiles accumulated in one MFMA chain before the sum is added into C with a VALU add (see bx_flush_tiles);
  // flush_diag: the same for the diagonal tiles of a SYRK (sums of squares: every product has the same sign)
  int flush_tiles, flush_diag;
  // optional start barrier of an XCD group (VIVIT_BX_SYNC, see bx_sync_enabled): one zeroed counter per group
  int *sync;
};

#if defined(BX_STAMP)
// Diagnostic build only (scripts/probe/bx_clock.py; never in the product library): every workgroup stamps s_memtime (core
// clock) and s_memrealtime (100 MHz) around its K loop into a buffer of its own -- the in-kernel clock the chip holds
// under this kernel's load is d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, "DVFS give-back" item 6).
__device__ unsigned long long *g_bx_stamp = nullptr;
__device__ unsigned int g_bx_stamp_cap = 0;
#endif

#ifndef BX_S16_TRANSPOSED
#define BX_S16_TRANSPOSED 0   // 1: B fragment first (a lane holds four consecutive columns of a row; needs the asm block BX_KLOOP_ASM_F9)
#endif
#ifndef BX_FLUSH_ROWS
#define BX_FLUSH_ROWS 1   // rows of 32 x 32 blocks whose old values are in flight together in the register flush (2: the same time; 4: spills, + 4 %)
#endif
#ifndef BX_SWZ_MASK
#define BX_SWZ_MASK 7   // rows of the flush patch are rotated by (row & mask) float4 units (experiments: 3)
#endif
#ifndef BX_SHAPE16
#define BX_SHAPE16 1   // 0: the six-product kernel on v_mfma_f32_32x32x16_bf16 as in rounds 2-5 (same-box comparisons)
#endif
#include "bx_kloop_asm.inc"
// (both forms are parsed in every build: the text of a discarded `if constexpr` branch is still checked against its operand list)
#define BX_KLOOP_TEXT16 BX_KLOOP_ASM_TEXT
#define BX_KLOOP_CLOB16 BX_KLOOP_ASM_CLOBBERS
#define BX_KLOOP_TEXT32 BX_KLOOP_ASM32_TEXT
#define BX_KLOOP_CLOB32 BX_KLOOP_ASM32_CLOBBERS
#define BX_KLOOP_UNROLL 3
static_assert(BX_KLOOP_ASM_UNROLL == BX_KLOOP_UNROLL && BX_KLOOP_ASM32_UNROLL == BX_KLOOP_UNROLL, "tiles per trip of the asm blocks");
#if defined(BX_KLOOP_TEXT_OVERRIDE)     // attribution / experiment builds: a block of bx_kloop_asm_variants.inc by name, of the
#include "bx_kloop_asm_variants.inc"    // form BX_SHAPE16 selects (python scripts/gen_bx_kloop.py --variants; not part of the product)
static_assert(BX_KLOOP_UNROLL_OVERRIDE == BX_KLOOP_UNROLL, "tiles per trip of the asm blocks");
#if BX_SHAPE16
#undef BX_KLOOP_TEXT16
#undef BX_KLOOP_CLOB16
#define BX_KLOOP_TEXT16 BX_KLOOP_TEXT_OVERRIDE
#define BX_KLOOP_CLOB16 BX_KLOOP_CLOB_OVERRIDE
#else
#undef BX_KLOOP_TEXT32
#undef BX_KLOOP_CLOB32
#define BX_KLOOP_TEXT32 BX_KLOOP_TEXT_OVERRIDE
#define BX_KLOOP_CLOB32 BX_KLOOP_CLOB_OVERRIDE
#endif
#endif

// ASM (default since round 6): the steady part of the K loop runs as ONE hand-scheduled inline-asm block (bx_kloop_asm.inc,
// generated by scripts/gen_bx_kloop.py: fixed register map, every memory instruction placed between the MFMAs by hand);
// prologue, the last tiles of a pass, the chain flushes and the epilogues stay the C++ below, which is also the reference
// implementation (ASM = false, VIVIT_BX_ASM=0).  Same instructions in the same order per accumulator: bit-identical
// results (tests/test_bx_asm_gpu.py).
//
// MFMA shape (second half of round 6).  The six-product kernel computes on v_mfma_f32_16x16x32_bf16, two partial products fused
// along the instruction's K = 32 (see "S16" at the fragment loads): the same flops as six v_mfma_f32_32x32x16_bf16 per 32 x 32
// block, in twice as many instructions of half the length.  It needs MORE core cycles per K tile (3072 of them are MFMA cycles
// either way; in-kernel stamps, profiles/r06_bx16_timeline*.log: 3440-3470 against 3250-3280, 40 fragment reads instead of 27)
// -- and runs faster, because the kernel is power-limited and the chip holds a higher clock on this shape (2.0-2.1 against 1.8 GHz
// on half-zero data, 1.8 against 1.7 on N(0,1); MI355X_MICROARCH.md, DVFS give-back item 7): the headline-shaped SYRK takes
// 875 / 780 ms (N(0,1) / half zeros) against 921 / 819 ms on the same box (profiles/r06_syrk_ab_s16.log).  -DBX_SHAPE16=0 builds
// the 32 x 32 x 16 form (its own asm block, BX_KLOOP_ASM32) for such comparisons.
//
// What the blocks do differently from the compiler's schedule, in core cycles per K tile (32 x 32 x 16 form, where they were
// measured one by one; profiles/r06_bx_attribution*.log; the C++ loop: 3525):
//   * the twelve global -> LDS requests take a scalar base + ONE 32-bit lane offset instead of twelve 64-bit per-lane
//     pointers (global_load_lds_dwordx4 v, s[..]): the requests cost ~30 cycles per tile instead of ~290 -- it is the address
//     registers of a request, not its issue slot, that hold up the SIMD (eight waves, two per SIMD, did not hide it: same 3500);
//   * never more than two ds_read_b128 per 32-cycle MFMA gap (a third one by every wave saturates the LDS array for that gap: ~100)
//     -- one per 16-cycle gap in the 16 x 16 x 32 form;
//   * one request per gap over the second half of the tile, never beside fragment reads (all in the last row: + 190; one per gap
//     right behind the barrier, 16 x 16 x 32 form: + 370);
//   * an accumulator comes back every 8th instruction at the earliest (16 x 16 x 32 form: every 2nd costs ~ 25);
//   * three tiles per trip with the stage registers renamed instead of rotated (- 45); M0 written one gap ahead of its request
//     instead of s_nop in front of it (- 30).
//   32 x 32 x 16 form: 3283 (no barrier: 3235; no requests: 3253; neither: 3226).  16 x 16 x 32 form: 3440 (no barrier: 3400; no
//   requests: 3380; neither: 3350).  Fewer cycles come back as time only in part (DVFS give-back): - 7 % cycles were - 3.8 % time.
template <int NPROD, bool ASM = false>
__global__ __launch_bounds__(256, 1) void gemm256_bx_kernel(GemmBxArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_bx[];
  constexpr bool S16 = NPROD == 6 && BX_SHAPE16 != 0;   // the MFMA shape (accumulator layout below)
#if defined(BX_STAMP) && BX_STAMP == 2   // timeline build (scripts/probe/bx_timeline.py): 8 words per workgroup
  const unsigned long long stamp_entry = __builtin_amdgcn_s_memrealtime();
#endif
#if !(defined(BX_EXP) && BX_EXP == 2)   // (experiment 2: timing without the gate check)
  if (p.gate && (*p.gate & p.gate_mask) != 0) return;  // the fp32 MFMA kernel takes this chunk
#endif
  int ti, tj, zsplit;
  {
    const int nt_all = (int)(p.K / BK);
    if (!map_tile_z(p.syrk, p.sbw, p.tiles_m, p.tiles_n, p.kt_split > 0 ? (nt_all + p.kt_split - 1) / p.kt_split : 1, ti, tj, zsplit)) return;
  }

  const int tid = threadIdx.x;
  if (p.sync) {
    // The 32 workgroups of an XCD group stream the same 12 operand panels; they hit in that XCD's L2 only while they are
    // within ~14 K tiles of each other.  Workgroups start when a CU frees up, so the start times of a group random-walk
    // apart over the rounds of a launch.  Start barrier: wait (at most 50 us: no deadlock if fewer CUs are available)
    // until the whole group has arrived.  Older groups never wait for newer ones.
    if (tid == 0) {
      int group;
      const int expect = xcd_group(p.syrk, p.sbw, p.tiles_m, p.tiles_n, group);
      int *cnt = p.sync + (int64_t)blockIdx.y * (gridDim.x >> 5) + group;
      __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect &&
             __builtin_amdgcn_s_memrealtime() - t0 < 5000ull)
        __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
  }
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  // Accumulator layout.  The six-product kernel computes on v_mfma_f32_16x16x32_bf16 (S16): the 32 x 32 block acc[i][j] is four
  // 16 x 16 tiles (tr, tc) in registers 4 (2 tr + tc) .. + 3, lane l holding rows 4 (l / 16) .. + 3 of column l % 16 of each.
  // (S16T, -DBX_S16_TRANSPOSED=1: with the B fragment as the instruction's first operand a lane holds row l % 16 and the four
  // CONSECUTIVE columns 4 (l / 16) .. + 3 instead -- bit-identical sums, one 16-byte access per tile in the flush and a mirrored
  // store that is coalesced as it stands; measured SLOWER on the same box, profiles/r06_syrk_ab_transposed.log: headline-shaped
  // SYRK 875 / 783 against 866 / 772 ms, rank-1024 update of 20480^2 4.18 against 3.94 ms.)
  // The other kernels on v_mfma_f32_32x32x16_bf16 (register e = rows (e & 3) + 8 (e >> 2) + 4 (l / 32) of column l % 32).
  const int r16 = lane & 15, kb = lane >> 4;
  // element e of a block: (row, column) = (lrow + erc(e), lcol + ecc(e)), a lane part and a part that is a constant per register
  constexpr bool S16T = S16 && BX_S16_TRANSPOSED != 0;
  const int lrow = S16T ? r16 : S16 ? 4 * kb : 4 * h, lcol = S16T ? 4 * kb : S16 ? r16 : r;
  auto erc = [](int e) __attribute__((always_inline)) -> int {
    return S16T ? 16 * (e >> 3) : S16 ? 16 * (e >> 3) + (e & 3) : (e & 3) + 8 * (e >> 2);
  };
  auto ecc = [](int e) __attribute__((always_inline)) -> int { return S16T ? 16 * ((e >> 2) & 1) + (e & 3) : S16 ? 16 * ((e >> 2) & 1) : 0; };
  auto erow = [&](int e) __attribute__((always_inline)) -> int { return lrow + erc(e); };
  auto ecol = [&](int e) __attribute__((always_inline)) -> int { return lcol + ecc(e); };
  const int64_t row0 = (int64_t)ti * B2, col0 = (int64_t)tj * B2;
  int nt = (int)(p.K / BK);
  int64_t kt0 = 0;
  if (p.kt_split > 0) {
    kt0 = (int64_t)zsplit * p.kt_split;
    nt = nt - (int)kt0 < p.kt_split ? nt - (int)kt0 : p.kt_split;
  }

  // (S16: see the accumulator layout above -- the 32 x 32 block is four separate 4-register accumulators)
  constexpr int NQ = S16 ? 4 : 1, QW = 16 / NQ;
  typedef typename std::conditional<S16, f32x4, f32x16>::type AccV;
  AccV acc[4][4][NQ];
  auto aget = [&](int i, int j, int e) __attribute__((always_inline)) -> float { return acc[i][j][e / QW][e % QW]; };
  auto aset = [&](int i, int j, int e, float v) __attribute__((always_inline)) { acc[i][j][e / QW][e % QW] = v; };
  auto clear_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) aset(i, j, e, 0.f);
  };
  clear_acc();

  gptr Cout = p.kt_split > 0 ? (gptr)p.slab + (int64_t)zsplit * p.M * p.N : (gptr)p.C;
  const int64_t ldc = p.kt_split > 0 ? p.N : p.ldc;
  const float alpha_ = p.kt_split > 0 ? 1.f : p.alpha, beta_ = p.kt_split > 0 ? 0.f : p.beta;
  const bool full_tile = row0 + B2 <= p.M && col0 + B2 <= p.N;
  // C <- C' + alpha * acc with C' = beta * C on the first flush and C afterwards; acc <- final value.  The loads go to
  // the L2 (sc1): earlier chains of this tile were added into C by L2 atomics (flush_mid), which the L1 does not see.
  auto ld_l2 = [](gptr q) __attribute__((always_inline)) -> float {
    return __hip_atomic_load((const float *)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  typedef f32x4 __attribute__((address_space(1))) *gptr4w;
  const bool c_vec = (reinterpret_cast<uintptr_t>(Cout) & 15) == 0 && (ldc & 3) == 0 && (col0 & 3) == 0;
  auto flush_to_c = [&](bool first) __attribute__((always_inline)) {
    const float beta = first ? beta_ : 1.f;
    int opaque = 0;
    __asm__ volatile("" : "+v"(opaque));  // keeps the 256 output addresses out of LICM's reach (see gemm256_kernel)
    // BX_FLUSH_ROWS rows of four 32 x 32 blocks at a time: 64 loads per row in flight, then as many stores (block by block - 16
    // loads, wait, 16 stores - the read-modify-write cost ~40 us per 256 x 256 tile: half of a K = 512 update's time)
    constexpr int NR = BX_FLUSH_ROWS;
    // (the 16-byte loads of the vector path are plain loads: earlier chains of this tile went into C by L2 atomics, which the L1
    // does not see -- drop its lines first)
    if (S16T && !first) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
    for (int i0 = 0; i0 < 4; i0 += NR) {
      __asm__ volatile("" ::: "memory");
      const int64_t rbase0 = row0 + wm * 128 + i0 * 32 + lrow + opaque;
      if (S16T && full_tile && c_vec) {
        // a lane's four values of a 16 x 16 tile are consecutive in its row of C: one 16-byte load and store per tile
        f32x4 old[NR][4][4];
        if (beta != 0.f) {
#pragma unroll
          for (int ii = 0; ii < NR; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              gptr cbase = Cout + (rbase0 + ii * 32) * ldc + (col0 + wn * 128 + j * 32 + lcol);
#pragma unroll
              for (int q = 0; q < 4; ++q) old[ii][j][q] = *(gptr4w)(cbase + (int64_t)erc(4 * q) * ldc + ecc(4 * q));
            }
        }
#pragma unroll
        for (int ii = 0; ii < NR; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            gptr cbase = Cout + (rbase0 + ii * 32) * ldc + (col0 + wn * 128 + j * 32 + lcol);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x4 v;
#pragma unroll
              for (int e4 = 0; e4 < 4; ++e4) v[e4] = alpha_ * aget(i0 + ii, j, 4 * q + e4);
              if (beta != 0.f) v += beta * old[ii][j][q];
              *(gptr4w)(cbase + (int64_t)erc(4 * q) * ldc + ecc(4 * q)) = v;
#pragma unroll
              for (int e4 = 0; e4 < 4; ++e4) aset(i0 + ii, j, 4 * q + e4, v[e4]);
            }
          }
      } else if (full_tile) {
        float old[NR][4][16];
        if (beta != 0.f) {
#pragma unroll
          for (int ii = 0; ii < NR; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              gptr cbase = Cout + (rbase0 + ii * 32) * ldc + (col0 + wn * 128 + j * 32 + lcol);
#pragma unroll
              for (int e = 0; e < 16; ++e) old[ii][j][e] = ld_l2(cbase + (int64_t)erc(e) * ldc + ecc(e));
            }
        }
#pragma unroll
        for (int ii = 0; ii < NR; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            gptr cbase = Cout + (rbase0 + ii * 32) * ldc + (col0 + wn * 128 + j * 32 + lcol);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              float v = alpha_ * aget(i0 + ii, j, e);
              if (beta != 0.f) v += beta * old[ii][j][e];
              cbase[(int64_t)erc(e) * ldc + ecc(e)] = v;
              aset(i0 + ii, j, e, v);
            }
          }
      } else {
#pragma unroll
        for (int ii = 0; ii < NR; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int64_t row = rbase0 + ii * 32 + erc(e), col = col0 + wn * 128 + j * 32 + ecol(e);
              float v = alpha_ * aget(i0 + ii, j, e);
              if (row < p.M && col < p.N) {
                gptr c = Cout + row * ldc + col;
                if (beta != 0.f) v += beta * ld_l2(c);
                *c = v;
              }
              aset(i0 + ii, j, e, v);
            }
          }
      }
    }
  };

  // The flush INSIDE the K loop (end of an accumulation chain): C += alpha * acc with no-return fp32 atomics executed in
  // the L2 (global_atomic_add_f32: a correctly rounded fp32 add, exactly the VALU add of the other flushes), or a plain
  // store when this is the tile's first flush and beta = 0 (the host passes only beta = 0 or 1 and scales C beforehand
  // otherwise).  This workgroup is the only writer of its tile and a wave's memory operations on one address stay in
  // order, so the result is the same deterministic sum -- but nothing is loaded and nothing is waited for: the
  // read-modify-write happens where the data lives while the next chain's MFMAs run, and the global -> LDS pipeline is
  // not restarted (that alone cost 34 us per chain).  (Out of line -- accumulators copied to a private array, a noinline
  // function issuing the atomics -- the flush cost 140 us: 512 KB of scratch traffic per workgroup.)
  auto flush_mid = [&](bool store) __attribute__((always_inline)) {
    int opaque = 0;
    __asm__ volatile("" : "+v"(opaque));
#if defined(BX_VARIANT) && BX_VARIANT == 2   // timing only: no flush at all
    return;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t rb = row0 + wm * 128 + i * 32 + lrow + opaque, cb0 = col0 + wn * 128 + j * 32 + lcol;
        gptr cb = Cout + rb * ldc + cb0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int dr = erc(e), dc = ecc(e);
          if (full_tile || (rb + dr < p.M && cb0 + dc < p.N)) {
            if (store) cb[(int64_t)dr * ldc + dc] = alpha_ * aget(i, j, e);
            else __builtin_amdgcn_global_atomic_fadd_f32(cb + (int64_t)dr * ldc + dc, alpha_ * aget(i, j, e));
          }
        }
      }
  };

  // ---- DMA sources: wave w fills blocks w and w + 4 of every piece of both operands (12 instructions per K tile),
  // each instruction 1 KB of consecutive global bytes (blocked piece layout).  Row blocks beyond the matrix read the
  // last block (their outputs are never stored).  The pointers advance by one k tile per request.
  gcptr16 srcA[2], srcB[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int blk = wave + 4 * u;
    int64_t ba = row0 / 32 + blk, bb = col0 / 32 + blk;   // 32-row block of the operand (clamped: never stored rows)
    ba = ba < p.nrbA ? ba : p.nrbA - 1;
    bb = bb < p.nrbB ? bb : p.nrbB - 1;
    srcA[u] = (gcptr16)p.A + (kt0 * p.nrbA + ba) * 512 + 8 * lane;          // 1 KB block = 512 bf16; lane -> its 16 bytes
    srcB[u] = (gcptr16)p.B + (kt0 * p.nrbB + bb) * 512 + 8 * lane;
  }
  const int64_t stepA = p.nrbA * 512, stepB = p.nrbB * 512;  // one k tile further
  // (The addresses stay per-lane 64-bit registers, advanced by one v_lshl_add_u64 per request.  The scalar form -- block address in an
  // SGPR pair advanced by s_add_u32 / s_addc_u32, one shared 32-bit lane offset, `global_load_lds_dwordx4 v, s[..]` -- measured 1.5-4 %
  // SLOWER on the SYRK shape (225.1 / 248.8 against 227-234 / 252.5 TFLOP/s, round 5, scripts/probe/syrk_ab.sh).)
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_bx +
                                                       (unsigned)(wave * 1024));
  auto dma16b = [&](gcptr16 src, unsigned lds_byte_addr) __attribute__((always_inline)) {
    __asm__ volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory");
  };
  // part q of the 12 requests of one K tile: q = 0..3 -> (block u = q / 2, operand q % 2), three pieces each
  auto issue_part = [&](int st, int q) __attribute__((always_inline)) {
    const int u = q >> 1;
    if ((q & 1) == 0) {
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        dma16b(srcA[u] + pc * p.strideA, lds0 + (unsigned)(st * BX_STAGE + pc * BX_PIECE + 4 * u * 1024));
      srcA[u] += stepA;
    } else {
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        dma16b(srcB[u] + pc * p.strideB, lds0 + (unsigned)(st * BX_STAGE + BX_OPER + pc * BX_PIECE + 4 * u * 1024));
      srcB[u] += stepB;
    }
  };
  auto issue = [&](int st) __attribute__((always_inline)) {  // the next not yet requested K tile into stage st
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_part(st, q);
  };
  // ---- one K tile of 16: 16 output tiles x NPROD bf16 MFMAs per wave; the smallest partial products go in first.
  // The B pieces of the wave's four column tiles stay in registers for the tile (48), the A pieces stream per row tile.
  // S16: one v_mfma_f32_16x16x32_bf16 adds TWO partial products -- its K = 32 is the tile's 16 k twice, first with one pair of
  // pieces and then with another.  Lane l supplies row (column) l % 16 and k block l / 16 of the instruction's 32: blocks 0, 1 are
  // the two 8-k halves of the first piece, blocks 2, 3 those of the second, so a fragment is still ONE ds_read_b128 with the piece
  // picked per lane.  Three instructions per 16 x 16 tile and K tile, smallest terms first:
  //   [a2 | a0] x [b0 | b2]  ->  lo hi + hi lo        [a1 | a1] x [b1 | b0]  ->  mid mid + mid hi        [a0 | a0] x [b1 | b0]  ->  hi mid + hi hi
  // i.e. three A fragments per 16 rows and two B fragments per 16 columns (the B fragments of the wave's 128 columns stay in
  // registers for the tile: 64).  40 fragment reads per wave and K tile against 28 for the 32 x 32 x 16 shape, 192 instructions of
  // 8 passes against 96 of 16 -- same flops, but the chip holds a ~12 % higher clock on this shape (profiles/r06_bx_mfma16_timing.log).
  const unsigned fofsA = S16 ? (unsigned)((wm * 4) * 1024 + (kb & 1) * 512 + r16 * 16) : (unsigned)((wm * 4) * 1024 + h * 512 + r * 16);
  const unsigned fofsB = S16 ? (unsigned)((wn * 4) * 1024 + (kb & 1) * 512 + r16 * 16) : (unsigned)((wn * 4) * 1024 + h * 512 + r * 16);
  constexpr int NCA = 3, NCB = S16 ? 2 : 3, NSUB = S16 ? 2 : 1;   // fragments per 32-row block: NCA (NCB) combinations x NSUB halves
  // byte offset of combination c of A (d of B) inside a stage: the piece this lane reads
  unsigned cofsA[NCA], cofsB[NCB];
  if (S16) {
    cofsA[0] = (unsigned)((kb >> 1 ? 0 : 2) * BX_PIECE) + fofsA;   // [a2 | a0]
    cofsA[1] = (unsigned)(1 * BX_PIECE) + fofsA;                   // [a1 | a1]
    cofsA[2] = fofsA;                                              // [a0 | a0]
    cofsB[0] = (unsigned)(BX_OPER + (kb >> 1 ? 2 : 0) * BX_PIECE) + fofsB;   // [b0 | b2]
    cofsB[1] = (unsigned)(BX_OPER + (kb >> 1 ? 0 : 1) * BX_PIECE) + fofsB;   // [b1 | b0]
  } else {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
      cofsA[pc] = (unsigned)(pc * BX_PIECE) + fofsA;
      if (pc < NCB) cofsB[pc] = (unsigned)(BX_OPER + pc * BX_PIECE) + fofsB;
    }
  }
  struct FragB {
    bf16x8 v[NCB][4 * NSUB];
  };
  struct FragA {
    bf16x8 v[NCA][NSUB];
  };
  auto load_b_col = [&](int st, int j, FragB &f) __attribute__((always_inline)) {   // the fragments of 32-column block j
    const unsigned char *sS = smem_bx + st * BX_STAGE;
#pragma unroll
    for (int d = 0; d < NCB; ++d)
#pragma unroll
      for (int tc = 0; tc < NSUB; ++tc) f.v[d][NSUB * j + tc] = *reinterpret_cast<const bf16x8 *>(sS + cofsB[d] + j * 1024 + tc * 256);
  };
  auto load_b = [&](int st) __attribute__((always_inline)) -> FragB {
    FragB f;
#pragma unroll
    for (int j = 0; j < 4; ++j) load_b_col(st, j, f);
    return f;
  };
  auto load_a = [&](int st, int i) __attribute__((always_inline)) -> FragA {   // the fragments of 32-row block i
    const unsigned char *sS = smem_bx + st * BX_STAGE;
    FragA f;
#pragma unroll
    for (int c = 0; c < NCA; ++c)
#pragma unroll
      for (int tr = 0; tr < NSUB; ++tr) f.v[c][tr] = *reinterpret_cast<const bf16x8 *>(sS + cofsA[c] + i * 1024 + tr * 256);
    return f;
  };
  auto mfma_row = [&](auto iconst, const FragA &fa, const FragB &fb, auto &&between) __attribute__((always_inline)) {
    constexpr int i = decltype(iconst)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      between(j);
      if constexpr (S16) {
#pragma unroll
        for (int m = 0; m < 3; ++m)   // (the four tiles in turn: consecutive instructions never share an accumulator)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[i][j][t] = S16T ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb.v[m == 0 ? 0 : 1][2 * j + (t & 1)], fa.v[m][t >> 1], acc[i][j][t], 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v[m][t >> 1], fb.v[m == 0 ? 0 : 1][2 * j + (t & 1)], acc[i][j][t], 0, 0, 0);
      } else {
        f32x16 c = acc[i][j][0];
        if (NPROD >= 9) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v[2][0], fb.v[2][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v[2][0], fb.v[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v[1][0], fb.v[2][j], c, 0, 0, 0);
        }
        if (NPROD >= 6) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v[2][0], fb.v[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v[0][0], fb.v[2][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v[1][0], fb.v[1][j], c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v[1][0], fb.v[0][j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v[0][0], fb.v[1][j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v[0][0], fb.v[0][j], c, 0, 0, 0);
        acc[i][j][0] = c;
      }
    }
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>;
  using J3 = std::integral_constant<int, 3>;
  // Pipeline (tile t lives in stage t % 3).  The fragment reads of a row tile are issued BEFORE the MFMAs of the
  // previous one, and the first fragments of tile t + 1 during the second half of tile t, so the matrix pipe never
  // waits for LDS.  In the middle of tile t every wave waits for its own share of tile t + 1 (requested one tile ago);
  // the barrier there publishes tile t + 1 and certifies that every wave is completely past tile t - 1, whose stage then
  // receives the requests for tile t + 2.
  // One pipelined pass over all K tiles.  Every `flush_tiles` tiles (rounded to the two-tile trip) an MFMA chain ends:
  // its sum goes into C (flush_mid) and the accumulators restart from zero -- see bx_flush_tiles for why chains are short.
#if defined(BX_STAMP)
  const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  bool first_flush = true;
  const int flush_tiles = (((p.syrk != 0 && ti == tj) ? p.flush_diag : p.flush_tiles) + 1) & ~1;
  const bool mirrored = p.syrk == 1 && ti != tj;   // the mirror store wants the final values in the accumulators
  {
    const int t1 = nt;
    // (tile 1 is requested behind tile 0 BEFORE the wait for tile 0: its latency runs beside tile 0's instead of behind the barrier)
    issue(0);
    if (1 < t1) {
      issue(1);
      __asm__ volatile("s_waitcnt vmcnt(12)" ::: "memory");   // all but the 12 requests of tile 1: tile 0 has landed
    } else {
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __asm__ volatile("s_barrier" ::: "memory");
#if defined(BX_STAMP) && BX_STAMP == 2
    const unsigned long long stamp_loop0 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && g_bx_stamp && p.syrk == 2 && blockIdx.x < g_bx_stamp_cap) g_bx_stamp[8 * blockIdx.x + 3] = stamp_loop0;
#endif
    // two named fragment sets in ping-pong (tile t uses one and fills the other for tile t + 1)
    FragB fbX = load_b(0), fbY;
    FragA faX = load_a(0, 0), faY;
    int st = 0, t = 0, next_flush = flush_tiles;
    auto tile = [&](const FragB &fb, const FragA &fa, FragB &fbn, FragA &fan) __attribute__((always_inline)) {
      const int st1 = st == 2 ? 0 : st + 1, st2 = st == 0 ? 2 : st - 1;   // stages of tiles t + 1 and t + 2 (= t - 1)
      const int stn = t + 1 < t1 ? st1 : st;                              // (last tile: harmless re-read of its own stage)
      auto nothing = [](int) {};
      FragA fa1 = load_a(st, 1);
      mfma_row(J0{}, fa, fb, nothing);
      FragA fa2 = load_a(st, 2);
      mfma_row(J1{}, fa1, fb, nothing);
      __builtin_amdgcn_sched_barrier(0);
      // own share of tile t + 1 has landed.  The wait is the compiler-VISIBLE builtin on purpose: hipcc cannot see the asm DMA
      // requests, but it does track its own memory operations (accumulator rows it keeps in scratch around a flush, spill
      // reloads in front of the loop); with those pending in its model it put s_waitcnt vmcnt(12/8/4/0) in front of the first
      // MFMAs of every second tile -- where the hardware counter also holds the DMA requests just issued (+ 40 us per 128 K
      // tiles).  Seeing this vmcnt(0) it knows nothing of its own is pending afterwards.
      __asm__ volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      __asm__ volatile("s_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      FragA fa3 = load_a(st, 3);
      // The 12 requests of tile t + 2 go out three at a time in ROW 3, behind the fragment reads of each column tile -- away
      // from the barrier: a burst right behind it stalls the wave at issue while the matrix pipe runs dry (round 2), and three
      // per column tile of row 2 (rounds 2-4), still within ~800 cycles of the release, cost 3 % of the kernel.  Same-box A/B
      // at n = 40 960, P = 131 072 (scripts/probe/syrk_ab.sh, profiles/r05_syrk_ab*.log), N(0,1) / half-zero data: row 2
      // (rounds 2-4) 225.1 / 246.3 TFLOP/s; two per column tile of row 2 + one per column tile of row 3 231.0 / 254.0; row 3
      // (this) 233.4 / 256.4 (238.3 / 263.6 against 231.8 / 257.2 on a faster box); row 3, one behind every second MFMA
      // 233.2 / 251.0; rows 2 + 3, one behind every fourth MFMA 232.3 / 250.6; row 3 with the wait + barrier moved between rows 2
      // and 3 234.1 / 259.4 against 238.3 / 263.6 on that box.  The requests then have rows 0 and 1 of the next tile (48 MFMAs,
      // ~0.9 us) + what is left of row 3 to land before the mid-tile wait.
      const bool req = t + 2 < t1;
      mfma_row(J2{}, fa2, fb, nothing);
      // first fragments of tile t + 1: the B pieces of column tile j behind the MFMAs of column tile j - 1 of row 3
      {
        mfma_row(J3{}, fa3, fb, [&](int j) __attribute__((always_inline)) {
          load_b_col(stn, j, fbn);
          __builtin_amdgcn_sched_barrier(0);
          if (req) issue_part(st2, j);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      fan = load_a(stn, 0);
      st = st1;
      ++t;
    };
    while (true) {
      const int tc = next_flush < t1 ? next_flush : t1;
      if constexpr (ASM && NPROD == 6) {
        // every tile of the asm block requests tile t + 2: it runs up to the last two tiles of the pass (the chain ends of this
        // loop stay where they are)
        // (a whole number of the block's trips AND of this loop's two-tile trips)
        int na = (tc < t1 - 2 ? tc : t1 - 2) - t;
        na -= na % (2 * BX_KLOOP_UNROLL);
        if (na > 0) {
          const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_bx;
          const int64_t strideA_b = 2 * p.strideA, strideB_b = 2 * p.strideB, stepA_b = 2 * stepA, stepB_b = 2 * stepB;
          __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the fragment sets of the C++ loop are not carried into the block)
          if constexpr (S16) {
            // the block names its accumulators: acc[i][j][q] is pinned to a[16 (4 i + j) + 4 q .. + 3]
            __asm__ volatile(BX_KLOOP_TEXT16
                             : BX_KLOOP_ASM_ACC(acc)
                             : "v"(lds_base + cofsA[0]), "v"(lds_base + cofsA[1]), "v"(lds_base + cofsA[2]), "v"(lds_base + cofsB[0]),
                               "v"(lds_base + cofsB[1]), "v"(srcA[0]), "v"(srcA[1]), "v"(srcB[0]), "v"(srcB[1]), "s"(strideA_b), "s"(strideB_b),
                               "s"(stepA_b), "s"(stepB_b), "s"(lds0), "s"(st), "s"(na), "s"(0)
                             : BX_KLOOP_CLOB16);
          } else {
            const unsigned fa_lds = lds_base + fofsA, fb_lds = lds_base + BX_OPER + fofsB;
            __asm__ volatile(BX_KLOOP_TEXT32
                             : "+a"(acc[0][0][0]), "+a"(acc[0][1][0]), "+a"(acc[0][2][0]), "+a"(acc[0][3][0]), "+a"(acc[1][0][0]), "+a"(acc[1][1][0]),
                               "+a"(acc[1][2][0]), "+a"(acc[1][3][0]), "+a"(acc[2][0][0]), "+a"(acc[2][1][0]), "+a"(acc[2][2][0]), "+a"(acc[2][3][0]),
                               "+a"(acc[3][0][0]), "+a"(acc[3][1][0]), "+a"(acc[3][2][0]), "+a"(acc[3][3][0])
                             : "v"(fa_lds), "v"(fb_lds), "v"(srcA[0]), "v"(srcA[1]), "v"(srcB[0]), "v"(srcB[1]), "s"(strideA_b), "s"(strideB_b),
                               "s"(stepA_b), "s"(stepB_b), "s"(lds0), "s"(st), "s"(na), "s"(0)
                             : BX_KLOOP_CLOB32);
          }
          // the block leaves the request pointers na tiles further and the stage of the new tile t: redo both here (cheap,
          // and the operands above stay plain inputs -- 16 read-write accumulator operands already count twice)
          srcA[0] += (int64_t)na * stepA; srcA[1] += (int64_t)na * stepA;
          srcB[0] += (int64_t)na * stepB; srcB[1] += (int64_t)na * stepB;
          st = (st + na) % 3;
          t += na;
          fbX = load_b(st);
          faX = load_a(st, 0);
        }
      }
      while (t + 2 <= tc) {  // two tiles per trip, no exit in between (the sets swap roles and are back in place)
        tile(fbX, faX, fbY, faY);
        tile(fbY, faY, fbX, faX);
      }
      if (t + 2 > t1) break;  // at most one tile left: it joins this chain
      // end of a chain; its memory operations drain behind the next tile's MFMAs.  (As a register read-modify-write like the last
      // flush: the same within 2 % on split-K Gram matrices and on a five-chain product, profiles/r06_splitk_compact.log.)
      flush_mid(first_flush && beta_ == 0.f);
      first_flush = false;
      clear_acc();
      next_flush += flush_tiles;
      // the fragments of tile t are read again (nothing but the accumulators is carried across the flush)
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      fbX = load_b(st);
      faX = load_a(st, 0);
    }
    if (t < t1) tile(fbX, faX, fbY, faY);
  }
#if defined(BX_STAMP)
#if BX_STAMP == 2   // (the launches that add a chunk into the lower tiles without mirroring: all but the last of a Gram SYRK)
  if (tid == 0 && g_bx_stamp && p.syrk == 2 && blockIdx.x < g_bx_stamp_cap) {
    int hwid, xcc;
    __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long *w = g_bx_stamp + 8 * blockIdx.x;
    w[0] = __builtin_amdgcn_s_memtime() - stamp_c0;
    w[1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
    w[2] = stamp_entry;
    w[4] = __builtin_amdgcn_s_memrealtime();   // end of the K loop
    w[6] = ((unsigned long long)(unsigned)xcc << 32) | (unsigned)hwid;
  }
#else
  if (tid == 0 && g_bx_stamp && blockIdx.y == 0 && blockIdx.x < g_bx_stamp_cap) {
    g_bx_stamp[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - stamp_c0;
    g_bx_stamp[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
  }
#endif
#endif
  // The LAST flush of a full, 16-byte-aligned tile that must read C (beta != 0 on a single-chain product, or a SYRK tile
  // whose final values also go to the transposed tile).  The operand stages are dead now, so the old values of C are
  // fetched by the same global -> LDS DMA as the operands, two 32 x 32 tiles (8 KB) per wave and request, three requests
  // (24 KB per wave, no registers) in flight while a pair of tiles is combined and stored (a plain read-modify-write in
  // registers holds 64 four-byte loads = 16 KB in flight).  `mirror`: the final values are written back to the patch and read
  // column-wise: lane r stores element (row r, col c) to C[col0 + c][row0 + r], 128 contiguous bytes per mirror row.
  auto flush_final = [&](bool first, bool mirror) __attribute__((always_inline)) {
    const float beta = first ? beta_ : 1.f;
    float *ts = reinterpret_cast<float *>(smem_bx + BX_PATCH_OFFSET) + wave * 1024;
    const int rr = lane >> 3, c4 = lane & 7;
    int opaque = 0;
    __asm__ volatile("" : "+v"(opaque));
    gptr cwave = Cout + (row0 + wm * 128 + rr + opaque) * ldc + (col0 + wn * 128 + 4 * c4);
    const bool pre = beta != 0.f;
    const unsigned oldb = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(wave * (32768 - 1024)));   // this wave's 4 x 8 KB of the stages
    const unsigned char *oldp = smem_bx + wave * 32768 + lane * 16;
    auto issue_pair = [&](int k) __attribute__((always_inline)) {   // old values of tiles 2 k, 2 k + 1 into buffer k & 3
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int u = 2 * k + t;
          dma16b((gcptr16)(const void __attribute__((address_space(1))) *)(cwave + (int64_t)((u >> 2) * 32 + 8 * it) * ldc + (u & 3) * 32),
                 oldb + (unsigned)((k & 3) * 8192 + t * 4096 + it * 1024));
        }
    };
    auto wait_vm = [](int n) __attribute__((always_inline)) {
      switch (n) {
        case 16: __asm__ volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 24: __asm__ volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 32: __asm__ volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        case 56: __asm__ volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
        default: __asm__ volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
      }
    };
    __syncthreads();   // every wave is past its last fragment reads: the stages may be overwritten
    if (pre) {
      // earlier chains of this tile may have been added into C by L2 atomics: the DMA must not be served from a stale L1 line
      if (!first) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      issue_pair(0); issue_pair(1); issue_pair(2);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (pre) {
        // operations issued after the requests of pair k: the later requests (8 each) and the stores of the pairs in
        // between (8 per pair, + 32 mirror stores); vmcnt counts in issue order
        const int S = mirror ? 40 : 8;
        const int after = k == 0 ? 16 : k == 1 ? 16 + S : k < 6 ? 16 + 2 * S : k == 6 ? 8 + 2 * S : 2 * S;
        wait_vm(after < 24 ? 16 : after < 32 ? 24 : after < 56 ? 32 : after < 63 ? 56 : 63);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int u = 2 * k + t;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = erow(e);
#if defined(BX_FLUSH_EXP) && BX_FLUSH_EXP == 1   // timing only: no patch writes
          if (p.alpha == 12345.678f)
#endif
          ts[row * 32 + (ecol(e) ^ ((row & BX_SWZ_MASK) << 2))] = aget(u >> 2, u & 3, e);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          float *tp = ts + (8 * it + rr) * 32 + 4 * (c4 ^ (rr & BX_SWZ_MASK));
          f32x4 v = *reinterpret_cast<const f32x4 *>(tp);
          v = alpha_ * v;
          if (pre) v += beta * *reinterpret_cast<const f32x4 *>(oldp + (k & 3) * 8192 + t * 4096 + it * 1024);
          *(gptr4w)(cwave + (int64_t)((u >> 2) * 32 + 8 * it) * ldc + (u & 3) * 32) = v;
          if (mirror) *reinterpret_cast<f32x4 *>(tp) = v;
        }
        if (mirror) {   // transposed copy of tile u = (i, j)
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          gptr mbase = (gptr)p.C + (col0 + wn * 128 + (u & 3) * 32 + h) * p.ldc + (row0 + wm * 128 + (u >> 2) * 32 + r);
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int col = 2 * q + h;
            mbase[(int64_t)(2 * q) * p.ldc] = ts[r * 32 + (col ^ ((r & BX_SWZ_MASK) << 2))];
          }
        }
      }
      if (pre && k + 3 < 8) {
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS reads of buffer (k + 3) & 3 == (k - 1) & 3 are done
        issue_pair(k + 3);
      }
    }
  };
  // Which last flush: since the asm K loop the register read-modify-write (flush_to_c) wins -- inside flush_final hipcc reloads
  // spilled lane constants from scratch and waits for them with vmcnt(0), which also waits for every old-value request in flight
  // (same box, headline-shaped SYRK on N(0,1) / half-zero data and a rank-1024 update of 20480^2, profiles/r06_syrk_ab_flush.log:
  // flush_final 893 / 801 ms, 4.47 ms; flush_to_c 867 / 773 ms, 3.97 ms; flush_final for mirrored tiles only 916 / 807 ms, 5.04 ms).
  // -DBX_FINAL_DMA=1 / 2 builds the LDS-DMA flush for every full tile / for mirrored tiles only.
#if defined(BX_FINAL_DMA) && BX_FINAL_DMA == 2
  const bool fast_tile = full_tile && c_vec && mirrored;
#elif defined(BX_FINAL_DMA) && BX_FINAL_DMA == 1
  const bool fast_tile = full_tile && c_vec;
#else
  const bool fast_tile = false;
#endif
#if defined(BX_EXP) && BX_EXP == 1   // timing only: no final flush
  if (p.alpha == 12345.678f)
#endif
#if defined(BX_FINAL_ATOMIC) && BX_FINAL_ATOMIC
  // experiment: C += alpha acc by no-return L2 atomics whenever the launch accumulates (beta = 1) and nothing needs the final values.
  // Nothing is loaded or waited for -- and it is slower: the 1024 atomic instructions of a tile take 43 us to issue against 18 us for the
  // read-modify-write (headline-shaped SYRK 865 / 759 against 843 / 744 ms, rank-1024 update 4.62 against 3.84 ms; profiles/r06_syrk_ab_atomic.log)
  if (!mirrored && (!first_flush || beta_ == 1.f))
    flush_mid(false);
  else
#endif
  if (!fast_tile)
    flush_to_c(first_flush);          // edge tiles / unaligned C; the mirror store below takes the values from the registers
  else if (mirrored || first_flush)
    flush_final(first_flush, mirrored);   // final values to the tile (and, for a SYRK, to its transposed image)
  else
    flush_mid(false);

  if (S16T && mirrored && !fast_tile) {
    // the transposed image: for a fixed register the lanes of a row group hold 16 consecutive ROWS of one column of the tile,
    // i.e. 64 consecutive bytes of the image's row -- stored as they are
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t mcol = row0 + wm * 128 + i * 32 + lrow, mrow0 = col0 + wn * 128 + j * 32 + lcol;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t mr = mrow0 + ecc(e), mc = mcol + erc(e);
          if (mr < p.N && mc < p.M) *((gptr)p.C + mr * p.ldc + mc) = aget(i, j, e);
        }
      }
  } else if (mirrored && !fast_tile) {
    __syncthreads();
    float *ts = reinterpret_cast<float *>(smem_bx) + wave * (32 * 33);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int e = 0; e < 16; ++e) ts[ecol(e) * 33 + erow(e)] = aget(i, j, e);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int64_t mrow0 = col0 + wn * 128 + j * 32;
        const int64_t mcol = row0 + wm * 128 + i * 32 + r;
#pragma unroll
        for (int rr = 0; rr < 32; rr += 2) {
          const int64_t mrow = mrow0 + rr + h;
          if (mrow < p.N && mcol < p.M) {
            gptr c = (gptr)p.C + mrow * p.ldc + mcol;
            *c = ts[(rr + h) * 33 + r];
          }
        }
      }
  }
#if defined(BX_STAMP) && BX_STAMP == 2
  if (tid == 0 && g_bx_stamp && p.syrk == 2 && blockIdx.x < g_bx_stamp_cap) g_bx_stamp[8 * blockIdx.x + 7] = __builtin_amdgcn_s_memrealtime();   // flush issued
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's flush is in memory
  __syncthreads();
  if (tid == 0 && g_bx_stamp && p.syrk == 2 && blockIdx.x < g_bx_stamp_cap) g_bx_stamp[8 * blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---------------------------------------------------------------------------------------------
// Streaming variant for outputs with at most 64 rows (the panel product P^T = V^T A22 of the band
// reduction: 64 x m x m, its big operand read exactly once from HBM, 32 flop/byte).  Such a product is
// HBM-bound and needs ~100 KB in flight per CU; the register-staged tile has 20-40.  Same global -> LDS
// DMA and swizzled tiles as gemm256_kernel, 64 x 256 tile (four waves of 64 x 64), SEVEN LDS stages of
// 20 KB: six K tiles (120 KB) are in flight while one is being multiplied.  Accumulators are 64 registers,
// nothing spills, so the waits are plain asm `s_waitcnt vmcnt(20)` (all but the four newest tiles landed).
constexpr int G64_NST = 7;
constexpr int G64_TA = 64 * BK, G64_TB = 256 * BK, G64_STG = G64_TA + G64_TB;  // floats
constexpr int GEMM64_LDS_BYTES = G64_NST * G64_STG * 4;                        // 140 KB

template <int ALAY, int BLAY>
__global__ __launch_bounds__(256, 1) void gemm64_dma_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem3[];
  const int tj = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t col0 = (int64_t)tj * 256;
  const int64_t kbeg = (int64_t)blockIdx.y * p.kchunk;
  const int64_t kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;
  const int nt = (int)((kend - kbeg) / BK);  // K and kchunk are multiples of 16 (host)

  f32x16 acc[2][2];
  auto clear_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };
  clear_acc();

  const bool partial = p.ksplit > 1;
  gptr Cout = (gptr)(partial ? p.slab + (int64_t)blockIdx.y * p.M * p.N : p.C);
  const int64_t ldc = partial ? p.N : p.ldc;
  const float alpha = partial ? 1.f : p.alpha;
  const float beta0 = partial ? 0.f : p.beta;
  auto flush_to_c = [&](bool first) __attribute__((always_inline)) {
    const float beta = first ? beta0 : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t col = col0 + wave * 64 + j * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (row < p.M && col < p.N) {
            gptr c = Cout + row * ldc + col;
            float v = alpha * acc[i][j][e];
            if (beta != 0.f) v += beta * *c;
            *c = v;
          }
        }
      }
  };

  float fa[2][2][4], fb[2][2][4];  // [half parity][tile][k pair]
  auto frags = [&](int st, int q, int par) __attribute__((always_inline)) {
#if defined(G64_VAR) && G64_VAR >= 2   // timing-only: no LDS reads either (the DMA stream and the barriers alone)
    return;
#endif
    const float *sa = smem3 + st * G64_STG, *sb = sa + G64_TA;
#pragma unroll
    for (int i = 0; i < 2; ++i) frag_half<ALAY, 64>(sa, i * 32 + r, q, h, fa[par][i]);
#pragma unroll
    for (int j = 0; j < 2; ++j) frag_half<BLAY, 256>(sb, wave * 64 + j * 32 + r, q, h, fb[par][j]);
  };
  auto mfma_half = [&](int par) __attribute__((always_inline)) {
#if defined(G64_VAR) && G64_VAR >= 1   // timing-only builds (scripts/probe/variants.sh): no products, fragments kept alive
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int i = 0; i < 2; ++i) __asm__ volatile("" ::"v"(fa[par][i][tt]), "v"(fb[par][i][tt]));
#else
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[par][i][tt], fb[par][j][tt], acc[i][j], 0, 0, 0);
#endif
  };

  // DMA sources: wave w moves block w of the A tile and blocks w, w+4, w+8, w+12 of the B tile
  gcptr srcA = dma_src<ALAY, 64>(p.A, p.lda, 0, p.M, wave, lane) + (ALAY == LAY_K ? kbeg : kbeg * p.lda);
  gcptr srcB[4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
    srcB[u] = dma_src<BLAY, 256>(p.B, p.ldb, col0, p.N, wave + 4 * u, lane) + (BLAY == LAY_K ? kbeg : kbeg * p.ldb);
  const int64_t stepA = (ALAY == LAY_K) ? BK : (int64_t)BK * p.lda, stepB = (BLAY == LAY_K) ? BK : (int64_t)BK * p.ldb;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem3 +
                                                       (unsigned)(wave * 256 * 4));
  auto issue = [&](int st) __attribute__((always_inline)) {
    const unsigned base = lds0 + (unsigned)(st * G64_STG * 4);
    dma16(srcA, base);
    srcA += stepA;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      dma16(srcB[u], base + (unsigned)((G64_TA + 4 * u * 256) * 4));
      srcB[u] += stepB;
    }
  };

  bool first_flush = true;
  constexpr int FL = 8192 / BK;
  for (int c0 = 0; c0 < nt; c0 += FL) {
    const int c1 = c0 + FL < nt ? c0 + FL : nt;
    if (c0 > 0) clear_acc();
    __syncthreads();
    // tiles c0 .. c0+NST-2 requested up front; tile t lives in stage (t - c0) % NST
    int issued = c0;
    for (; issued < c1 && issued < c0 + G64_NST - 1; ++issued) issue((issued - c0) % G64_NST);
    // first tile landed (everything but the younger requests)
    if (issued - c0 >= 2) {
      // wait for tile c0 only if enough tiles are in flight to express it with a constant; else wait for all
      if (issued - c0 == G64_NST - 1) __asm__ volatile("s_waitcnt vmcnt(25)" ::: "memory");  // (NST-2) * 5
      else __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    frags(0, 0, 0);
    for (int t = c0; t < c1; ++t) {
      const int st = (t - c0) % G64_NST, st1 = (t + 1 - c0) % G64_NST;
      // own part of tile t+1 landed: at most tiles t+2 .. t+NST-2 (NST-3 of them) may still be in flight
      if (issued - (t + 2) >= G64_NST - 3) __asm__ volatile("s_waitcnt vmcnt(20)" ::: "memory");
      else __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (issued < c1) {  // request tile t+NST-1 into the stage tile t-1 has left
        issue((issued - c0) % G64_NST);
        ++issued;
      }
      frags(st, 1, 1);
      mfma_half(0);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < c1) {
        __syncthreads();  // every wave's part of tile t+1 is in LDS; tile t-1's stage is free for the next request
        frags(st1, 0, 0);
      }
      mfma_half(1);
      __builtin_amdgcn_sched_barrier(0);
    }
    flush_to_c(first_flush);
    first_flush = false;
  }
}

// ---- The same 64-row streaming product on the bf16 pipe (round 4).
// The fp32 kernel above is bound by its own MFMAs (32 of 64 cycles per K tile and wave: 2.1 ms for 64 x 40960 x 40960
// where the DMA stream alone takes 1.1 ms, scripts/probe/panel_product.py with the G64_VAR builds).  Here every product is
// six v_mfma_f32_32x32x16_bf16 on exact three-way bf16 splits (24 MFMAs of 32 cycles per K tile and wave):
//   * the small operand A (64 x K, read by every workgroup) is split ONCE by g64_split_a_kernel into the fragment
//     order of the MFMA -- per K tile [row tile 2][piece 3][lane 64][8 bf16] = 6 KB -- and streamed by DMA like B;
//   * the big operand B (K x N or N x K, fp32, read exactly once from HBM) is split in registers by the wave that
//     multiplies it: 16 values per lane and K tile (~100 VALU instructions), issued between the MFMAs of the PREVIOUS
//     tile (pieces are double-buffered in registers; one wave per SIMD owns the SIMD's 512 registers).
// Splitting both operands in registers was measured too: ~200 VALU instructions per tile do not fit the MFMA gaps
// (1.88 ms).  Element j of a lane's eight values of a K tile is k = 8 (j >> 2) + 4 h + (j & 3) for both operands (the order
// frag_half delivers B in), which is all a 16-deep MFMA needs.  One DMA stream runs over the whole K range of the
// workgroup; accumulation chains are closed in registers every 2048 k (the MFMA's truncating accumulator: see
// bx_chain_tiles) and C or the split-K slab is written once.
constexpr int G64X_NST = 7;
constexpr int G64X_TA = 6 * 1024 / 4;                 // floats: the A pieces of one K tile (6 KB)
constexpr int G64X_STG = G64X_TA + G64_TB;            // 22 KB per stage
constexpr int GEMM64X_LDS_BYTES = G64X_NST * G64X_STG * 4;  // 154 KB
constexpr int G64X_CHAIN = 2048 / BK;

// this lane's K index inside a K tile for element j of its MFMA fragment
__device__ __forceinline__ int g64x_k(int h, int j) { return 8 * (j >> 2) + 4 * h + (j & 3); }

template <int ALAY>
__global__ __launch_bounds__(128) void g64_split_a_kernel(const float *__restrict__ A, int64_t lda, int64_t M, uint4 *__restrict__ out) {
  const int64_t kt = blockIdx.x;
  const int i = threadIdx.x >> 6, l = threadIdx.x & 63, h = l >> 5;
  const int64_t row = 32 * i + (l & 31);
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int64_t k = kt * BK + g64x_k(h, j);
    f[j] = row < M ? (ALAY == LAY_K ? A[row * lda + k] : A[k * lda + row]) : 0.f;
  }
  unsigned hh[4], mm[4], ll[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) bx_split2(f[2 * u], f[2 * u + 1], hh[u], mm[u], ll[u]);
  uint4 *o = out + (kt * 6 + i * 3) * 64 + l;
  o[0] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
  o[64] = make_uint4(mm[0], mm[1], mm[2], mm[3]);
  o[128] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}

template <int BLAY>
__global__ __launch_bounds__(256, 1) void gemm64_bx_kernel(GemmArgs p, const uint4 *__restrict__ apieces) {
  extern __shared__ __attribute__((aligned(16))) float smem3[];
  const int tj = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t col0 = (int64_t)tj * 256;
  const int64_t kbeg = (int64_t)blockIdx.y * p.kchunk;
  const int64_t kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;
  const int nt = (int)((kend - kbeg) / BK);  // K and kchunk are multiples of 16 (host)

  f32x16 acc[2][2], tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f, tot[i][j][e] = 0.f;

  // DMA sources: wave w moves bytes [1536 w, 1536 w + 1536) of the A pieces (one full and one half-wave request) and
  // blocks w, w+4, w+8, w+12 of the B tile
#if defined(G64_VADDR)
  gcptr srcA = (gcptr)(reinterpret_cast<const char *>(apieces) + (kbeg / BK) * 6144 + wave * 1536 + lane * 16);
#endif
  gcptr srcB[4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
    srcB[u] = dma_src<BLAY, 256>(p.B, p.ldb, col0, p.N, wave + 4 * u, lane) + (BLAY == LAY_K ? kbeg : kbeg * p.ldb);
  const int64_t stepB = (BLAY == LAY_K) ? BK : (int64_t)BK * p.ldb;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem3;
  const unsigned ldsA = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave * 1536)),
                 ldsB = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(G64X_TA * 4 + wave * 1024));
#if !defined(G64_VADDR)
  // Round 6: the requests take a wave-uniform base in scalar registers + ONE 32-bit offset per lane and stream instead of a
  // 64-bit pointer per lane (global_load_lds_dwordx4 v, s[..]): in the 256-tile kernel's K loop that is what a request's cost
  // to the SIMD hangs on (~25 cycles against ~3, gemm256_bx_kernel).  A lane's offset from the first row of the tile is
  // below 256 rows x ldb x 4 bytes; it does not change along K.  (-DG64_VADDR: the per-lane pointers of rounds 4-5.)  Same-box
  // A/B on N(0,1) data (scripts/probe/g64_ab.sh, profiles/r06_g64_ab.log): m = 40 960: 1820-1886 against 1884-1898 us;
  // m = 20 480: 470-484 against 509-515 us.
  gcptr baseA = (gcptr)(reinterpret_cast<const char *>(apieces) + (kbeg / BK) * 6144) + __builtin_amdgcn_readfirstlane(wave * (1536 / 4));
  gcptr baseB = (gcptr)p.B + (BLAY == LAY_K ? (col0 < p.N ? col0 : 0) * p.ldb + kbeg : kbeg * p.ldb + (col0 < p.N ? col0 : 0));
  const unsigned voffA = (unsigned)lane * 16u;
  unsigned voffB[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) voffB[u] = (unsigned)((const char *)srcB[u] - (const char *)baseB);
  auto dma16s = [](gcptr base, unsigned voff, unsigned lds_byte_addr) __attribute__((always_inline)) {
    __asm__ volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff), "s"(base) : "memory");
  };
  auto issue = [&](int st) __attribute__((always_inline)) {   // 6 requests per wave
    const unsigned so = (unsigned)(st * G64X_STG * 4);
    dma16s(baseA, voffA, ldsA + so);
    if (lane < 32) dma16s(baseA + 256, voffA, ldsA + so + 1024);   // (+256 floats = 1 KB; counted by vmcnt whatever the exec mask)
    baseA += 6144 / 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) dma16s(baseB, voffB[u], ldsB + so + (unsigned)(4 * u * 1024));
    baseB += stepB;
  };
#else
  auto issue = [&](int st) __attribute__((always_inline)) {   // 6 requests per wave
    const unsigned so = (unsigned)(st * G64X_STG * 4);
    dma16(srcA, ldsA + so);
    if (lane < 32) dma16(srcA + 256, ldsA + so + 1024);   // (+256 floats = 1 KB; counted by vmcnt whatever the exec mask)
    srcA += 6144 / 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      dma16(srcB[u], ldsB + so + (unsigned)(4 * u * 1024));
      srcB[u] += stepB;
    }
  };
#endif

  struct P3 { bf16x8 h, m, l; };
  P3 pa[2][2], pb[2][2];   // [tile parity][row / column tile]
  float raw[2][8];         // the next K tile's B fragments as read from LDS (columns 0-31, 32-63 of the wave's 64)
  auto read_next = [&](int st, int par) __attribute__((always_inline)) {
    const float *sa = smem3 + st * G64X_STG, *sb = sa + G64X_TA;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bf16x8 *q = reinterpret_cast<const bf16x8 *>(sa) + (i * 3) * 64 + lane;
      pa[par][i].h = q[0];
      pa[par][i].m = q[64];
      pa[par][i].l = q[128];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      frag_half<BLAY, 256>(sb, wave * 64 + j * 32 + r, 0, h, *reinterpret_cast<float(*)[4]>(&raw[j][0]));
      frag_half<BLAY, 256>(sb, wave * 64 + j * 32 + r, 1, h, *reinterpret_cast<float(*)[4]>(&raw[j][4]));
    }
  };
  auto split_raw = [&](int par) __attribute__((always_inline)) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned hh[4], mm[4], ll[4];
#if defined(G64_VAR) && G64_VAR == 4   // timing-only: the products without the split
#pragma unroll
      for (int u = 0; u < 4; ++u) hh[u] = __float_as_uint(raw[j][u]), mm[u] = __float_as_uint(raw[j][4 + u]), ll[u] = hh[u] ^ mm[u];
#else
      // (13 instructions per pair as hipcc compiles it; a 9-instruction form -- packed subtractions in inline asm, the
      // packed conversion hidden from the optimiser -- ran 7 % SLOWER on the same box: hazard s_nops around the asm)
#pragma unroll
      for (int u = 0; u < 4; ++u) bx_split2(raw[j][2 * u], raw[j][2 * u + 1], hh[u], mm[u], ll[u]);
#endif
      pb[par][j].h = __builtin_bit_cast(bf16x8, (u32x4){hh[0], hh[1], hh[2], hh[3]});
      pb[par][j].m = __builtin_bit_cast(bf16x8, (u32x4){mm[0], mm[1], mm[2], mm[3]});
      pb[par][j].l = __builtin_bit_cast(bf16x8, (u32x4){ll[0], ll[1], ll[2], ll[3]});
    }
  };
  auto mfma_bx = [&](int par) __attribute__((always_inline)) {
    // six partial products per output tile, smallest first; the four tiles' chains interleaved
#define G64_BX4(PA, PB)                                                                                                 \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] =             \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[par][i].PA, pb[par][j].PB, acc[i][j], 0, 0, 0);
#if defined(G64_VAR) && G64_VAR == 3   // timing-only: the split without the products
    _Pragma("unroll") for (int i = 0; i < 2; ++i) __asm__ volatile("" ::"v"(pa[par][i].h), "v"(pa[par][i].m), "v"(pa[par][i].l),
                                                                   "v"(pb[par][i].h), "v"(pb[par][i].m), "v"(pb[par][i].l));
#else
    G64_BX4(l, h) G64_BX4(h, l) G64_BX4(m, m) G64_BX4(m, h) G64_BX4(h, m) G64_BX4(h, h)
#endif
#undef G64_BX4
  };

  int issued = 0, ist = 0;   // tiles requested so far; the stage the next request goes to
  for (; issued < nt && issued < G64X_NST - 1; ++issued, ++ist) issue(ist);
  if (issued == G64X_NST - 1) __asm__ volatile("s_waitcnt vmcnt(30)" ::: "memory");  // (NST-2) * 6: tile 0 landed
  else __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  read_next(0, 0);
  split_raw(0);
  int t = 0, st1 = 1;
  // tile t: its pieces are in registers (parity par); tile t+1 is read from LDS and split while tile t's 24 MFMAs run
  auto step = [&](int par) __attribute__((always_inline)) {
    if (issued < nt) {
      // own part of tile t+1 landed: tiles t+2 .. t+NST-2 (NST-3 of them) may still be in flight
      __asm__ volatile("s_waitcnt vmcnt(24)" ::: "memory");
#if !(defined(G64_VAR) && G64_VAR == 5)   // timing-only 5: no requests after the first NST-1 tiles
      issue(ist);  // tile t+NST-1 into the stage tile t-1 has left (its LDS reads ended before the previous barrier)
#endif
      ++issued;
      ist = ist + 1 == G64X_NST ? 0 : ist + 1;
    } else {
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();           // every wave's part of tile t+1 is in LDS
    read_next(st1, par ^ 1);   // (after the last tile: a stale stage, the values are not used)
    __builtin_amdgcn_sched_barrier(0);
    mfma_bx(par);
    split_raw(par ^ 1);
    // (an empty use, so that the split stays in this block: the compiler sinks it into the next step's otherwise)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __asm__ volatile("" : "+v"(pb[par ^ 1][j].h), "+v"(pb[par ^ 1][j].m), "+v"(pb[par ^ 1][j].l));
    // the split's ~100 VALU instructions between the MFMAs: 4 MFMAs first (the LDS reads are on their way), then 5 : 1
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
    for (int u = 0; u < 20; ++u) {
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    ++t;
    st1 = st1 + 1 == G64X_NST ? 0 : st1 + 1;
  };
  static_assert(G64X_CHAIN % 2 == 0, "chains hold whole pairs of tiles (the register parity of the pieces)");
  while (t < nt) {
    const int len = nt - t < G64X_CHAIN ? nt - t : G64X_CHAIN;   // this chain; only the last one can be odd
    for (int c = 0; c < len / 2; ++c) {
      step(0);
      step(1);
    }
    if (len & 1) step(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)   // close the chain in registers
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          tot[i][j][e] += acc[i][j][e];
          acc[i][j][e] = 0.f;
        }
  }

  const bool partial = p.ksplit > 1;
  gptr Cout = (gptr)(partial ? p.slab + (int64_t)blockIdx.y * p.M * p.N : p.C);
  const int64_t ldc = partial ? p.N : p.ldc;
  const float alpha = partial ? 1.f : p.alpha, beta = partial ? 0.f : p.beta;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = col0 + wave * 64 + j * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < p.M && col < p.N) {
          gptr c = Cout + row * ldc + col;
          float v = alpha * tot[i][j][e];
          if (beta != 0.f) v += beta * *c;
          *c = v;
        }
      }
    }
}

// C = alpha * sum_z slab[z] + beta * C  (fixed summation order); SYRK slabs hold the lower
// tiles only, the upper triangle is read from the transposed position.
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float *__restrict__ slab, float *__restrict__ C,
                                                          int64_t M, int64_t N, int64_t ldc, int ksplit,
                                                          float alpha, float beta, int syrk, int tile = BM) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * N) return;
  const int64_t i = idx / N, j = idx - i * N;
  int64_t src = idx;
  if (syrk && (j / tile) > (i / tile)) src = j * N + i;
  float s = 0.f;
  for (int z = 0; z < ksplit; ++z) s += slab[(int64_t)z * M * N + src];
  float v = alpha * s;
  if (beta != 0.f) v += beta * C[i * ldc + j];
  C[i * ldc + j] = v;
}

// ---------------------------------------------------------------------------------------------
// Deep-K products with a small output, both operands K-contiguous (the band reduction's 64 x 64 ... 64 x 384 Gram
// blocks over m = 4e4 rows: S = V^T V, V^T [V1 W1 ...], X^T V).  The general tile kernels move one 16-k tile (64 bytes
// per operand row) per pipeline step and are latency-bound here (0.2-0.5 TB/s; 105-140 us per product at m = 40 960).
// This kernel gives every workgroup a 64 x 64 output block and a K range, and streams 128 k at a time: each operand
// row contributes 512 contiguous bytes per step, all 16 float4 loads of a thread are in flight while the previous
// step's 64 MFMAs run from LDS (row stride 132 floats: conflict-free ds_read_b128 rows).  MFMA step pairs
// k = 8 j + i (lanes 0-31) with k = 8 j + 4 + i (lanes 32-63) for both operands, so fragments are plain float4 reads.
// Split-K partials go to a slab [split][M][N] and are summed in a fixed order by gemm_tsk_reduce_kernel.
constexpr int TSK_KC = 128, TSK_LD = TSK_KC + 4, TSK_NL = 64 * (TSK_KC / 4) / 256;  // 8 float4 per thread and operand
struct TskArgs {
  const float *A, *B;
  float *C, *slab;
  int64_t lda, ldb, ldc, K, kchunk;
  int M, N, nsplit;
  float alpha, beta;
};

__global__ __launch_bounds__(256, 2) void gemm_tsk_kernel(TskArgs p) {
  __shared__ __attribute__((aligned(16))) float sA[64 * TSK_LD];
  __shared__ __attribute__((aligned(16))) float sB[64 * TSK_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int jb = blockIdx.y, ks = blockIdx.x;
  const int64_t kbeg = (int64_t)ks * p.kchunk;
  const int64_t kend = kbeg + p.kchunk < p.K ? kbeg + p.kchunk : p.K;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float4 ra[TSK_NL], rb[TSK_NL];
  // loads are unconditional (clamped addresses) and masked only when they are written to LDS: a select right behind
  // each load makes hipcc wait for it before issuing the next one (16 serialised round trips: 10 us per step)
  auto gload = [&](int64_t k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TSK_NL; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / (TSK_KC / 4), c = (idx - row * (TSK_KC / 4)) * 4;
      const int64_t k = k0 + c;
      const int64_t kc = k < kend ? k : kbeg;  // K % 4 == 0: all four lanes of the float4 in or out
      const int brow = 64 * jb + row;
      ra[i] = *reinterpret_cast<const float4 *>(p.A + (int64_t)(row < p.M ? row : 0) * p.lda + kc);
      rb[i] = *reinterpret_cast<const float4 *>(p.B + (int64_t)(brow < p.N ? brow : 0) * p.ldb + kc);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  gload(kbeg);
  for (int64_t k0 = kbeg; k0 < kend; k0 += TSK_KC) {
    __syncthreads();  // every wave is done with the previous step's fragments
#pragma unroll
    for (int i = 0; i < TSK_NL; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / (TSK_KC / 4), c = (idx - row * (TSK_KC / 4)) * 4;
      const bool kok = k0 + c < kend;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4 *>(sA + row * TSK_LD + c) = (kok && row < p.M) ? ra[i] : z;
      *reinterpret_cast<float4 *>(sB + row * TSK_LD + c) = (kok && 64 * jb + row < p.N) ? rb[i] : z;
    }
    __syncthreads();
    if (k0 + TSK_KC < kend) gload(k0 + TSK_KC);  // in flight during the MFMAs
    const float *pa = sA + (32 * wm + r) * TSK_LD + 4 * h, *pb = sB + (32 * wn + r) * TSK_LD + 4 * h;
#pragma unroll
    for (int j = 0; j < TSK_KC / 8; ++j) {
      const float4 a4 = *reinterpret_cast<const float4 *>(pa + 8 * j), b4 = *reinterpret_cast<const float4 *>(pb + 8 * j);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
    }
  }
  const bool direct = p.nsplit == 1;
  float *out = direct ? p.C : p.slab + (int64_t)ks * p.M * p.N;
  const int64_t ldo = direct ? p.ldc : p.N;
  const int j = 64 * jb + 32 * wn + r;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = 32 * wm + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (i < p.M && j < p.N) {
      float v = acc[e];
      if (direct) {
        v *= p.alpha;
        if (p.beta != 0.f) v += p.beta * out[(int64_t)i * ldo + j];
      }
      out[(int64_t)i * ldo + j] = v;
    }
  }
}

// C = alpha * sum_z slab[z] + beta * C: 32 consecutive output elements x 8 interleaved split subsets per workgroup
// (coalesced 128-byte reads, 8 x fewer dependent loads per thread than one thread per element), fixed summation order.
__global__ __launch_bounds__(256) void gemm_tsk_reduce_kernel(const float *__restrict__ slab, float *__restrict__ C, int64_t MN,
                                                              int N, int64_t ldc, int nsplit, float alpha, float beta) {
  __shared__ float part[8][32];
  const int tid = threadIdx.x, e = tid & 31, sub = tid >> 5;
  const int64_t idx = (int64_t)blockIdx.x * 32 + e;
  float s0 = 0.f, s1 = 0.f;
  if (idx < MN) {
    int z = sub;
    for (; z + 8 < nsplit; z += 16) {
      s0 += slab[(int64_t)z * MN + idx];
      s1 += slab[(int64_t)(z + 8) * MN + idx];
    }
    if (z < nsplit) s0 += slab[(int64_t)z * MN + idx];
  }
  part[sub][e] = s0 + s1;
  __syncthreads();
  if (sub == 0 && idx < MN) {
    const float s = ((part[0][e] + part[1][e]) + (part[2][e] + part[3][e])) + ((part[4][e] + part[5][e]) + (part[6][e] + part[7][e]));
    const int64_t i = idx / N, j = idx - i * N;
    float v = alpha * s;
    if (beta != 0.f) v += beta * C[i * ldc + j];
    C[i * ldc + j] = v;
  }
}

static bool tsk_shape(int64_t M, int64_t N, int64_t K) { return M <= 64 && N <= 1024 && K >= 2048 && (K & 3) == 0; }

static void tsk_plan(int64_t N, int64_t K, int &nsplit, int64_t &kchunk) {
  const int64_t nblk = cdiv(N, 64);
  static int total = -1;
  if (total < 0) { const char *e = getenv("VIVIT_TSK_WGS"); total = e ? atoi(e) : 512; }
  int64_t want = total / nblk, maxs = K / (2 * TSK_KC);   // two workgroups per CU; at least two steps per split
  if (want < 1) want = 1;
  int64_t s = want < maxs ? want : maxs;
  if (s < 1) s = 1;
  kchunk = cdiv(cdiv(K, s), TSK_KC) * TSK_KC;
  nsplit = (int)cdiv(K, kchunk);
}

static size_t tsk_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (!tsk_shape(M, N, K)) return 0;
  int ns;
  int64_t kc;
  tsk_plan(N, K, ns, kc);
  return ns > 1 ? (size_t)ns * (size_t)M * (size_t)N * sizeof(float) : 0;
}

static bool use_tsk(int alay, int blay, const float *A, const float *B, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                    bool syrk) {
  static int forced = -2;
  if (forced == -2) {
    const char *e = getenv("VIVIT_GEMM_TSK");
    forced = e ? atoi(e) : -1;
  }
  if (forced == 0 || syrk || alay != LAY_K || blay != LAY_K || !tsk_shape(M, N, K)) return false;
  return (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0 && (lda & 3) == 0 && (ldb & 3) == 0;
}

static int tsk_launch(const float *A, const float *B, float *C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                      float alpha, float beta, void *workspace, size_t workspace_bytes, hipStream_t stream) {
  TskArgs p;
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.K = K; p.M = (int)M; p.N = (int)N;
  p.alpha = alpha; p.beta = beta;
  tsk_plan(N, K, p.nsplit, p.kchunk);
  p.slab = nullptr;
  if (p.nsplit > 1) {
    if (!workspace || workspace_bytes < (size_t)p.nsplit * (size_t)M * (size_t)N * sizeof(float)) return VIVIT_E_WORKSPACE;
    p.slab = static_cast<float *>(workspace);
  }
  gemm_tsk_kernel<<<dim3((unsigned)p.nsplit, (unsigned)cdiv(N, 64)), 256, 0, stream>>>(p);
  int st = launch_status();
  if (st != VIVIT_OK || p.nsplit == 1) return st;
  gemm_tsk_reduce_kernel<<<(unsigned)cdiv(M * N, 32), 256, 0, stream>>>(p.slab, C, M * N, (int)N, ldc, p.nsplit, alpha, beta);
  return launch_status();
}

// wave grid of the tile: 1 x 4 waves (64 x 256) when the output has at most 64 rows and is wide
static int pick_wm(int64_t M, int64_t N, bool syrk) { return (!syrk && M <= 64 && N > 128) ? 1 : 2; }

static void choose_split(int64_t M, int64_t N, int64_t K, bool syrk, int &ksplit, int64_t &kchunk) {
  const int wm = pick_wm(M, N, syrk);
  const int64_t tm = cdiv(M, 64 * wm), tn = cdiv(N, 256 / wm);
  const int64_t tiles = syrk ? tm * (tm + 1) / 2 : tm * tn;
  ksplit = 1;
  kchunk = cdiv(K, BK) * BK;
  if (kchunk < BK) kchunk = BK;
  const int64_t ktiles = cdiv(K, BK);
  // Fill at least ~2 workgroups per CU when the output has few tiles and K is deep; keep every
  // split at least 32 K tiles long and the slab modest.
  if (tiles < 256 && ktiles >= 64) {
    // one resident round (2 workgroups per CU x 256 CUs) for compute-bound shapes; a one-tile-wide
    // output streams its big operand once and is bandwidth-bound: more, shorter splits keep enough
    // bytes in flight
    static int skinny_want = -1;
    if (skinny_want < 0) { const char *e = getenv("VIVIT_SKINNY_WANT"); skinny_want = e ? atoi(e) : 2048; }
    int64_t want = (tm == 1 || tn == 1) ? skinny_want / tiles : 512 / tiles;
    int64_t maxs = (tm == 1 || tn == 1) ? ktiles / 8 : ktiles / 32;
    int64_t s = want < maxs ? want : maxs;
    // a single-tile output (the 64 x 64 Gram blocks of the band reduction's panels: both operands stream 64 rows
    // x m) has nothing but split-K to spread over the chip (with the slot rotation of map_tile: before it every
    // split's only valid workgroup sat on XCD 0 and more splits bought nothing)
    static int cap1 = -1;
    if (cap1 < 0) { const char *e = getenv("VIVIT_SPLIT_CAP1"); cap1 = e ? atoi(e) : 64; }
    const int64_t cap = tiles <= 2 ? cap1 : 64;
    if (s > cap) s = cap;
    while (s > 1 && (size_t)s * (size_t)M * (size_t)N * 4 > ((size_t)1 << 30)) --s;
    if (s > 1) {
      kchunk = cdiv(ktiles, s) * BK;
      ksplit = (int)cdiv(K, kchunk);
    }
  }
}

static size_t gemm64_workspace_bytes(int64_t M, int64_t N, int64_t K, int *ksplit_out, int64_t *kchunk_out, size_t *slab_out = nullptr);
static bool gemm256_plan(int64_t M, int64_t N, int64_t K, bool syrk, int *ksplit_out, int64_t *kchunk_out, int max_split = 32);

static int gemm_split_mode();
static size_t bx_workspace_bytes(int64_t M, int64_t N, int64_t K, bool same);
static bool bx_splitk_shape(int64_t M, int64_t N, int64_t K, bool syrk, bool same, int *nsplit_out, int *kt_split_out,
                            size_t *bytes_out);
size_t gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, bool syrk) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int ksplit;
  int64_t kchunk;
  choose_split(M, N, K, syrk, ksplit, kchunk);
  size_t b = ksplit > 1 ? (size_t)ksplit * (size_t)M * (size_t)N * sizeof(float) : 0;
  {
    int s256;
    int64_t kc256;
    if (gemm256_plan(M, N, K / BK * BK, syrk, &s256, &kc256)) {
      if (s256 > 1) {
        const size_t b256 = (size_t)s256 * (size_t)M * (size_t)N * sizeof(float);
        if (b256 > b) b = b256;
      }
      if (gemm_split_mode() != 0) {  // operand pieces of the bf16-pipe path (gemm256_launch prefers it to split-K)
        const size_t bb = bx_workspace_bytes(M, N, K / BK * BK, syrk);
        if (bb > b) b = bb;
      }
    }
  }
  {  // bf16-pipe split-K for small outputs with a deep contraction (bx_splitk_launch)
    size_t bs = 0;
    if (bx_splitk_shape(M, N, K / BK * BK, syrk, syrk, nullptr, nullptr, &bs) && bs > b) b = bs;
    if (!syrk && bx_splitk_shape(M, N, K / BK * BK, false, false, nullptr, nullptr, &bs) && bs > b) b = bs;
  }
  if (!syrk) {  // the deep-K small-output kernel may be chosen instead (tsk_launch)
    const size_t bt = tsk_workspace_bytes(M, N, K);
    if (bt > b) b = bt;
  }
  if (!syrk && M <= 64 && N >= 2048 && K >= 2048) {  // the streaming kernel may be chosen instead (gemm64_launch)
    const size_t b64 = gemm64_workspace_bytes(M, N, K / BK * BK, nullptr, nullptr);
    if (b64 > b) b = b64;
  }
  return b;
}

__global__ __launch_bounds__(256) void scale_c_kernel(float *__restrict__ C, int64_t M, int64_t N, int64_t ldc, float beta) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * N) return;
  const int64_t i = idx / N, j = idx - i * N;
  C[i * ldc + j] = beta == 0.f ? 0.f : beta * C[i * ldc + j];
}

// The 256 x 256 tile pays off once the output has enough of them to fill the chip (one per CU).
// Split-K (slabs + fixed-order reduce) is implemented for small outputs with a deep contraction, but stays
// opt-in (VIVIT_GEMM256_SPLIT=1): measured on the Gram matrices of small batches (n = 1280, P = 4e5) the
// streamed K-major operand then reaches only 0.2 TB/s - 17 splits x 1280 row streams 1.6 MB apart - and the
// 128 x 128 tile with its 2 workgroups per CU is twice as fast (27 ms vs 59 ms); to be revisited with the
// seven-stage pipeline of gemm64_dma_kernel.
static bool gemm256_plan(int64_t M, int64_t N, int64_t K, bool syrk, int *ksplit_out, int64_t *kchunk_out, int max_split) {
  static int forced = -2, split = 0;
  if (forced == -2) {
    const char *e = getenv("VIVIT_GEMM256");
    forced = e ? atoi(e) : -1;
    const char *e2 = getenv("VIVIT_GEMM256_SPLIT");
    split = e2 ? atoi(e2) : 0;
  }
  if (ksplit_out) *ksplit_out = 1;
  if (kchunk_out) *kchunk_out = cdiv(K, BK) * BK;
  static int kmin = -1;
  if (kmin < 0) { const char *e = getenv("VIVIT_GEMM256_KMIN"); kmin = e ? atoi(e) : 512; }
  if (forced == 0 || K < kmin) return false;
  const int64_t tm = cdiv(M, B2), tn = cdiv(N, B2);
  const int64_t tiles = syrk ? tm * (tm + 1) / 2 : tm * tn;
  // one workgroup per CU: prologue (first DMA round trip) and epilogue (256 KB of C) are not overlapped with
  // another workgroup's main loop, so the contraction must be long enough to amortise them
  if (!split && tiles < 200) return false;
  if (!split) max_split = max_split < 4 ? max_split : 4;  // large outputs: only to fill the last round of workgroups
  if (M < 512 || N < 512) return false;
  const int64_t ktiles = K / BK;
  const double flops = (syrk ? 1.0 : 2.0) * (double)M * (double)N * (double)K;
  const double t_mfma = flops / 140e12;
  double best = 0.0;
  int best_s = 0;
  for (int s = 1; s <= max_split; ++s) {
    if (s > 1 && ktiles / s < 128) break;  // every split keeps >= 2048 k
    if (s > 1 && (size_t)s * (size_t)M * (size_t)N * 4 > ((size_t)1 << 30)) break;  // slab cap (no slab for s = 1)
    const int64_t wgs = tiles * s;
    const double fill = (double)wgs / (double)(256 * cdiv(wgs, 256));
    const double t_slab = s > 1 ? 2.0 * s * (double)M * (double)N * 4.0 / 3e12 : 0.0;
    const double eff = fill / (1.0 + t_slab / t_mfma);
    if (eff > best + 1e-9) { best = eff; best_s = s; }
  }
  if (best_s == 0 || best < 0.6) return false;
  if (ksplit_out && best_s > 1) {
    const int64_t kchunk = cdiv(ktiles, best_s) * BK;
    *kchunk_out = kchunk;
    *ksplit_out = (int)cdiv(K, kchunk);
  }
  return true;
}

// Which matrix pipe the 256-tile NT products (Gram SYRK, K-contiguous GEMMs) use: 6 (default) = bf16 pipe with exact
// three-way operand splits and the 6 partial products that are >= 2^-16 of a product; 9 = all nine; 0 = fp32 MFMA
// (gemm256_kernel); 3 = hi hi + hi mid + mid hi of the three-way split (per-product error 2^-16: experiments only).
// VIVIT_GEMM_SPLIT overrides.
static int gemm_split_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char *e = getenv("VIVIT_GEMM_SPLIT");
    mode = e ? atoi(e) : 6;
    if (mode != 0 && mode != 3 && mode != 6 && mode != 9) mode = 6;
  }
  return mode;
}

// Length of one MFMA accumulation chain of the bf16-pipe kernel, in K tiles.  v_mfma_f32_32x32x16_bf16 is NOT a chain
// of correctly rounded fmas: it adds the accumulator and two 8-product group sums after aligning them to the largest
// exponent with about one guard bit, and what is shifted out is TRUNCATED (scripts/probe/mfma_round.hip: c = 1 plus
// sixteen products of 3/64 ulp returns 1; one product of 0.51 ulp rounds up correctly).  Once the accumulator is more
// than ~2^6 times a group sum (chains beyond ~512 k) the low bits of every group sum are cut off towards zero: noise of
// twice the fp32 chain's rounding for sums of random signs, and a BIAS for sums of equal signs -- the diagonal of a Gram
// matrix, entries of correlated rows.  Measured on n = 5120, K = 401 408 (independent N(0,1) rows), chain length ->
// off-diagonal error per term of the random walk / mean relative error of the diagonal:
//     8192 -> 3.2e-6 / -2.5e-6     4096 -> 2.3e-6 / -2.0e-6     2048 -> 1.7e-6 / -1.3e-6     1024 -> 1.2e-6 / -5.5e-7
//      512 -> 9.8e-7 / -6e-8       fp32 MFMA kernel (chains of 8192, correctly rounded): 1.6e-6 / -3.2e-7
// and what a shorter chain costs at the headline shape (n = 40 960: C is 6.7 GB, every flush is HBM traffic that
// competes with the operand panels for the L2 / Infinity Cache): 8192 -> 4096: + 2 %, -> 2048: + 6 %, -> 1024: + 10 %;
// diagonal tiles at 512: + 1-2 % per launch, + 5 % at the headline shape (140 ms).  Default: 4096 k (1.4 x the fp32 MFMA
// kernel's random-sign error) on ALL tiles -- the same-sign sums of a Gram matrix are its diagonal ENTRIES, which the
// public SYRK computes separately in fp64 (syrk_diag_kernel: 16 ms instead of 140) --, 1024 k in the split-K form for
// small outputs (whose yardstick is the 128-tile fp32 kernel with its chains of 2048); the eigensolver's internal
// products on orthogonal factors (random signs, the measured off-diagonal case) keep 8192.
// VIVIT_BX_FLUSH / VIVIT_BX_FLUSH_DIAG / VIVIT_BX_FLUSH_SPLITK / VIVIT_BX_FLUSH_INTERNAL override (in k).
static int bx_env_tiles(const char *name, int dflt_k) {
  const char *e = getenv(name);
  int ft = (e ? atoi(e) : dflt_k) / BK;
  return ft < 1 ? 1 : ft;
}
static int bx_flush_tiles() { static int ft = -1; if (ft < 0) ft = bx_env_tiles("VIVIT_BX_FLUSH", 4096); return ft; }
static int bx_flush_diag() { static int ft = -1; if (ft < 0) ft = bx_env_tiles("VIVIT_BX_FLUSH_DIAG", 4096); return ft; }
static int bx_flush_internal() { static int ft = -1; if (ft < 0) ft = bx_env_tiles("VIVIT_BX_FLUSH_INTERNAL", 8192); return ft; }
static int bx_flush_splitk() { static int ft = -1; if (ft < 0) ft = bx_env_tiles("VIVIT_BX_FLUSH_SPLITK", 1024); return ft; }

// columns of an operand split at a time (workspace: 6 bytes per element of the chunk and operand)
// Public products (the Gram SYRKs of the caller) take ONE accumulation chain per launch (round 4): 4096 columns instead of
// 65 536 made the headline Gram build 3-4 % faster on four boxes (2.78-2.80 -> 2.67-2.71 s; 32 768: - 1.5 %, 16 384: - 2.5 %,
// 6144 = a chain and a half: worse than either neighbour, 2048: + 5 %) -- no flush in the middle of a launch, and the
// workgroups of an XCD start every chain together again (they drift apart by whole tiles otherwise, which is what the
// re-fetch traffic of section 4.1b pays for).  The eigensolver's internal products keep 65 536 (their step was 12 ms
// slower with the short chunks).  VIVIT_GEMM_SPLIT_KC sets both.
static int64_t bx_chunk_cols(int64_t K, bool pub) {
  static int64_t kc_env = -2;
  if (kc_env == -2) {
    const char *e = getenv("VIVIT_GEMM_SPLIT_KC");
    kc_env = e ? atoll(e) / 16 * 16 : -1;
    if (e && kc_env < 16) kc_env = 16;
  }
  const int64_t kc = kc_env > 0 ? kc_env : (pub ? (int64_t)bx_flush_tiles() * BK : 65536);
  return K < kc ? K : kc;
}
static bool bx_public_product();
static size_t bx_piece_bytes(int64_t M, int64_t N, int64_t K, bool same) {
  // (the public workspace queries run inside a BxStrictScope like the public launches: 1 GB of pieces instead of 16 GB at
  // n = 40 960; the eigensolver's own queries and launches see the long chunk)
  const int64_t kc = bx_chunk_cols(K, bx_public_product());
  const int64_t ra = cdiv(M, 32) * 32, rb = cdiv(N, 32) * 32;
  return (size_t)6 * (size_t)kc * (size_t)(same ? ra : ra + rb);
}
// pieces of one chunk + one range flag per chunk (BX_GATE)
// XCD-group start barrier of the bf16-pipe kernel (experiment, VIVIT_BX_SYNC=1): counters per launch
static bool bx_sync_enabled() {
  static int on = -1;
  if (on < 0) { const char *e = getenv("VIVIT_BX_SYNC"); on = e ? atoi(e) : 0; }
  return on != 0;
}
static size_t bx_sync_ints(int64_t M, int64_t N) {   // upper bound of 8 x (number of super-blocks)
  const int64_t tm = cdiv(M, B2), tn = cdiv(N, B2);
  return (size_t)8 * (size_t)(tm * tn / 256 + tm + tn + 2);
}
static size_t bx_workspace_bytes(int64_t M, int64_t N, int64_t K, bool same) {
  const size_t nch = (size_t)cdiv(K, bx_chunk_cols(K, true));   // (the larger of the two counts)
  return bx_piece_bytes(M, N, K, same) + 256 + 4 * nch + 256 + (bx_sync_enabled() ? 4 * nch * bx_sync_ints(M, N) + 256 : 0);
}

// Which range flags send a chunk to the fp32 MFMA kernel.  The public products (vivit_gram_syrk_f32, vivit_gemm_*_f32)
// honour both bits and so keep fp32-MFMA semantics for every input; the eigensolver's internal products on orthogonal
// factors only reroute non-finite / out-of-range chunks (a localised eigenvector has entries below 2^-100 whose
// 2^-126-level piece is immaterial, and the reroute would cost that chunk the bf16 pipe's 2.7x).
static thread_local int tls_bx_gate_mask = BX_GATE_RANGE;
// true inside a public product (vivit_gram_syrk_f32 / vivit_gemm_*_f32): the profile (roofline.achieved of bench.py) counts
// the Gram SYRKs of the caller, not the reflector Gram matrices the eigensolver's back-transformation builds internally
static bool bx_public_product() { return (tls_bx_gate_mask & BX_GATE_TINY) != 0; }

struct BxStrictScope {
  int saved;
  BxStrictScope() : saved(tls_bx_gate_mask) { tls_bx_gate_mask = BX_GATE_RANGE | BX_GATE_TINY; }
  ~BxStrictScope() { tls_bx_gate_mask = saved; }
};

#ifndef BX_ASM_DEFAULT
#define BX_ASM_DEFAULT 1
#endif
// VIVIT_BX_ASM=1 / 0: the hand-scheduled K loop (gemm256_bx_kernel<6, true>) or the C++ loop (the reference implementation)
static bool bx_asm_enabled() {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("VIVIT_BX_ASM");
    on = e ? (atoi(e) != 0) : BX_ASM_DEFAULT;
  }
  return on != 0;
}
static void bx_launch6(dim3 grid, const GemmBxArgs &q, hipStream_t stream) {
  if (bx_asm_enabled())
    gemm256_bx_kernel<6, true><<<grid, 256, GEMM256BX_LDS_BYTES, stream>>>(q);
  else
    gemm256_bx_kernel<6><<<grid, 256, GEMM256BX_LDS_BYTES, stream>>>(q);
}

static bool gemm256_attrs() {
  static unsigned long long attr_done = 0;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (!(attr_done & (1ull << (dev & 63)))) {
      const void *fns[4] = {reinterpret_cast<const void *>(gemm256_kernel<LAY_K, LAY_K>),
                            reinterpret_cast<const void *>(gemm256_kernel<LAY_K, LAY_M>),
                            reinterpret_cast<const void *>(gemm256_kernel<LAY_M, LAY_K>),
                            reinterpret_cast<const void *>(gemm256_kernel<LAY_M, LAY_M>)};
      for (const void *f : fns)
        if (!ensure_dynamic_lds(f, GEMM256_LDS_BYTES, attr_done)) return false;
      const void *bx[4] = {reinterpret_cast<const void *>(gemm256_bx_kernel<3>), reinterpret_cast<const void *>(gemm256_bx_kernel<6>),
                           reinterpret_cast<const void *>(gemm256_bx_kernel<9>), reinterpret_cast<const void *>(gemm256_bx_kernel<6, true>)};
      for (const void *f : bx)
        if (!ensure_dynamic_lds(f, GEMM256BX_LDS_BYTES, attr_done)) return false;
      attr_done |= 1ull << (dev & 63);
    }
  }
  return true;
}

static int gemm256_launch(int alay, int blay, GemmArgs p, bool syrk, void *workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!gemm256_attrs()) return VIVIT_E_LAUNCH;
  // never more splits than the caller's workspace holds (callers size it for their largest problem; the plan
  // of a smaller one may differ)
  const size_t slab1 = (size_t)p.M * (size_t)p.N * sizeof(float);
  const int max_split = workspace ? (int)(workspace_bytes / slab1 < 32 ? workspace_bytes / slab1 : 32) : 1;
  gemm256_plan(p.M, p.N, p.K, syrk, &p.ksplit, &p.kchunk, max_split < 1 ? 1 : max_split);
  // the bf16-pipe kernel has no split-K: a last round of workgroups that is not full costs less than its 1.6x speed
  const bool bx_same = p.A == p.B && p.lda == p.ldb && p.M == p.N && alay == blay;
  if (gemm_split_mode() != 0 && workspace && workspace_bytes >= bx_workspace_bytes(p.M, p.N, p.K, bx_same)) {
    p.ksplit = 1;
    p.kchunk = cdiv(p.K, BK) * BK;
  }
  p.slab = p.ksplit > 1 ? static_cast<float *>(workspace) : nullptr;
  if (getenv("VIVIT_GEMM_DEBUG"))
    fprintf(stderr, "gemm256: M=%lld N=%lld K=%lld syrk=%d ksplit=%d kchunk=%lld max_split=%d lay=%d%d bxws=%d\n", (long long)p.M,
            (long long)p.N, (long long)p.K, (int)syrk, p.ksplit, (long long)p.kchunk, max_split, alay, blay,
            (int)(workspace && workspace_bytes >= bx_workspace_bytes(p.M, p.N, p.K, bx_same)));
  p.tiles_m = (int)cdiv(p.M, B2);
  p.tiles_n = (int)cdiv(p.N, B2);
  p.syrk = syrk ? 1 : 0;
  p.a_vec = ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (p.lda & 3) == 0) ? 1 : 0;
  p.b_vec = ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0 && (p.ldb & 3) == 0) ? 1 : 0;
  p.desc = nullptr;
  const int sbw = syrk ? 16 : sb_width(p.tiles_m, p.tiles_n), sbh = 256 / sbw;
  p.sbw = sbw;
  const int64_t sbm = cdiv(p.tiles_m, sbh), sbn = cdiv(p.tiles_n, sbw);
  const int64_t nsb = syrk ? sbm * (sbm + 1) / 2 : sbm * sbn;
  if (nsb * 256 > 0x7fffffffLL) return VIVIT_E_UNSUPPORTED;
  dim3 grid((unsigned)(nsb * 256), (unsigned)p.ksplit, 1);
  const bool prof = syrk && p.A == p.B && prof_enabled() && bx_public_product();
  if (prof) prof_begin(0, (double)p.M * (double)(p.M + 1) * (double)p.K, stream);
  const int bx = gemm_split_mode();
  if (bx != 0 && p.ksplit == 1 && workspace &&
      workspace_bytes >= bx_workspace_bytes(p.M, p.N, p.K, p.A == p.B && p.lda == p.ldb && p.M == p.N && alay == blay)) {
    // fp32 product on the bf16 pipe: K in chunks of BX_KC columns, per chunk the operand pieces (bx_split_kernel)
    // and one pure-bf16 launch that accumulates into C (beta = 1 from the second chunk on)
    const bool same = p.A == p.B && p.lda == p.ldb && p.M == p.N && alay == blay;
    const int64_t kc_max = bx_chunk_cols(p.K, bx_public_product());
    const int64_t nrbA = cdiv(p.M, 32), nrbB = cdiv(p.N, 32);
    unsigned short *PA = static_cast<unsigned short *>(workspace);
    const int64_t strideA = nrbA * 32 * kc_max, strideB = nrbB * 32 * kc_max;
    unsigned short *PB = same ? PA : PA + 3 * strideA;
    int *flags = reinterpret_cast<int *>(align_up(reinterpret_cast<uintptr_t>(workspace) + bx_piece_bytes(p.M, p.N, p.K, same), 256));
    const int64_t nchunks = cdiv(p.K, kc_max);
    int *sync = nullptr;
    const size_t sync_ints = bx_sync_ints(p.M, p.N);
    if (bx_sync_enabled()) sync = reinterpret_cast<int *>(align_up(reinterpret_cast<uintptr_t>(flags + nchunks), 256));
    if (hipMemsetAsync(flags, 0, 4 * (size_t)nchunks + (sync ? 256 + 4 * (size_t)nchunks * sync_ints : 0), stream) != hipSuccess)
      return VIVIT_E_LAUNCH;
    GemmBxArgs q;
    q.A = PA; q.B = PB; q.strideA = strideA; q.strideB = same ? strideA : strideB;
    q.nrbA = nrbA; q.nrbB = same ? nrbA : nrbB;
    q.C = p.C; q.M = p.M; q.N = p.N; q.ldc = p.ldc; q.alpha = p.alpha;
    q.tiles_m = p.tiles_m; q.tiles_n = p.tiles_n; q.syrk = p.syrk; q.sbw = p.sbw;
    q.slab = nullptr; q.kt_split = 0;
    q.gate_mask = tls_bx_gate_mask;
    // the eigensolver's own products (orthogonal factors: sums of random signs, nothing correlated) keep chains of 8192
    const bool internal = tls_bx_gate_mask == BX_GATE_RANGE;
    q.flush_tiles = internal ? bx_flush_internal() : bx_flush_tiles();
    q.flush_diag = internal ? bx_flush_internal() : bx_flush_diag();
    float beta0 = p.beta;
    if (beta0 != 0.f && beta0 != 1.f) {   // the in-loop flushes add into C: C <- beta C once, then beta = 1
      scale_c_kernel<<<(unsigned)cdiv(p.M * p.N, 256), 256, 0, stream>>>(p.C, p.M, p.N, p.ldc, beta0);
      beta0 = 1.f;
    }
    int st = VIVIT_OK;
    int64_t chunk = 0;
    for (int64_t k0 = 0; k0 < p.K && st == VIVIT_OK; k0 += kc_max, ++chunk) {
      const int64_t kc = (p.K - k0) < kc_max ? (p.K - k0) : kc_max;
      const unsigned gy = (unsigned)cdiv(kc / 16, 4);
      int *flag = flags + chunk;
      if (alay == LAY_K)
        bx_split_kernel<LAY_K><<<dim3((unsigned)nrbA, gy), 256, 0, stream>>>(p.A, p.M, p.lda, k0, kc, PA, strideA, nrbA, flag);
      else
        bx_split_kernel<LAY_M><<<dim3((unsigned)nrbA, gy), 256, 0, stream>>>(p.A, p.M, p.lda, k0, kc, PA, strideA, nrbA, flag);
      if (!same) {
        if (blay == LAY_K)
          bx_split_kernel<LAY_K><<<dim3((unsigned)nrbB, gy), 256, 0, stream>>>(p.B, p.N, p.ldb, k0, kc, PB, strideB, nrbB, flag);
        else
          bx_split_kernel<LAY_M><<<dim3((unsigned)nrbB, gy), 256, 0, stream>>>(p.B, p.N, p.ldb, k0, kc, PB, strideB, nrbB, flag);
      }
      q.K = kc;
      q.beta = k0 == 0 ? beta0 : 1.f;
      // SYRK: only the last chunk mirrors the finished lower tiles into the upper triangle (2 = lower tiles, no mirror)
      q.syrk = (p.syrk == 1 && k0 + kc < p.K) ? 2 : p.syrk;
      q.gate = flag;
      q.sync = sync ? sync + chunk * (int64_t)sync_ints : nullptr;
      if (bx == 6)
        bx_launch6(grid, q, stream);
      else if (bx == 9)
        gemm256_bx_kernel<9><<<grid, 256, GEMM256BX_LDS_BYTES, stream>>>(q);
      else
        gemm256_bx_kernel<3><<<grid, 256, GEMM256BX_LDS_BYTES, stream>>>(q);
      st = launch_status();
      if (st != VIVIT_OK) break;
      // BX_GATE: the same chunk on the fp32 MFMA kernel; every workgroup returns at once unless the chunk is flagged
      GemmArgs f = p;
      f.A = p.A + (alay == LAY_K ? k0 : k0 * p.lda);
      f.B = p.B + (blay == LAY_K ? k0 : k0 * p.ldb);
      f.K = kc; f.kchunk = kc; f.ksplit = 1; f.slab = nullptr;
      f.beta = q.beta; f.syrk = q.syrk;
      f.a_vec = ((reinterpret_cast<uintptr_t>(f.A) & 15) == 0 && (f.lda & 3) == 0) ? 1 : 0;
      f.b_vec = ((reinterpret_cast<uintptr_t>(f.B) & 15) == 0 && (f.ldb & 3) == 0) ? 1 : 0;
      f.gate = flag; f.gate_mask = q.gate_mask;
#if !defined(BX_NO_STANDIN)
      if (alay == LAY_K && blay == LAY_K)
        gemm256_kernel<LAY_K, LAY_K><<<grid, 256, GEMM256_LDS_BYTES, stream>>>(f);
      else if (alay == LAY_K && blay == LAY_M)
        gemm256_kernel<LAY_K, LAY_M><<<grid, 256, GEMM256_LDS_BYTES, stream>>>(f);
      else if (alay == LAY_M && blay == LAY_K)
        gemm256_kernel<LAY_M, LAY_K><<<grid, 256, GEMM256_LDS_BYTES, stream>>>(f);
      else
        gemm256_kernel<LAY_M, LAY_M><<<grid, 256, GEMM256_LDS_BYTES, stream>>>(f);
#endif
      st = launch_status();
    }
    if (st == VIVIT_OK && p.syrk == 1) {
      bx_sym_diag_kernel<<<(unsigned)p.tiles_m, 256, 0, stream>>>(p.C, p.M, p.ldc);
      st = launch_status();
    }
    if (prof) prof_end(0, stream);
    return st;
  }
  if (alay == LAY_K && blay == LAY_K)
    gemm256_kernel<LAY_K, LAY_K><<<grid, 256, GEMM256_LDS_BYTES, stream>>>(p);
  else if (alay == LAY_K && blay == LAY_M)
    gemm256_kernel<LAY_K, LAY_M><<<grid, 256, GEMM256_LDS_BYTES, stream>>>(p);
  else if (alay == LAY_M && blay == LAY_K)
    gemm256_kernel<LAY_M, LAY_K><<<grid, 256, GEMM256_LDS_BYTES, stream>>>(p);
  else
    gemm256_kernel<LAY_M, LAY_M><<<grid, 256, GEMM256_LDS_BYTES, stream>>>(p);
  int st = launch_status();
  if (st == VIVIT_OK && p.ksplit > 1) {
    gemm_reduce_kernel<<<(unsigned)cdiv(p.M * p.N, 256), 256, 0, stream>>>(p.slab, p.C, p.M, p.N, p.ldc, p.ksplit, p.alpha,
                                                                           p.beta, p.syrk, B2);
    st = launch_status();
  }
  if (prof) prof_end(0, stream);
  return st;
}

// ---- bf16-pipe split-K for SMALL outputs with a deep contraction (Gram matrices of small batches: n = 1280,
// P = 4e5 has 15 lower 256-tiles).  The whole operand is split into its three bf16 pieces once (blocked layout: a k
// tile of a 256-row block is 8 KB contiguous per piece, so a K split streams long runs, unlike the 1280 row streams
// 1.6 MB apart of the fp32 operand), one launch with the k tiles divided over blockIdx.y writes partial tiles to a slab,
// and the fixed-order reduce mirrors the lower tiles of a SYRK.
static bool bx_splitk_shape(int64_t M, int64_t N, int64_t K, bool syrk, bool same, int *nsplit_out, int *kt_split_out,
                            size_t *bytes_out) {
  static int forced = -2;
  if (forced == -2) { const char *e = getenv("VIVIT_GEMM_BXSPLITK"); forced = e ? atoi(e) : -1; }
  if (forced == 0 || gemm_split_mode() != 6) return false;
  if (K < 16384 || (K % BK) != 0 || M < 256 || N < 256) return false;
  const int64_t tm = cdiv(M, B2), tn = cdiv(N, B2);
  const int64_t tiles = syrk ? tm * (tm + 1) / 2 : tm * tn;
  if (tiles > 100) return false;                       // enough tiles: the plain bf16-pipe launch fills the chip
  const int64_t ra = cdiv(M, 32) * 32, rb = cdiv(N, 32) * 32;
  const size_t pieces = (size_t)6 * (size_t)K * (size_t)(same ? ra : ra + rb);
  if (pieces > ((size_t)8 << 30)) return false;        // whole-K pieces: bounded scratch
  const int64_t nt = K / BK;
  int64_t s = 512 / tiles;                             // ~2 rounds of one workgroup per CU
  if (s > nt / 64) s = nt / 64;                        // at least 64 k tiles per split
  if (s < 2) return false;
  {
    // Compact grid (map_tile_z: one super-block, every split's tiles on ONE XCD, splits dealt round-robin to the 8 XCDs): the
    // number of splits is 8 g, and what counts is the busiest XCD -- ceil(tiles g / 32) rounds of ceil(nt / 8 g) K tiles on its 32
    // CUs -- plus the slab the reduce has to read back.  (34 splits of 15 tiles gave two XCDs 75 workgroups and six 60: three
    // rounds where the others needed two.)
    const int sbw = syrk ? 16 : sb_width((int)tm, (int)tn), sbh = 256 / sbw;
    const int64_t nsb = syrk ? cdiv(tm, sbh) * (cdiv(tm, sbh) + 1) / 2 : cdiv(tm, sbh) * cdiv(tn, sbw);
    if (nsb == 1) {
      double best = 0.0;
      int64_t gbest = 0;
      for (int64_t g = 1; g <= 16 && nt / (8 * g) >= 64; ++g) {
        const double tile_us = 1.85, slab_us = 8.0 * (double)g * (double)M * (double)N * 8.0 / 5.0e6;
        const double cost = (double)(cdiv(tiles * g, 32) * cdiv(nt, 8 * g)) * tile_us + slab_us;
        if (gbest == 0 || cost < best * 0.97) { best = cost; gbest = g; }   // (fewer splits unless more are worth 3 %)
      }
      if (gbest > 0) s = 8 * gbest;
    }
  }
  const int64_t kts = cdiv(nt, s);
  s = cdiv(nt, kts);
  if (nsplit_out) *nsplit_out = (int)s;
  if (kt_split_out) *kt_split_out = (int)kts;
  if (bytes_out) *bytes_out = pieces + 256 + (size_t)s * (size_t)M * (size_t)N * sizeof(float) + 256 + 256;  // + range flag
  return true;
}

static int bx_splitk_launch(int alay, int blay, const GemmArgs &p, bool syrk, void *workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!gemm256_attrs()) return VIVIT_E_LAUNCH;
  const bool same = p.A == p.B && p.lda == p.ldb && p.M == p.N && alay == blay;
  int nsplit, kts;
  size_t need;
  if (!bx_splitk_shape(p.M, p.N, p.K, syrk, same, &nsplit, &kts, &need)) return VIVIT_E_UNSUPPORTED;
  if (!workspace || workspace_bytes < need) return VIVIT_E_WORKSPACE;
  const int64_t nrbA = cdiv(p.M, 32), nrbB = cdiv(p.N, 32);
  const int64_t strideA = nrbA * 32 * p.K, strideB = nrbB * 32 * p.K;
  unsigned short *PA = static_cast<unsigned short *>(workspace);
  unsigned short *PB = same ? PA : PA + 3 * strideA;
  const size_t pieces = (size_t)6 * (size_t)p.K * (size_t)(same ? nrbA * 32 : (nrbA + nrbB) * 32);
  float *slab = reinterpret_cast<float *>(align_up(reinterpret_cast<uintptr_t>(workspace) + pieces, 256));
  int *flag = reinterpret_cast<int *>(align_up(reinterpret_cast<uintptr_t>(slab) + (size_t)nsplit * (size_t)p.M * (size_t)p.N * sizeof(float), 256));
  if (hipMemsetAsync(flag, 0, 4, stream) != hipSuccess) return VIVIT_E_LAUNCH;
  const bool prof = syrk && same && prof_enabled() && bx_public_product();
  if (prof) prof_begin(0, (double)p.M * (double)(p.M + 1) * (double)p.K, stream);
  const unsigned gy = (unsigned)cdiv(p.K / 16, 4);
  if (alay == LAY_K)
    bx_split_kernel<LAY_K><<<dim3((unsigned)nrbA, gy), 256, 0, stream>>>(p.A, p.M, p.lda, 0, p.K, PA, strideA, nrbA, flag);
  else
    bx_split_kernel<LAY_M><<<dim3((unsigned)nrbA, gy), 256, 0, stream>>>(p.A, p.M, p.lda, 0, p.K, PA, strideA, nrbA, flag);
  if (!same) {
    if (blay == LAY_K)
      bx_split_kernel<LAY_K><<<dim3((unsigned)nrbB, gy), 256, 0, stream>>>(p.B, p.N, p.ldb, 0, p.K, PB, strideB, nrbB, flag);
    else
      bx_split_kernel<LAY_M><<<dim3((unsigned)nrbB, gy), 256, 0, stream>>>(p.B, p.N, p.ldb, 0, p.K, PB, strideB, nrbB, flag);
  }
  GemmBxArgs q;
  q.A = PA; q.B = PB; q.strideA = strideA; q.strideB = same ? strideA : strideB;
  q.nrbA = nrbA; q.nrbB = same ? nrbA : nrbB;
  q.C = p.C; q.M = p.M; q.N = p.N; q.K = p.K; q.ldc = p.ldc; q.alpha = 1.f; q.beta = 0.f;
  q.tiles_m = (int)cdiv(p.M, B2); q.tiles_n = (int)cdiv(p.N, B2);
  q.syrk = syrk ? 2 : 0;   // lower tiles only; the reduce mirrors
  const int sbw = syrk ? 16 : sb_width(q.tiles_m, q.tiles_n), sbh = 256 / sbw;
  q.sbw = sbw;
  q.slab = slab; q.kt_split = kts;
  q.gate = flag; q.gate_mask = tls_bx_gate_mask;
  q.flush_tiles = bx_flush_splitk();
  q.flush_diag = bx_flush_diag();
  q.sync = nullptr;
  const int64_t sbm = cdiv(q.tiles_m, sbh), sbn = cdiv(q.tiles_n, sbw);
  const int64_t nsb = syrk ? sbm * (sbm + 1) / 2 : sbm * sbn;
  dim3 grid((unsigned)(nsb * 256), (unsigned)nsplit);
  if (nsb == 1) {   // one (partial) super-block: the compact grid of map_tile_z, all tiles of a split on one XCD
    const int64_t d = q.tiles_m < q.tiles_n ? q.tiles_m : q.tiles_n;
    const int64_t v = syrk ? d * (d + 1) / 2 : (int64_t)q.tiles_m * q.tiles_n;
    grid = dim3((unsigned)(8 * v * cdiv(nsplit, 8)), 1);
    q.sbw = -sbw;
  }
  bx_launch6(grid, q, stream);
  int st = launch_status();
  if (st != VIVIT_OK) return st;
  {  // BX_GATE: the fp32 MFMA kernel with the same K split and slab; returns at once unless the operand is flagged
    GemmArgs f = p;
    f.ksplit = nsplit; f.kchunk = (int64_t)kts * BK; f.slab = slab;
    f.tiles_m = q.tiles_m; f.tiles_n = q.tiles_n; f.syrk = q.syrk; f.sbw = q.sbw; f.desc = nullptr;
    f.a_vec = f.b_vec = 1;
    f.gate = flag; f.gate_mask = q.gate_mask;
    if (alay == LAY_K && blay == LAY_K)
      gemm256_kernel<LAY_K, LAY_K><<<grid, 256, GEMM256_LDS_BYTES, stream>>>(f);
    else if (alay == LAY_K && blay == LAY_M)
      gemm256_kernel<LAY_K, LAY_M><<<grid, 256, GEMM256_LDS_BYTES, stream>>>(f);
    else if (alay == LAY_M && blay == LAY_K)
      gemm256_kernel<LAY_M, LAY_K><<<grid, 256, GEMM256_LDS_BYTES, stream>>>(f);
    else
      gemm256_kernel<LAY_M, LAY_M><<<grid, 256, GEMM256_LDS_BYTES, stream>>>(f);
    st = launch_status();
    if (st != VIVIT_OK) return st;
  }
  gemm_reduce_kernel<<<(unsigned)cdiv(p.M * p.N, 256), 256, 0, stream>>>(slab, p.C, p.M, p.N, p.ldc, nsplit, p.alpha, p.beta,
                                                                       syrk ? 1 : 0, B2);
  st = launch_status();
  if (st == VIVIT_OK && syrk) {
    bx_sym_diag_kernel<<<(unsigned)q.tiles_m, 256, 0, stream>>>(p.C, p.M, p.ldc);
    st = launch_status();
  }
  if (prof) prof_end(0, stream);
  return st;
}

// 64-row streaming kernel: split-K so that ~2 workgroups per CU exist (one resident at a time: 140 KB LDS)
static bool use_gemm64(int alay, int blay, const float *A, const float *B, int64_t M, int64_t N, int64_t K, int64_t lda,
                       int64_t ldb) {
  static int forced = -2;
  if (forced == -2) {
    const char *e = getenv("VIVIT_GEMM64");
    forced = e ? atoi(e) : -1;
  }
  if (forced == 0) return false;
  const bool vec = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0 &&
                   (ldb & 3) == 0 && (alay == LAY_K || ((M & 3) == 0 && M >= 4)) && (blay == LAY_K || ((N & 3) == 0 && N >= 4));
  return vec && M <= 64 && N >= 2048 && K >= 2048 && (K % BK) == 0;
}

// products of the 64-row streaming kernel on the bf16 pipe (exact three-way splits) unless the fp32 pipe is asked for
// (VIVIT_GEMM_SPLIT=0 or VIVIT_GEMM64_BX=0)
static bool gemm64_bx_enabled() {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("VIVIT_GEMM64_BX");
    on = (e ? atoi(e) != 0 : true) && gemm_split_mode() != 0;
  }
  return on != 0;
}

static size_t gemm64_workspace_bytes(int64_t M, int64_t N, int64_t K, int *ksplit_out, int64_t *kchunk_out, size_t *slab_out) {
  const int64_t tiles = cdiv(N, 256), ktiles = K / BK;
  // one workgroup per CU: pick the split count (>= 2 rounds of work, every split >= 64 K tiles) whose last
  // round of 256 workgroups is fullest
  int64_t s = 1;
  double best = 0.0;
  for (int64_t c = 1; c <= 24; ++c) {
    if (c > 1 && ktiles / c < 64) break;
    const int64_t wgs = tiles * c;
    const double fill = (double)wgs / (double)(256 * cdiv(wgs, 256));
    const double score = wgs >= 512 ? fill : fill * 0.5 * (double)wgs / 512.0;  // too few workgroups: latency-bound
    if (score > best + 1e-9) { best = score; s = c; }
  }
  const int64_t kchunk = cdiv(ktiles, s) * BK;
  const int ksplit = (int)cdiv(K, kchunk);
  if (ksplit_out) *ksplit_out = ksplit;
  if (kchunk_out) *kchunk_out = kchunk;
  const size_t slab = ksplit > 1 ? (size_t)ksplit * (size_t)M * (size_t)N * sizeof(float) : 0;
  if (slab_out) *slab_out = slab;
  // + the bf16 pieces of the 64-row operand (gemm64_bx_kernel): 6 KB per K tile, after the slab
  return gemm64_bx_enabled() ? align_up(slab, 256) + (size_t)ktiles * 6144 : slab;
}

static int gemm64_launch(int alay, int blay, GemmArgs p, void *workspace, size_t workspace_bytes, hipStream_t stream) {
  static unsigned long long attr_done = 0;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return VIVIT_E_LAUNCH;
    if (!(attr_done & (1ull << (dev & 63)))) {
      const void *fns[4] = {reinterpret_cast<const void *>(gemm64_dma_kernel<LAY_K, LAY_K>),
                            reinterpret_cast<const void *>(gemm64_dma_kernel<LAY_K, LAY_M>),
                            reinterpret_cast<const void *>(gemm64_dma_kernel<LAY_M, LAY_K>),
                            reinterpret_cast<const void *>(gemm64_dma_kernel<LAY_M, LAY_M>)};
      for (int f = 0; f < 4; ++f)
        if (!ensure_dynamic_lds(fns[f], GEMM64_LDS_BYTES, attr_done)) return VIVIT_E_LAUNCH;
      if (!ensure_dynamic_lds(reinterpret_cast<const void *>(gemm64_bx_kernel<LAY_K>), GEMM64X_LDS_BYTES, attr_done) ||
          !ensure_dynamic_lds(reinterpret_cast<const void *>(gemm64_bx_kernel<LAY_M>), GEMM64X_LDS_BYTES, attr_done))
        return VIVIT_E_LAUNCH;
      attr_done |= 1ull << (dev & 63);
    }
  }
  size_t slab_bytes = 0;
  const size_t need = gemm64_workspace_bytes(p.M, p.N, p.K, &p.ksplit, &p.kchunk, &slab_bytes);
  p.slab = nullptr;
  if (p.ksplit > 1) {
    if (!workspace || workspace_bytes < slab_bytes) return VIVIT_E_WORKSPACE;
    p.slab = static_cast<float *>(workspace);
  }
  // The bf16-pipe form needs room for the pieces of A behind the slab.  Which kernel runs depends on SHAPE and ENVIRONMENT
  // only (results are bit-identical from call to call, include/vivit_hip.h): a workspace smaller than the query's answer is
  // refused -- it is never a silent switch to the fp32 kernel, whose summation order (and speed) differs.
  uint4 *apieces = nullptr;
  if (gemm64_bx_enabled()) {
    if (!workspace || workspace_bytes < need) return VIVIT_E_WORKSPACE;
    apieces = reinterpret_cast<uint4 *>(static_cast<char *>(workspace) + align_up(slab_bytes, 256));
  }
  p.tiles_m = 1;
  p.tiles_n = (int)cdiv(p.N, 256);
  p.syrk = 0;
  p.desc = nullptr;
  dim3 grid((unsigned)p.tiles_n, (unsigned)p.ksplit, 1);
  if (apieces) {   // products on the bf16 pipe: split the 64-row operand once, then stream
    if (alay == LAY_K)
      g64_split_a_kernel<LAY_K><<<(unsigned)(p.K / BK), 128, 0, stream>>>(p.A, p.lda, p.M, apieces);
    else
      g64_split_a_kernel<LAY_M><<<(unsigned)(p.K / BK), 128, 0, stream>>>(p.A, p.lda, p.M, apieces);
    if (blay == LAY_K)
      gemm64_bx_kernel<LAY_K><<<grid, 256, GEMM64X_LDS_BYTES, stream>>>(p, apieces);
    else
      gemm64_bx_kernel<LAY_M><<<grid, 256, GEMM64X_LDS_BYTES, stream>>>(p, apieces);
  }
  else if (alay == LAY_K && blay == LAY_K)
    gemm64_dma_kernel<LAY_K, LAY_K><<<grid, 256, GEMM64_LDS_BYTES, stream>>>(p);
  else if (alay == LAY_K && blay == LAY_M)
    gemm64_dma_kernel<LAY_K, LAY_M><<<grid, 256, GEMM64_LDS_BYTES, stream>>>(p);
  else if (alay == LAY_M && blay == LAY_K)
    gemm64_dma_kernel<LAY_M, LAY_K><<<grid, 256, GEMM64_LDS_BYTES, stream>>>(p);
  else
    gemm64_dma_kernel<LAY_M, LAY_M><<<grid, 256, GEMM64_LDS_BYTES, stream>>>(p);
  int st = launch_status();
  if (st != VIVIT_OK) return st;
  if (p.ksplit > 1) {
    gemm_reduce_kernel<<<(unsigned)cdiv(p.M * p.N, 256), 256, 0, stream>>>(p.slab, p.C, p.M, p.N, p.ldc, p.ksplit, p.alpha,
                                                                           p.beta, 0);
    st = launch_status();
  }
  return st;
}

int gemm_launch(int alay, int blay, const float *A, const float *B, float *C, int64_t M, int64_t N,
                int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float alpha, float beta, bool syrk,
                void *workspace, size_t workspace_bytes, hipStream_t stream) {
  if (M < 0 || N < 0 || K < 0) return VIVIT_E_BADARG;
  if (M == 0 || N == 0) return VIVIT_OK;
  if (!C || ldc < N) return VIVIT_E_BADARG;
  if (K == 0) {  // empty contraction: C = beta * C
    scale_c_kernel<<<(unsigned)cdiv(M * N, 256), 256, 0, stream>>>(C, M, N, ldc, beta);
    return launch_status();
  }
  if (!A || !B) return VIVIT_E_BADARG;
  if (lda < (alay == LAY_K ? K : M) || ldb < (blay == LAY_K ? K : N)) return VIVIT_E_BADARG;
  if (syrk && (M != N || alay != blay)) return VIVIT_E_BADARG;

  GemmArgs p;
  p.A = A; p.B = B; p.C = C;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.alpha = alpha; p.beta = beta;
  {
    const bool vec = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0 &&
                     (ldb & 3) == 0 && (alay == LAY_K || (M & 3) == 0) && (blay == LAY_K || (N & 3) == 0);
    const int64_t Kmain = K / BK * BK;
    if (vec && gemm256_plan(M, N, Kmain, syrk, nullptr, nullptr)) {
      p.K = Kmain;
      int st = gemm256_launch(alay, blay, p, syrk, workspace, workspace_bytes, stream);
      if (st != VIVIT_OK || Kmain == K) return st;
      // ragged K tail (< 16) through the small-tile kernel, accumulating
      const float *At = A + (alay == LAY_K ? Kmain : Kmain * lda), *Bt = B + (blay == LAY_K ? Kmain : Kmain * ldb);
      return gemm_launch(alay, blay, At, Bt, C, M, N, K - Kmain, lda, ldb, ldc, alpha, 1.f, syrk, workspace, workspace_bytes, stream);
    }
  }
  {
    const bool vec = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0 &&
                     (ldb & 3) == 0 && (alay == LAY_K || (M & 3) == 0) && (blay == LAY_K || (N & 3) == 0);
    const int64_t Kmain = K / BK * BK;
    const bool same = A == B && lda == ldb && M == N && alay == blay;
    size_t need = 0;
    if (vec && bx_splitk_shape(M, N, Kmain, syrk, same, nullptr, nullptr, &need) && workspace && workspace_bytes >= need) {
      p.K = Kmain;
      int st = bx_splitk_launch(alay, blay, p, syrk, workspace, workspace_bytes, stream);
      if (st != VIVIT_OK || Kmain == K) return st;
      const float *At = A + (alay == LAY_K ? Kmain : Kmain * lda), *Bt = B + (blay == LAY_K ? Kmain : Kmain * ldb);
      return gemm_launch(alay, blay, At, Bt, C, M, N, K - Kmain, lda, ldb, ldc, alpha, 1.f, syrk, workspace, workspace_bytes, stream);
    }
  }
  if (!syrk && use_gemm64(alay, blay, A, B, M, N, K, lda, ldb)) return gemm64_launch(alay, blay, p, workspace, workspace_bytes, stream);
  if (use_tsk(alay, blay, A, B, M, N, K, lda, ldb, syrk))
    return tsk_launch(A, B, C, M, N, K, lda, ldb, ldc, alpha, beta, workspace, workspace_bytes, stream);
  choose_split(M, N, K, syrk, p.ksplit, p.kchunk);
  p.slab = nullptr;
  if (p.ksplit > 1) {
    const size_t need = (size_t)p.ksplit * (size_t)M * (size_t)N * sizeof(float);
    if (!workspace || workspace_bytes < need) return VIVIT_E_WORKSPACE;
    p.slab = static_cast<float *>(workspace);
  }
  const int wm = pick_wm(M, N, syrk);
  p.tiles_m = (int)cdiv(M, 64 * wm);
  p.tiles_n = (int)cdiv(N, 256 / wm);
  p.syrk = syrk ? 1 : 0;
  p.a_vec = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
  p.b_vec = ((reinterpret_cast<uintptr_t>(B) & 15) == 0 && (ldb & 3) == 0) ? 1 : 0;
  p.desc = nullptr;

  const int sbw = syrk ? 16 : sb_width(p.tiles_m, p.tiles_n), sbh = 256 / sbw;
  p.sbw = sbw;
  const int64_t sbm = cdiv(p.tiles_m, sbh), sbn = cdiv(p.tiles_n, sbw);
  const int64_t nsb = syrk ? sbm * (sbm + 1) / 2 : sbm * sbn;
  if (nsb * 256 > 0x7fffffffLL) return VIVIT_E_UNSUPPORTED;
  dim3 grid((unsigned)(nsb * 256), (unsigned)p.ksplit, 1);
  dim3 block(256, 1, 1);
  const bool prof = syrk && A == B && prof_enabled() && bx_public_product();  // only the caller's Gram SYRK is profiled as such
  if (prof) prof_begin(0, (double)M * (double)(M + 1) * (double)K, stream);
  if (wm == 1) {
    if (alay == LAY_K && blay == LAY_K)
      gemm_kernel<LAY_K, LAY_K, 1><<<grid, block, 0, stream>>>(p);
    else if (alay == LAY_K && blay == LAY_M)
      gemm_kernel<LAY_K, LAY_M, 1><<<grid, block, 0, stream>>>(p);
    else if (alay == LAY_M && blay == LAY_K)
      gemm_kernel<LAY_M, LAY_K, 1><<<grid, block, 0, stream>>>(p);
    else
      gemm_kernel<LAY_M, LAY_M, 1><<<grid, block, 0, stream>>>(p);
  } else if (alay == LAY_K && blay == LAY_K)
    gemm_kernel<LAY_K, LAY_K><<<grid, block, 0, stream>>>(p);
  else if (alay == LAY_K && blay == LAY_M)
    gemm_kernel<LAY_K, LAY_M><<<grid, block, 0, stream>>>(p);
  else if (alay == LAY_M && blay == LAY_K)
    gemm_kernel<LAY_M, LAY_K><<<grid, block, 0, stream>>>(p);
  else
    gemm_kernel<LAY_M, LAY_M><<<grid, block, 0, stream>>>(p);
  if (prof) prof_end(0, stream);
  int st = launch_status();
  if (st != VIVIT_OK) return st;
  if (p.ksplit > 1) {
    gemm_reduce_kernel<<<(unsigned)cdiv(M * N, 256), 256, 0, stream>>>(p.slab, C, M, N, ldc, p.ksplit, alpha,
                                                                       beta, p.syrk);
    st = launch_status();
  }
  return st;
}

int gemm_lower_launch(const float *A, const float *B, float *C, int64_t n, int64_t K, int64_t lda, int64_t ldb,
                      int64_t ldc, float alpha, float beta, hipStream_t stream) {
  if (n <= 0) return VIVIT_OK;
  if (!A || !B || !C || K <= 0 || lda < n || ldb < n || ldc < n) return VIVIT_E_BADARG;
  GemmArgs p;
  p.A = A; p.B = B; p.C = C;
  p.M = n; p.N = n; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.alpha = alpha; p.beta = beta;
  p.ksplit = 1;
  p.kchunk = cdiv(K, BK) * BK;
  p.slab = nullptr;
  p.tiles_m = p.tiles_n = (int)cdiv(n, BM);
  p.syrk = 2;
  p.a_vec = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
  p.b_vec = ((reinterpret_cast<uintptr_t>(B) & 15) == 0 && (ldb & 3) == 0) ? 1 : 0;
  p.desc = nullptr;
  p.sbw = SB;
  const int64_t sbm = cdiv(p.tiles_m, SB);
  const int64_t nsb = sbm * (sbm + 1) / 2;
  gemm_kernel<LAY_M, LAY_M><<<dim3((unsigned)(nsb * 256), 1, 1), 256, 0, stream>>>(p);
  return launch_status();
}

int gemm_batched_launch(int alay, int blay, const GemmDesc *desc, int batch, int64_t maxM, int64_t maxN, float alpha,
                        float beta, hipStream_t stream) {
  if (batch <= 0 || maxM <= 0 || maxN <= 0) return VIVIT_OK;
  if (!desc) return VIVIT_E_BADARG;
  GemmArgs p;
  p.A = nullptr; p.B = nullptr; p.C = nullptr;
  p.M = maxM; p.N = maxN; p.K = 0; p.lda = p.ldb = p.ldc = 0;
  p.alpha = alpha; p.beta = beta;
  p.ksplit = 1;
  p.kchunk = BK;
  p.slab = nullptr;
  p.tiles_m = (int)cdiv(maxM, BM);
  p.tiles_n = (int)cdiv(maxN, BN);
  p.syrk = 0;
  p.a_vec = p.b_vec = 0;
  p.desc = desc;
  const int sbw = sb_width(p.tiles_m, p.tiles_n), sbh = 256 / sbw;
  p.sbw = sbw;
  const int64_t nsb = cdiv(p.tiles_m, sbh) * cdiv(p.tiles_n, sbw);
  if (nsb * 256 > 0x7fffffffLL || batch > 65535) return VIVIT_E_UNSUPPORTED;
  dim3 grid((unsigned)(nsb * 256), 1, (unsigned)batch);
  if (alay == LAY_K && blay == LAY_K)
    gemm_kernel<LAY_K, LAY_K><<<grid, 256, 0, stream>>>(p);
  else if (alay == LAY_K && blay == LAY_M)
    gemm_kernel<LAY_K, LAY_M><<<grid, 256, 0, stream>>>(p);
  else if (alay == LAY_M && blay == LAY_K)
    gemm_kernel<LAY_M, LAY_K><<<grid, 256, 0, stream>>>(p);
  else
    gemm_kernel<LAY_M, LAY_M><<<grid, 256, 0, stream>>>(p);
  return launch_status();
}

} // namespace vivit

namespace vivit {

// dout[i] = beta G[i][i] + alpha sum_k A[i][k]^2, fp64 accumulation, one workgroup per row (16 float4 loads in flight per thread)
template <bool VEC>
__global__ __launch_bounds__(256) void syrk_diag_kernel(const float *__restrict__ A, int64_t K, int64_t lda, const float *__restrict__ G,
                                                        int64_t ldg, float alpha, float beta, float *__restrict__ dout) {
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  const float *a = A + row * lda;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if constexpr (VEC) {
    const int64_t K4 = K / 4;
    int64_t c = tid;
    for (; c + 3 * 256 < K4; c += 4 * 256) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(a + 4 * (c + 256 * u));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[0] = fma((double)v[u].x, (double)v[u].x, acc[0]);
        acc[1] = fma((double)v[u].y, (double)v[u].y, acc[1]);
        acc[2] = fma((double)v[u].z, (double)v[u].z, acc[2]);
        acc[3] = fma((double)v[u].w, (double)v[u].w, acc[3]);
      }
    }
    for (; c < K4; c += 256) {
      const float4 v = *reinterpret_cast<const float4 *>(a + 4 * c);
      acc[0] = fma((double)v.x, (double)v.x, acc[0]);
      acc[1] = fma((double)v.y, (double)v.y, acc[1]);
      acc[2] = fma((double)v.z, (double)v.z, acc[2]);
      acc[3] = fma((double)v.w, (double)v.w, acc[3]);
    }
    for (int64_t k = 4 * K4 + tid; k < K; k += 256) acc[0] = fma((double)a[k], (double)a[k], acc[0]);
  } else {
    for (int64_t k = tid; k < K; k += 256) acc[k & 3] = fma((double)a[k], (double)a[k], acc[k & 3]);
  }
  double sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  if (tid == 0) {
    const double tot = (red[0] + red[1]) + (red[2] + red[3]);
    const double old = beta != 0.f ? (double)beta * (double)G[row * ldg + row] : 0.0;
    dout[row] = (float)(old + (double)alpha * tot);
  }
}

__global__ __launch_bounds__(256) void syrk_diag_store_kernel(float *__restrict__ G, int64_t ldg, int64_t n, const float *__restrict__ d) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) G[i * ldg + i] = d[i];
}

}  // namespace vivit

using namespace vivit;

extern "C" {

#if defined(BX_STAMP)
int vivit_debug_bx_stamp_buffer(void *buf, unsigned int capacity) {
  unsigned long long *b = static_cast<unsigned long long *>(buf);
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_bx_stamp), &b, sizeof(b)) != hipSuccess) return VIVIT_E_LAUNCH;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_bx_stamp_cap), &capacity, sizeof(capacity)) != hipSuccess) return VIVIT_E_LAUNCH;
  return VIVIT_OK;
}
#endif

int vivit_gemm_split_mode(void) { return gemm_split_mode(); }

// The DIAGONAL of the public Gram product is computed on its own: G[i][i] = beta G[i][i] + alpha sum_k a_ik^2 is the one
// place where every term of the sum has the same sign, i.e. where the bf16 MFMA's truncation of aligned partial sums
// (see bx_flush_tiles) adds up to a bias instead of averaging out -- the reason the diagonal TILES ran chains of 512 k
// (+ 140 ms of flushes at the headline shape).  One streaming pass over A (66.7 GB: ~16 ms) with fp64 accumulation
// gives correctly rounded diagonal entries for every path, and the diagonal tiles keep the chain length of all others.
static size_t syrk_diag_bytes(int64_t n) { return align_up(sizeof(float) * (size_t)n, 256) + 256; }
constexpr int64_t SYRK_DIAG_MIN_K = 1024;   // below: the product's own diagonal (chains never end; nothing to fix)

size_t vivit_gram_syrk_f32_workspace_bytes(int64_t n, int64_t p) {
  BxStrictScope strict;   // sized for the public launch (one accumulation chain per chunk of the operand pieces)
  return gemm_workspace_bytes(n, n, p, true) + syrk_diag_bytes(n);
}

int vivit_gram_syrk_f32(const float *A, int64_t n, int64_t p, int64_t lda, float *G, int64_t ldg, float alpha,
                        float beta, void *workspace, size_t workspace_bytes, void *stream) {
  BxStrictScope strict;  // public product: both range bits reroute a chunk to the fp32 MFMA kernel
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool fix_diag = A && G && workspace && n > 0 && p >= SYRK_DIAG_MIN_K && workspace_bytes >= syrk_diag_bytes(n) + gemm_workspace_bytes(n, n, p, true);
  float *diag = nullptr;
  if (fix_diag) {
    diag = reinterpret_cast<float *>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
    const bool vec = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0;
    if (vec) syrk_diag_kernel<true><<<(unsigned)n, 256, 0, s>>>(A, p, lda, G, ldg, alpha, beta, diag);
    else syrk_diag_kernel<false><<<(unsigned)n, 256, 0, s>>>(A, p, lda, G, ldg, alpha, beta, diag);
    workspace = reinterpret_cast<char *>(workspace) + syrk_diag_bytes(n);
    workspace_bytes -= syrk_diag_bytes(n);
  }
  const int st = gemm_launch(LAY_K, LAY_K, A, A, G, n, n, p, lda, lda, ldg, alpha, beta, true, workspace, workspace_bytes, s);
  if (st != VIVIT_OK || !fix_diag) return st;
  syrk_diag_store_kernel<<<(unsigned)cdiv(n, 256), 256, 0, s>>>(G, ldg, n, diag);
  return launch_status();
}

size_t vivit_gemm_f32_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  BxStrictScope strict;   // sized for the public launch
  size_t b = gemm_workspace_bytes(m, n, k, false);
  if (skinny_applicable(m, n, k)) {
    const size_t sb = skinny_workspace_bytes(m, k, n);
    if (sb > b) b = sb;
  }
  return b;
}

int vivit_gemm_nt_f32(const float *A, const float *B, float *C, int64_t m, int64_t n, int64_t k, int64_t lda,
                      int64_t ldb, int64_t ldc, float alpha, float beta, void *workspace, size_t workspace_bytes,
                      void *stream) {
  BxStrictScope strict;  // public product: both range bits reroute a chunk to the fp32 MFMA kernel
  return gemm_launch(LAY_K, LAY_K, A, B, C, m, n, k, lda, ldb, ldc, alpha, beta, false, workspace, workspace_bytes,
                     static_cast<hipStream_t>(stream));
}

int vivit_gemm_nn_f32(const float *A, const float *B, float *C, int64_t m, int64_t n, int64_t k, int64_t lda,
                      int64_t ldb, int64_t ldc, float alpha, float beta, void *workspace, size_t workspace_bytes,
                      void *stream) {
  BxStrictScope strict;
  // few output rows: HBM-bound streaming kernel instead of a mostly idle MFMA tile (K7/K8)
  if (skinny_applicable(m, n, k) && A && B && C && lda >= k && ldb >= n && ldc >= n &&
      workspace_bytes >= skinny_workspace_bytes(m, k, n) && workspace)
    return skinny_nn_launch(A, lda, B, ldb, C, ldc, m, k, n, alpha, beta, workspace, workspace_bytes,
                            static_cast<hipStream_t>(stream));
  return gemm_launch(LAY_K, LAY_M, A, B, C, m, n, k, lda, ldb, ldc, alpha, beta, false, workspace, workspace_bytes,
                     static_cast<hipStream_t>(stream));
}

int vivit_gemm_tn_f32(const float *A, const float *B, float *C, int64_t m, int64_t n, int64_t k, int64_t lda,
                      int64_t ldb, int64_t ldc, float alpha, float beta, void *workspace, size_t workspace_bytes,
                      void *stream) {
  BxStrictScope strict;  // public product: both range bits reroute a chunk to the fp32 MFMA kernel
  return gemm_launch(LAY_M, LAY_M, A, B, C, m, n, k, lda, ldb, ldc, alpha, beta, false, workspace, workspace_bytes,
                     static_cast<hipStream_t>(stream));
}

} // extern "C"
