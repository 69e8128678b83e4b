// Factor materialisation kernels: the per-sample parameter Jacobian-transpose products of Linear and Conv2d weights,
// V_t[v, n, *param] = d(out_n . M[v, n]) / d(param)   (BackPACK's `param_mjp(..., sum_batch=False)`, called by the
// reference at vivit/extensions/secondorder/vivit/base.py:84-92; Linear: einsum("vno,ni->vnoi"), Conv2d: unfold +
// einsum("vnol,nkl->vnok")).  Both are bound by writing V (4 n P bytes): the Linear kernel is a pure store stream, the
// Conv2d kernel gathers the input patch values on the fly (no im2col buffer) from the L1/L2-resident sample.
#include <cstdlib>

#include "common.h"

namespace vivit {

// V[(v,n)][o][i] = s[(v,n)][o] * z[n][i]; one thread per 4 consecutive i (float4 store when I % 4 == 0)
template <bool VEC>
__global__ __launch_bounds__(256) void linear_weight_mjp_kernel(const float *__restrict__ s, const float *__restrict__ z,
                                                                float *__restrict__ V, int64_t rows, int64_t N, int64_t O,
                                                                int64_t I) {
  const int64_t per_row = VEC ? (O * I) >> 2 : O * I;
  const int64_t total = rows * per_row;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / per_row;
    const int64_t e = (idx - row * per_row) * (VEC ? 4 : 1);
    const int64_t o = e / I, i = e - o * I;
    const float sv = s[row * O + o];
    const float *zr = z + (row % N) * I + i;
    if (VEC) {
      const float4 zv = *reinterpret_cast<const float4 *>(zr);
      *reinterpret_cast<float4 *>(V + row * O * I + e) = make_float4(sv * zv.x, sv * zv.y, sv * zv.z, sv * zv.w);
    } else {
      V[row * O * I + e] = sv * zr[0];
    }
  }
}

struct Conv2dGeom {
  int Cin, H, W, Cout, KH, KW, OH, OW, sh, sw, ph, pw, dh, dw;
};

// V[(v,n)][o][c][kh][kw] = sum_{oh,ow} M[(v,n)][o][oh][ow] * x[n][c][oh*sh - ph + kh*dh][ow*sw - pw + kw*dw]
// grid.x = (v,n) row, grid.y = chunk of 256 patch positions k = (c, kh, kw), grid.z = group of CV_OT output channels.
// A thread owns ONE patch position and CV_OT output channels in registers: the input value of an output position is
// loaded once (L1 / L2: the sample's planes are small) and used for CV_OT products; M is staged through LDS as
// [position][channel] so that the CV_OT channel values of a position are two broadcast ds_read_b128.  (The first version
// -- one output per thread, one LDS read and one global read per multiply-add -- ran at 2 TFLOP/s and was 70 of the
// 118 ms of config 4's factor back-propagation.)
constexpr int CV_LC = 64;   // output positions per staged chunk
constexpr int CV_OT = 8;    // output channels per thread
__global__ __launch_bounds__(256) void conv2d_weight_mjp_kernel(const float *__restrict__ M, const float *__restrict__ x,
                                                                float *__restrict__ V, int64_t N, Conv2dGeom g) {
  __shared__ __attribute__((aligned(16))) float sM[CV_LC * CV_OT];   // [position][channel of this group]
  const int64_t row = blockIdx.x;
  const int64_t n = row % N;
  const int K = g.Cin * g.KH * g.KW;
  const int L = g.OH * g.OW;
  const int kk = blockIdx.y * 256 + threadIdx.x;
  const int o0 = blockIdx.z * CV_OT;
  const bool active = kk < K;
  const int c = active ? kk / (g.KH * g.KW) : 0;
  const int kh = active ? (kk / g.KW) % g.KH : 0, kw = active ? kk % g.KW : 0;
  const float *Mrow = M + row * (int64_t)g.Cout * L;
  const float *xc = x + (n * g.Cin + c) * (int64_t)g.H * g.W;
  const int ih0 = kh * g.dh - g.ph, iw0 = kw * g.dw - g.pw;
  float acc[CV_OT];
#pragma unroll
  for (int t = 0; t < CV_OT; ++t) acc[t] = 0.f;
  for (int l0 = 0; l0 < L; l0 += CV_LC) {
    const int lc = (L - l0) < CV_LC ? (L - l0) : CV_LC;
    __syncthreads();
    for (int idx = threadIdx.x; idx < CV_OT * CV_LC; idx += 256) {   // coalesced along the positions of a channel
      const int t = idx / CV_LC, ll = idx - t * CV_LC;
      sM[ll * CV_OT + t] = (ll < lc && o0 + t < g.Cout) ? Mrow[(int64_t)(o0 + t) * L + l0 + ll] : 0.f;
    }
    __syncthreads();
    if (active) {
      int oh = l0 / g.OW, ow = l0 - oh * g.OW;
      for (int ll = 0; ll < lc; ++ll) {
        const int ih = oh * g.sh + ih0, iw = ow * g.sw + iw0;
        if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) {
          const float xv = xc[ih * g.W + iw];
          const float4 m0 = *reinterpret_cast<const float4 *>(sM + ll * CV_OT);
          const float4 m1 = *reinterpret_cast<const float4 *>(sM + ll * CV_OT + 4);
          acc[0] += m0.x * xv; acc[1] += m0.y * xv; acc[2] += m0.z * xv; acc[3] += m0.w * xv;
          acc[4] += m1.x * xv; acc[5] += m1.y * xv; acc[6] += m1.z * xv; acc[7] += m1.w * xv;
        }
        if (++ow == g.OW) { ow = 0; ++oh; }
      }
    }
  }
  if (active) {
#pragma unroll
    for (int t = 0; t < CV_OT; ++t)
      if (o0 + t < g.Cout) V[row * (int64_t)g.Cout * K + (int64_t)(o0 + t) * K + kk] = acc[t];
  }
}

// ---- the same rule on the fp32 matrix pipe (round 4) -----------------------------------------------------------------------
// Per (v, n) row the rule is a small GEMM  V[o][k] = sum_l M[o][l] P[l][k]  with the patch matrix P[l][k = (c, kh, kw)] =
// x[c][oh sh - ph + kh dh][ow sw - pw + kw dw]: 16..64 x 27..576 x 64..1024 in ResNet-32.  One workgroup per row (and group
// of 64 output channels) stages the sample's input planes WITH their zero border ([Cin][H + 2 ph][W + 2 pw]: the gather
// needs no bounds test) and the row's M transposed ([l][o], stride 16 RT + 1) in LDS; a wave owns 16-wide column tiles of
// V and runs v_mfma_f32_16x16x4_f32 over l: the A value of a lane is M[o = 16 rt + lane % 16][l = l0 + lane / 16], the B
// value the gathered x for (k = 16 ct + lane % 16, l = l0 + lane / 16), whose LDS address advances incrementally with l
// (ow += 4 with one wrap, hence OW >= 4).  One gather feeds RT = Cout / 16 MFMAs.  (The scalar kernel above: 8 FMAs per
// global + two LDS loads, 13 TFLOP/s, 10.5 of the 57 ms of config 4's factor back-propagation.)
constexpr int CWM_OC = 64;        // output channels per workgroup (RT <= 4 row tiles)
constexpr int CWM_THREADS = 1024;  // 16 waves: four per SIMD hide the LDS latency of the gather
constexpr int CWM_U = 8;           // steps (of four positions) whose LDS reads are in flight together
typedef float cwm_f32x4 __attribute__((ext_vector_type(4)));

// Work items of a workgroup: (column tile ct, split sp of the position range); S > 1 only when there are fewer column
// tiles than waves -- the partial tiles then go through the LDS region of M (dead after the products) and are summed in
// a fixed order.
template <int RT>
__global__ __launch_bounds__(CWM_THREADS) void conv2d_weight_mjp_mfma_kernel(const float *__restrict__ M, const float *__restrict__ x,
                                                                             float *__restrict__ V, int64_t N, Conv2dGeom g, int S) {
  extern __shared__ __attribute__((aligned(16))) float cwm_smem[];
  const int Hp = g.H + 2 * g.ph, Wp = g.W + 2 * g.pw, plane = Hp * Wp;
  const int L = g.OH * g.OW, L4 = L & ~3, Lp = (L + 3) & ~3;
  const int K = g.Cin * g.KH * g.KW, KK = g.KH * g.KW;
  const int o0 = blockIdx.y * CWM_OC;
  const int OC = (g.Cout - o0) < CWM_OC ? (g.Cout - o0) : CWM_OC;
  constexpr int OCs = 16 * RT + 1;
  float *sX = cwm_smem;                                  // [Cin][Hp][Wp], zero border
  float *sM = cwm_smem + ((g.Cin * plane + 3) & ~3);     // [Lp][OCs]; afterwards: the partial tiles [S][16 RT][K]
  const int64_t row = blockIdx.x, n = row % N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = blockDim.x, NW = NT >> 6;
  {
    // staging in batches: CWM_SB loads of a thread in flight before their LDS stores (a load-store pair per trip costs a
    // global round trip each: 34 of them per wave were 70 % of the kernel)
    constexpr int CWM_SB = 12;
    const float *xn = x + n * (int64_t)g.Cin * g.H * g.W;
    const int totx = g.Cin * plane;
    for (int base = 0; base < totx; base += NT * CWM_SB) {
      float v[CWM_SB];
#pragma unroll
      for (int u = 0; u < CWM_SB; ++u) {
        const int idx = base + u * NT + tid;
        const int ic = idx < totx ? idx : 0;
        const int rr = ic / Wp, wq = ic - rr * Wp;
        const int c = rr / Hp, ih = rr - c * Hp - g.ph, iw = wq - g.pw;
        const bool in = idx < totx && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
        v[u] = in ? xn[((int64_t)c * g.H + ih) * g.W + iw] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < CWM_SB; ++u) {
        const int idx = base + u * NT + tid;
        if (idx < totx) sX[idx] = v[u];
      }
    }
    const float *Mrow = M + (row * g.Cout + o0) * (int64_t)L;
    const int totm = 16 * RT * Lp;
    for (int base = 0; base < totm; base += NT * CWM_SB) {
      float v[CWM_SB];
#pragma unroll
      for (int u = 0; u < CWM_SB; ++u) {
        const int idx = base + u * NT + tid;
        const int ic = idx < totm ? idx : 0;
        const int o = ic / Lp, l = ic - o * Lp;                 // coalesced along the positions of a channel
        v[u] = (idx < totm && o < OC && l < L) ? Mrow[(int64_t)o * L + l] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < CWM_SB; ++u) {
        const int idx = base + u * NT + tid;
        if (idx < totm) {
          const int o = idx / Lp, l = idx - o * Lp;
          sM[l * OCs + o] = v[u];
        }
      }
    }
  }
  __syncthreads();
  const int j = lane & 15, kq = lane >> 4;
  const int nct = (K + 15) / 16;
  const int wrap = g.sh * Wp - g.OW * g.sw;
  const int nsteps = L4 / 4;   // whole steps; the last split also takes the 1..3 positions behind them
  cwm_f32x4 acc[RT];
  for (int item = wave; item < nct * S; item += NW) {
    const int ct = item / S, sp = item - ct * S;
    const int s0 = (int)((int64_t)sp * nsteps / S), s1 = (int)((int64_t)(sp + 1) * nsteps / S);
    const int kp = ct * 16 + j;
    const bool kvalid = kp < K;
    const int kpc = kvalid ? kp : 0;
    const int c = kpc / KK, r = kpc - c * KK, kh = r / g.KW, kw = r - kh * g.KW;
    const int lstart = 4 * s0 + kq;
    int oh = lstart / g.OW, ow = lstart - oh * g.OW;
    int pos = c * plane + (oh * g.sh + kh * g.dh) * Wp + ow * g.sw + kw * g.dw;
    const float *aM = sM + lstart * OCs + j;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (cwm_f32x4){0.f, 0.f, 0.f, 0.f};
    auto advance = [&]() __attribute__((always_inline)) {   // l += 4 (OW >= 4: at most one wrap)
      ow += 4;
      pos += 4 * g.sw;
      const bool wr = ow >= g.OW;
      ow = wr ? ow - g.OW : ow;
      pos = wr ? pos + wrap : pos;
    };
    int st = s0;
    for (; st + CWM_U <= s1; st += CWM_U) {
      float b[CWM_U], a[CWM_U][RT];
#pragma unroll
      for (int u = 0; u < CWM_U; ++u) {
        b[u] = sX[pos];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[u][rt] = aM[u * 4 * OCs + 16 * rt];
        advance();
      }
      __builtin_amdgcn_sched_barrier(0);   // (hipcc sinks the reads to their MFMAs otherwise: one exposed LDS latency per pair)
#pragma unroll
      for (int u = 0; u < CWM_U; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][rt], b[u], acc[rt], 0, 0, 0);
      aM += CWM_U * 4 * OCs;
    }
    for (; st < s1; ++st) {
      const float b = sX[pos];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aM[16 * rt], b, acc[rt], 0, 0, 0);
      aM += 4 * OCs;
      advance();
    }
    if (sp == S - 1 && L4 < L) {   // the last 1..3 positions: lanes beyond L contribute nothing (their M rows are zero; no stray read)
      const float b = (L4 + kq < L) ? sX[pos] : 0.f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aM[16 * rt], b, acc[rt], 0, 0, 0);
    }
    if (S == 1) {
      if (kvalid) {
        float *vp = V + (row * g.Cout + o0) * (int64_t)K + kp;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int o = 16 * rt + 4 * kq + e;
            if (o < OC) vp[(int64_t)o * K] = acc[rt][e];
          }
      }
    } else {
      // (S > 1 implies a single trip of this loop per wave: nct S <= NW -- the partials may overwrite M only after everybody's products)
      __syncthreads();
      if (kvalid) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int e = 0; e < 4; ++e) sM[((int64_t)sp * 16 * RT + 16 * rt + 4 * kq + e) * K + kp] = acc[rt][e];
      }
    }
  }
  if (S > 1) {
    if (wave >= nct * S) __syncthreads();   // waves without an item: the barrier of the others
    __syncthreads();
    float *vp = V + (row * g.Cout + o0) * (int64_t)K;
    for (int idx = tid; idx < OC * K; idx += NT) {
      float v = sM[idx];
      for (int q = 1; q < S; ++q) v += sM[(int64_t)q * 16 * RT * K + idx];
      vp[idx] = v;
    }
  }
}

// LDS bytes of the matrix-pipe kernel for a geometry (0: it does not apply) and the number of position splits
static size_t conv2d_weight_mjp_mfma_lds(const Conv2dGeom &g, int *splits, int *threads) {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("VIVIT_CONV_MFMA");
    on = e ? atoi(e) : 1;
  }
  *splits = 1;
  *threads = 256;
  if (!on || g.OW < 4) return 0;
  const int64_t plane = (int64_t)(g.H + 2 * g.ph) * (g.W + 2 * g.pw);
  const int64_t L = (int64_t)g.OH * g.OW, K = (int64_t)g.Cin * g.KH * g.KW;
  const int rt = (int)((((g.Cout < CWM_OC ? g.Cout : CWM_OC) + 15) / 16));
  const int64_t mfloats = ((L + 3) & ~3LL) * (16 * rt + 1);
  const int64_t floats = ((g.Cin * plane + 3) & ~3LL) + mfloats;
  if (floats * 4 > 156 * 1024 || L * K < 4096) return 0;   // (tiny problems: the scalar kernel's launch is cheaper)
  // fewer column tiles than waves: split the positions (at most one item per wave, partial tiles inside M's region,
  // every split at least CWM_U steps)
  // one workgroup per CU (LDS-bound): 16 waves; two to three: 8; more: 4
  *threads = floats * 4 > 80 * 1024 ? 1024 : (floats * 4 > 40 * 1024 ? 512 : 256);
  const int64_t nct = (K + 15) / 16, nw = *threads / 64;
  int S = (int)(nw / nct);
  while (S > 1 && (S * 16 * rt * K > mfloats || (L / 4) / S < CWM_U)) --S;
  *splits = S < 1 ? 1 : S;
  return (size_t)floats * 4;
}

} // namespace vivit

using namespace vivit;

extern "C" {

int vivit_linear_weight_mjp_f32(const float *s, const float *z, float *V, int64_t C, int64_t N, int64_t O, int64_t I,
                                void *stream) {
  if (C < 0 || N < 0 || O < 0 || I < 0) return VIVIT_E_BADARG;
  if (C == 0 || N == 0 || O == 0 || I == 0) return VIVIT_OK;
  if (!s || !z || !V) return VIVIT_E_BADARG;
  const int64_t rows = C * N;
  const bool vec = (I % 4 == 0) && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(V)) & 15) == 0;
  const int64_t work = rows * O * I / (vec ? 4 : 1);
  int64_t grid = cdiv(work, 256);
  if (grid > 65536) grid = 65536;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (vec)
    linear_weight_mjp_kernel<true><<<(unsigned)grid, 256, 0, st>>>(s, z, V, rows, N, O, I);
  else
    linear_weight_mjp_kernel<false><<<(unsigned)grid, 256, 0, st>>>(s, z, V, rows, N, O, I);
  return launch_status();
}

int vivit_conv2d_weight_mjp_f32(const float *M, const float *x, float *V, int64_t rows, int64_t N, int64_t Cin, int64_t H,
                                int64_t W, int64_t Cout, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int64_t sh,
                                int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw, void *stream) {
  if (rows < 0 || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || OH <= 0 || OW <= 0 ||
      sh <= 0 || sw <= 0 || ph < 0 || pw < 0 || dh <= 0 || dw <= 0)
    return VIVIT_E_BADARG;
  if (rows == 0) return VIVIT_OK;
  if (!M || !x || !V) return VIVIT_E_BADARG;
  // the geometry must be that of a convolution: every output position reads inside the padded input
  if ((OH - 1) * sh + (KH - 1) * dh - ph >= H + ph || (OW - 1) * sw + (KW - 1) * dw - pw >= W + pw) return VIVIT_E_BADARG;
  if (rows > 0x7fffffffLL || Cout * Cin * KH * KW > 0x7fffffffLL / 4 || cdiv(Cout, CV_OT) > 65535 || cdiv(Cin * KH * KW, 256) > 65535)
    return VIVIT_E_UNSUPPORTED;
  Conv2dGeom g{(int)Cin, (int)H, (int)W, (int)Cout, (int)KH, (int)KW, (int)OH, (int)OW,
               (int)sh,  (int)sw, (int)ph, (int)pw,  (int)dh, (int)dw};
  const int64_t outs = Cout * Cin * KH * KW;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // grid.x chunks of whole batches (a chunk starts at a multiple of N rows, so row % N inside the kernel is the sample index)
  const int64_t per = (1 << 20) / N > 0 ? (1 << 20) / N : 1;
  const int64_t step = N * per;
  if (step > 0x7fffffffLL) return VIVIT_E_UNSUPPORTED;
  int splits = 1, threads = 256;
  const size_t lds = conv2d_weight_mjp_mfma_lds(g, &splits, &threads);
  if (lds > 0) {
    static unsigned long long attr_done = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return VIVIT_E_LAUNCH;
    if (!(attr_done & (1ull << (dev & 63)))) {
      const void *fns[4] = {reinterpret_cast<const void *>(conv2d_weight_mjp_mfma_kernel<1>), reinterpret_cast<const void *>(conv2d_weight_mjp_mfma_kernel<2>),
                            reinterpret_cast<const void *>(conv2d_weight_mjp_mfma_kernel<3>), reinterpret_cast<const void *>(conv2d_weight_mjp_mfma_kernel<4>)};
      for (const void *f : fns)
        if (!ensure_dynamic_lds(f, 156 * 1024, attr_done)) return VIVIT_E_LAUNCH;
      attr_done |= 1ull << (dev & 63);
    }
  }
  for (int64_t r0 = 0; r0 < rows; r0 += step) {
    const int64_t rc = (rows - r0) < step ? rows - r0 : step;
    const float *Mc = M + r0 * Cout * OH * OW;
    float *Vc = V + r0 * outs;
    if (lds > 0) {
      const dim3 grid((unsigned)rc, (unsigned)cdiv(Cout, CWM_OC));
      const int rt = (int)cdiv(Cout < CWM_OC ? Cout : CWM_OC, 16);
      if (rt == 1) conv2d_weight_mjp_mfma_kernel<1><<<grid, threads, lds, st>>>(Mc, x, Vc, N, g, splits);
      else if (rt == 2) conv2d_weight_mjp_mfma_kernel<2><<<grid, threads, lds, st>>>(Mc, x, Vc, N, g, splits);
      else if (rt == 3) conv2d_weight_mjp_mfma_kernel<3><<<grid, threads, lds, st>>>(Mc, x, Vc, N, g, splits);
      else conv2d_weight_mjp_mfma_kernel<4><<<grid, threads, lds, st>>>(Mc, x, Vc, N, g, splits);
    } else {
      const dim3 grid((unsigned)rc, (unsigned)cdiv(Cin * KH * KW, 256), (unsigned)cdiv(Cout, CV_OT));
      conv2d_weight_mjp_kernel<<<grid, 256, 0, st>>>(Mc, x, Vc, N, g);
    }
  }
  return launch_status();
}

} // extern "C"
