// Back-propagation of the sqrt-GGN factor through parameter-free layers and convolutions: the transposed
// input-Jacobian products  M [V, N, *out] -> [V, N, *in]  that BackPACK's derivative classes provide to the reference
// through `MatToJacMat` (vivit/extensions/secondorder/vivit/__init__.py:84-118: SqrtGGN{ReLU, Sigmoid, Tanh, ...},
// SqrtGGN{Max,Avg}Pool2d, base.py:19,41 for Conv2d / BatchNorm), the loss-Hessian square roots that seed the
// back-propagation (SqrtGGN{CrossEntropyLoss, MSELoss}, __init__.py:84-86) and the reductions of the parameter rules of
// biases and BatchNorm.  Everything here is bandwidth-bound elementwise / gather work (coalesced along the innermost
// spatial dimension, one pass over M), except the Conv2d rule which is a direct transposed convolution with the
// filter slice in LDS and 16 input channels per thread in registers.
#include <cstdlib>

#include "common.h"

namespace vivit {

// ---- elementwise activations: out[v, n, e] = M[v, n, e] * f'(x[n, e]) ----------------------------------------------
enum { ACT_RELU = 0, ACT_SIGMOID = 1, ACT_TANH = 2, ACT_LEAKY_RELU = 3, ACT_LOGSIGMOID = 4, ACT_ELU = 5, ACT_SELU = 6 };

__device__ __forceinline__ float act_derivative(int kind, float x, float a) {
  switch (kind) {
    case ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case ACT_SIGMOID: { const float s = 1.f / (1.f + expf(-x)); return s * (1.f - s); }
    case ACT_TANH: { const float t = tanhf(x); return 1.f - t * t; }
    case ACT_LEAKY_RELU: return x > 0.f ? 1.f : a;
    case ACT_LOGSIGMOID: return 1.f / (1.f + expf(x));
    case ACT_ELU: return x > 0.f ? 1.f : a * expf(x);
    default: {  // SELU
      const float scale = 1.0507009873554804934193349852946f, alpha = 1.6732632423543772848170429916717f;
      return x > 0.f ? scale : scale * alpha * expf(x);
    }
  }
}

__global__ __launch_bounds__(256) void act_jac_t_kernel(const float *__restrict__ M, const float *__restrict__ x,
                                                        float *__restrict__ out, int64_t total, int64_t per_v, int kind, float a) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    out[i] = M[i] * act_derivative(kind, x[i % per_v], a);
}

// ---- per-channel scale (BatchNorm in eval mode): out[r, c, l] = M[r, c, l] * scale[c] -------------------------------
__global__ __launch_bounds__(256) void channel_scale_kernel(const float *__restrict__ M, const float *__restrict__ scale,
                                                            float *__restrict__ out, int64_t total, int64_t C, int64_t L) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    out[i] = M[i] * scale[(i / L) % C];
}

// ---- pooling ---------------------------------------------------------------------------------------------------------
struct PoolGeom {
  int H, W, OH, OW, kh, kw, sh, sw, ph, pw;
};

// argmax of every pooling window (first maximum in row-major scan order, as torch's max_pool2d): idx[(n,c), oh, ow]
__global__ __launch_bounds__(256) void maxpool_argmax_kernel(const float *__restrict__ x, int *__restrict__ idx, int64_t planes,
                                                             PoolGeom g) {
  const int64_t total = planes * g.OH * g.OW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t pl = i / (g.OH * g.OW);
    const int o = (int)(i - pl * (g.OH * g.OW)), oh = o / g.OW, ow = o - oh * g.OW;
    const float *xp = x + pl * (int64_t)g.H * g.W;
    float best = -INFINITY;
    int bi = -1;
    for (int a = 0; a < g.kh; ++a) {
      const int h = oh * g.sh - g.ph + a;
      if (h < 0 || h >= g.H) continue;
      for (int b = 0; b < g.kw; ++b) {
        const int w = ow * g.sw - g.pw + b;
        if (w < 0 || w >= g.W) continue;
        const float v = xp[h * g.W + w];
        if (bi < 0 || v > best || (v != v && best == best)) {   // first maximum wins; a NaN wins and stays (torch)
          best = v;
          bi = h * g.W + w;
        }
      }
    }
    idx[i] = bi;
  }
}

// out[r, c, h, w] = sum over the windows that contain (h, w) of  M[r, c, oh, ow] * weight,
//   weight = [idx[n, c, oh, ow] == h W + w]  (max pooling)   or   1 / (kh kw)  (average pooling, count_include_pad)
template <bool MAX>
__global__ __launch_bounds__(256) void pool_jac_t_kernel(const float *__restrict__ M, const int *__restrict__ idx,
                                                         float *__restrict__ out, int64_t rows_planes, int64_t planes_per_v,
                                                         PoolGeom g, float inv_area) {
  const int64_t total = rows_planes * g.H * g.W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t pl = i / (g.H * g.W);
    const int hw = (int)(i - pl * (g.H * g.W)), h = hw / g.W, w = hw - h * g.W;
    const float *Mp = M + pl * (int64_t)g.OH * g.OW;
    const int *ip = MAX ? idx + (pl % planes_per_v) * (int64_t)g.OH * g.OW : nullptr;
    // windows containing h: oh sh - ph <= h < oh sh - ph + kh
    int oh_lo = (h + g.ph - g.kh + g.sh) / g.sh, oh_hi = (h + g.ph) / g.sh;
    int ow_lo = (w + g.pw - g.kw + g.sw) / g.sw, ow_hi = (w + g.pw) / g.sw;
    if (h + g.ph - g.kh + 1 <= 0) oh_lo = 0;
    if (w + g.pw - g.kw + 1 <= 0) ow_lo = 0;
    if (oh_hi > g.OH - 1) oh_hi = g.OH - 1;
    if (ow_hi > g.OW - 1) ow_hi = g.OW - 1;
    float acc = 0.f;
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const float m = Mp[oh * g.OW + ow];
        if (MAX) acc += ip[oh * g.OW + ow] == hw ? m : 0.f;
        else acc += m;
      }
    out[i] = MAX ? acc : acc * inv_area;
  }
}

// ---- Conv2d input rule (groups = 1, zero padding): transposed convolution ------------------------------------------
//   out[r, ci, h, w] = sum_{co, a, b} M[r, co, oh, ow] * Wt[co, ci, a, b],   oh sh - ph + a dh = h,  ow sw - pw + b dw = w
// grid.x: tiles of 256 input positions (h, w) of one row r = (v, n); grid.y: groups of CIT input channels; the filter
// slice [Cout][CIT][KH KW] sits in LDS (read as broadcasts), a thread keeps CIT accumulators and reads every M value it
// needs once (neighbouring threads share them through L1).
constexpr int CIT = 16;
struct ConvGeom {
  int Cin, H, W, Cout, KH, KW, OH, OW, sh, sw, ph, pw, dh, dw;
};
__global__ __launch_bounds__(256) void conv2d_jac_t_kernel(const float *__restrict__ M, const float *__restrict__ Wt,
                                                           float *__restrict__ out, ConvGeom g, int64_t rows) {
  extern __shared__ float sW[];  // [Cout][KH*KW][CIT]
  const int KK = g.KH * g.KW;
  const int ci0 = blockIdx.y * CIT;
  const int nci = (g.Cin - ci0) < CIT ? (g.Cin - ci0) : CIT;
  for (int i = threadIdx.x; i < g.Cout * KK * CIT; i += 256) {
    const int c = i % CIT, k = (i / CIT) % KK, co = i / (CIT * KK);
    sW[i] = c < nci ? Wt[((int64_t)co * g.Cin + ci0 + c) * KK + k] : 0.f;
  }
  __syncthreads();
  const int HW = g.H * g.W, L = g.OH * g.OW;
  // planes of at most 128 positions: 256 / HW rows share a workgroup (an 8 x 8 plane used a quarter of the lanes)
  int64_t r;
  int hw;
  if (HW <= 128) {
    const int rpw = 256 / HW, rl = threadIdx.x / HW;
    r = (int64_t)blockIdx.x * rpw + rl;
    hw = threadIdx.x - rl * HW;
    if (rl >= rpw || r >= rows) return;
  } else {
    const int tiles = (HW + 255) / 256;
    r = blockIdx.x / tiles;
    hw = (int)(blockIdx.x % tiles) * 256 + threadIdx.x;
    if (hw >= HW) return;
  }
  const int h = hw / g.W, w = hw - h * g.W;
  float acc[CIT];
#pragma unroll
  for (int c = 0; c < CIT; ++c) acc[c] = 0.f;
  const float *Mr = M + r * (int64_t)g.Cout * L;
  for (int a = 0; a < g.KH; ++a) {
    const int th = h + g.ph - a * g.dh;
    if (th < 0 || th % g.sh != 0) continue;
    const int oh = th / g.sh;
    if (oh >= g.OH) continue;
    for (int b = 0; b < g.KW; ++b) {
      const int tw = w + g.pw - b * g.dw;
      if (tw < 0 || tw % g.sw != 0) continue;
      const int ow = tw / g.sw;
      if (ow >= g.OW) continue;
      const float *mp = Mr + oh * g.OW + ow;
      const float *wp = sW + (a * g.KW + b) * CIT;
      for (int co = 0; co < g.Cout; ++co) {
        const float m = mp[(int64_t)co * L];
        const float4 *w4 = reinterpret_cast<const float4 *>(wp + co * KK * CIT);
#pragma unroll
        for (int q = 0; q < CIT / 4; ++q) {
          const float4 ww = w4[q];
          acc[4 * q] += m * ww.x; acc[4 * q + 1] += m * ww.y; acc[4 * q + 2] += m * ww.z; acc[4 * q + 3] += m * ww.w;
        }
      }
    }
  }
  float *op = out + (r * g.Cin + ci0) * (int64_t)HW + hw;
#pragma unroll
  for (int c = 0; c < CIT; ++c)
    if (c < nci) op[(int64_t)c * HW] = acc[c];
}

// ---- the same rule on the fp32 matrix pipe (round 4) -----------------------------------------------------------------------
// Per row r = (v, n) the rule is a GEMM  out[ci][hw] = sum_q W'[ci][q] G[q][hw],  q = (co, a, b),  W'[ci][q] = Wt[co][ci][a][b],
// G[q][hw] = M[co][(h + ph - a dh) / sh][(w + pw - b dw) / sw] (zero unless the divisions are exact and in range):
// 16..64 x 64..1024 x 144..576 in ResNet-32.  One workgroup per row and group of 32 input channels stages
//   * M "up-sampled": sMu[co][u][t] with M[co][oh][ow] at u = oh sh + Bh, t = ow sw + Bw and zeros elsewhere (Bh = (KH-1) dh - ph
//     rows of border), so that G[q][hw] = sMu[tab[q] + h Wu + w] for EVERY q and hw -- no test, no division in the loop;
//   * the weights transposed, sW[q][ci] (stride 16 RT + 1), and tab[q] = co plane + (ph - a dh + Bh) Wu + (pw - b dw + Bw);
// output channels in chunks of CC when the whole of it does not fit (later chunks add into out).  A wave owns 16-wide
// column tiles of out (16 positions hw) and runs v_mfma_f32_16x16x4_f32 over q; one gathered value feeds RT MFMAs.  With
// fewer column tiles than waves the q range is split S ways and the partial tiles are summed through LDS in a fixed order.
// (Stride-s layers multiply the inserted zeros: s^2 of the MFMAs are idle work; two of ResNet-32's thirty layers.)
constexpr int CJM_IC = 32;   // input channels per workgroup (RT <= 2 row tiles)
constexpr int CJM_U = 8;     // steps (of four q) whose LDS reads are in flight together
typedef float cjm_f32x4 __attribute__((ext_vector_type(4)));

template <int RT>
__global__ __launch_bounds__(1024) void conv2d_jac_t_mfma_kernel(const float *__restrict__ M, const float *__restrict__ Wt,
                                                                 float *__restrict__ out, ConvGeom g, int CC, int S) {
  extern __shared__ __attribute__((aligned(16))) float cjm_smem[];
  const int KK = g.KH * g.KW;
  const int Bh = (g.KH - 1) * g.dh > g.ph ? (g.KH - 1) * g.dh - g.ph : 0, Bw = (g.KW - 1) * g.dw > g.pw ? (g.KW - 1) * g.dw - g.pw : 0;
  const int Hu = g.H + g.ph + Bh, Wu = g.W + g.pw + Bw, plane = Hu * Wu;
  const int HW = g.H * g.W, L = g.OH * g.OW;
  const int ci0 = blockIdx.y * CJM_IC;
  const int IC = (g.Cin - ci0) < CJM_IC ? (g.Cin - ci0) : CJM_IC;
  constexpr int WS = 16 * RT + 1;
  const int Qmax = (CC * KK + 3) & ~3;
  float *sMu = cjm_smem;                                  // [CC][Hu][Wu]
  float *sW = cjm_smem + ((CC * plane + 3) & ~3);          // [Qmax][WS]; afterwards: the partial tiles [items][16 RT][16]
  int *sT = reinterpret_cast<int *>(sW + Qmax * WS);       // [Qmax]
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = blockDim.x, NW = NT >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int nct = (HW + 15) / 16;
  const float *Mr = M + row * (int64_t)g.Cout * L;
  float *outr = out + (row * g.Cin + ci0) * (int64_t)HW;
  constexpr int SB = 12;   // staging loads of a thread in flight together
  for (int cc0 = 0; cc0 < g.Cout; cc0 += CC) {
    const int ccn = (g.Cout - cc0) < CC ? (g.Cout - cc0) : CC;
    const int Qc = ccn * KK, Qp = (Qc + 3) & ~3;
    if (cc0 > 0) __syncthreads();   // the previous chunk's reads (and partial sums) are done
    {
      const int totm = ccn * plane;
      for (int base = 0; base < totm; base += NT * SB) {
        float v[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int idx = base + u * NT + tid;
          const int ic = idx < totm ? idx : 0;
          const int co = ic / plane, rem = ic - co * plane;
          const int uh = rem / Wu - Bh, uw = rem % Wu - Bw;
          const int oh = uh / g.sh, ow = uw / g.sw;
          const bool in = idx < totm && uh >= 0 && uw >= 0 && oh * g.sh == uh && ow * g.sw == uw && oh < g.OH && ow < g.OW;
          v[u] = in ? Mr[(int64_t)(cc0 + co) * L + oh * g.OW + ow] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int idx = base + u * NT + tid;
          if (idx < totm) sMu[idx] = v[u];
        }
      }
      // weights: global order (co, ci, k) -> sW[(co KK + k)][ci]
      const int totw = ccn * IC * KK;
      for (int base = 0; base < totw; base += NT * SB) {
        float v[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int idx = base + u * NT + tid;
          const int ic = idx < totw ? idx : 0;
          const int co = ic / (IC * KK), rem = ic - co * (IC * KK);
          v[u] = Wt[((int64_t)(cc0 + co) * g.Cin + ci0) * KK + rem];
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int idx = base + u * NT + tid;
          if (idx < totw) {
            const int co = idx / (IC * KK), rem = idx - co * (IC * KK);
            const int i = rem / KK, k = rem - i * KK;
            sW[(co * KK + k) * WS + i] = v[u];
          }
        }
      }
      // zero rows of the weights (channels beyond IC, q beyond Qc) and the offset table
      for (int idx = tid; idx < Qp * 16 * RT; idx += NT) {
        const int q = idx / (16 * RT), i = idx - q * 16 * RT;
        if (q >= Qc || i >= IC) sW[q * WS + i] = 0.f;
      }
      for (int q = tid; q < Qp; q += NT) {
        int t = 0;
        if (q < Qc) {
          const int co = q / KK, k = q - co * KK, a = k / g.KW, b = k - a * g.KW;
          t = co * plane + (g.ph - a * g.dh + Bh) * Wu + (g.pw - b * g.dw + Bw);
        }
        sT[q] = t;
      }
    }
    __syncthreads();
    const int nsteps = Qp / 4;
    const int nitems = nct * S;
    for (int it0 = 0; it0 < nitems; it0 += NW) {   // (all waves make the same trips: barriers inside when S > 1)
      const int item = it0 + wave;
      const bool has = item < nitems;
      const int ct = has ? item / S : 0, sp = has ? item - ct * S : 0;
      const int s0 = (int)((int64_t)sp * nsteps / S), s1 = has ? (int)((int64_t)(sp + 1) * nsteps / S) : s0;
      const int hw = ct * 16 + j;
      const bool hvalid = has && hw < HW;
      const int hwc = hw < HW ? hw : HW - 1;
      const int h = hwc / g.W, w = hwc - h * g.W;
      const float *gb = sMu + h * Wu + w;
      const float *aW = sW + (4 * s0 + kq) * WS + j;
      const int *tq = sT + 4 * s0 + kq;
      cjm_f32x4 acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = (cjm_f32x4){0.f, 0.f, 0.f, 0.f};
      int st = s0;
      for (; st + CJM_U <= s1; st += CJM_U) {
        int t[CJM_U];
        float b[CJM_U], a[CJM_U][RT];
#pragma unroll
        for (int u = 0; u < CJM_U; ++u) t[u] = tq[4 * u];
#pragma unroll
        for (int u = 0; u < CJM_U; ++u)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) a[u][rt] = aW[u * 4 * WS + 16 * rt];
#pragma unroll
        for (int u = 0; u < CJM_U; ++u) b[u] = gb[t[u]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CJM_U; ++u)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][rt], b[u], acc[rt], 0, 0, 0);
        aW += CJM_U * 4 * WS;
        tq += CJM_U * 4;
      }
      for (; st < s1; ++st) {
        const float b = gb[tq[0]];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aW[16 * rt], b, acc[rt], 0, 0, 0);
        aW += 4 * WS;
        tq += 4;
      }
      if (S == 1) {
        if (hvalid) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ci = 16 * rt + 4 * kq + e;
              if (ci < IC) {
                float *op = outr + (int64_t)ci * HW + hw;
                *op = cc0 == 0 ? acc[rt][e] : *op + acc[rt][e];
              }
            }
        }
      } else {
        // partial tiles through the weights' region (S > 1 is only planned when all items fit one trip and the region)
        __syncthreads();
        if (has) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) sW[(item * 16 * RT + 16 * rt + 4 * kq + e) * 16 + j] = acc[rt][e];
        }
        __syncthreads();
        for (int idx = tid; idx < nct * 16 * RT * 16; idx += NT) {
          const int ct2 = idx / (16 * RT * 16), rem = idx - ct2 * (16 * RT * 16);
          const int ci = rem >> 4, hw2 = ct2 * 16 + (rem & 15);
          if (ci < IC && hw2 < HW) {
            float v = sW[(ct2 * S) * 16 * RT * 16 + rem];
            for (int q = 1; q < S; ++q) v += sW[(ct2 * S + q) * 16 * RT * 16 + rem];
            float *op = outr + (int64_t)ci * HW + hw2;
            *op = cc0 == 0 ? v : *op + v;
          }
        }
      }
    }
  }
}

struct ConvJtPlan {
  size_t lds;
  int CC, S, threads, RT;
};
// the matrix-pipe kernel's plan for a geometry (lds = 0: it does not apply)
static ConvJtPlan conv2d_jac_t_mfma_plan(const ConvGeom &g) {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("VIVIT_CONV_MFMA");
    on = e ? atoi(e) : 1;
  }
  ConvJtPlan p{0, 0, 1, 256, 1};
  if (!on) return p;
  const int64_t KK = (int64_t)g.KH * g.KW;
  const int64_t Bh = (g.KH - 1) * g.dh > g.ph ? (g.KH - 1) * g.dh - g.ph : 0, Bw = (g.KW - 1) * g.dw > g.pw ? (g.KW - 1) * g.dw - g.pw : 0;
  const int64_t plane = (g.H + g.ph + Bh) * (g.W + g.pw + Bw), HW = (int64_t)g.H * g.W;
  if (HW * g.Cout * KK < 4096) return p;   // (tiny problems: the scalar kernel's launch is cheaper)
  p.RT = (g.Cin < CJM_IC ? g.Cin : CJM_IC) > 16 ? 2 : 1;
  const int64_t WS = 16 * p.RT + 1;
  auto floats = [&](int64_t cc) { const int64_t qp = (cc * KK + 3) & ~3LL; return ((cc * plane + 3) & ~3LL) + qp * WS + qp; };
  int64_t cc = g.Cout;
  while (cc > 1 && floats(cc) * 4 > 150 * 1024) cc = (cc + 1) / 2;
  if (floats(cc) * 4 > 150 * 1024) return p;
  p.CC = (int)cc;
  p.lds = (size_t)floats(cc) * 4;
  p.threads = p.lds > 80 * 1024 ? 1024 : (p.lds > 40 * 1024 ? 512 : 256);
  const int64_t nct = (HW + 15) / 16, nw = p.threads / 64, qp = (cc * KK + 3) & ~3LL;
  int S = (int)(nw / nct);
  while (S > 1 && (nct * S * 16 * p.RT * 16 > qp * WS || (qp / 4) / S < CJM_U)) --S;
  p.S = S < 1 ? 1 : S;
  return p;
}

// ---- reductions of the parameter rules --------------------------------------------------------------------------------
// out[r] = sum_l M[r, l] * (X ? X[(r % rows_x), l] : 1): bias of a convolution / BatchNorm (X = null), BatchNorm weight
// (X = normalised input, shared by the V slices).  One wave per row, fixed summation order.
__global__ __launch_bounds__(256) void row_dot_kernel(const float *__restrict__ M, const float *__restrict__ X,
                                                      float *__restrict__ out, int64_t rows, int64_t rows_x, int64_t L) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float *m = M + r * L;
  const float *xr = X ? X + (r % rows_x) * L : nullptr;
  float acc = 0.f;
  if ((L & 3) == 0 && ((reinterpret_cast<uintptr_t>(M) | (X ? reinterpret_cast<uintptr_t>(X) : 0)) & 15) == 0) {
    // 16-byte loads, two of them in flight per operand and trip (rows of 256 .. 1024 floats: 1 .. 2 trips)
    const float4 *m4 = reinterpret_cast<const float4 *>(m), *x4 = reinterpret_cast<const float4 *>(xr);
    const int64_t L4 = L >> 2;
    float a0 = 0.f, a1 = 0.f;
    int64_t l = lane;
    for (; l + 64 < L4; l += 128) {
      const float4 p = m4[l], q = m4[l + 64];
      if (xr) {
        const float4 u = x4[l], v = x4[l + 64];
        a0 += (p.x * u.x + p.y * u.y) + (p.z * u.z + p.w * u.w);
        a1 += (q.x * v.x + q.y * v.y) + (q.z * v.z + q.w * v.w);
      } else {
        a0 += (p.x + p.y) + (p.z + p.w);
        a1 += (q.x + q.y) + (q.z + q.w);
      }
    }
    if (l < L4) {
      const float4 p = m4[l];
      if (xr) {
        const float4 u = x4[l];
        a0 += (p.x * u.x + p.y * u.y) + (p.z * u.z + p.w * u.w);
      } else {
        a0 += (p.x + p.y) + (p.z + p.w);
      }
    }
    acc = a0 + a1;
  } else {
    for (int64_t l = lane; l < L; l += 64) acc += xr ? m[l] * xr[l] : m[l];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) out[r] = acc;
}

// BatchNorm in eval mode, ALL its rules in one pass over the incoming factor (vivit_bn_eval_rules_f32): per row r = (v, n, c)
//   mx[r] = sum_l M[r, l] X[r % rows_x, l]      (weight rule: sum_l M xhat = (mx - mean_c msum) rstd_c, finished by the caller)
//   msum[r] = sum_l M[r, l]                      (bias rule; second term of the weight rule)
//   out[r, l] = M[r, l] scale[c]                 (input rule)
// One wave per row; same load pattern and summation order as row_dot_kernel, so mx / msum are bit-identical to its results.
// wmean / wrstd (optional, [C]): mx is finished in place into the weight rule's value (mx - wmean_c msum) wrstd_c.
__global__ __launch_bounds__(256) void bn_eval_rules_kernel(const float *__restrict__ M, const float *__restrict__ X,
                                                            const float *__restrict__ scale, float *__restrict__ out,
                                                            float *__restrict__ mx, float *__restrict__ msum, int64_t rows,
                                                            int64_t rows_x, int64_t C, int64_t L,
                                                            const float *__restrict__ wmean, const float *__restrict__ wrstd) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float *m = M + r * L;
  const float *xr = X + (r % rows_x) * L;
  float *o = out ? out + r * L : nullptr;
  const float sc = scale ? scale[r % C] : 1.f;
  float accx = 0.f, accm = 0.f;
  if ((L & 3) == 0 && ((reinterpret_cast<uintptr_t>(M) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    const float4 *m4 = reinterpret_cast<const float4 *>(m), *x4 = reinterpret_cast<const float4 *>(xr);
    float4 *o4 = reinterpret_cast<float4 *>(o);
    const int64_t L4 = L >> 2;
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
    int64_t l = lane;
    for (; l + 64 < L4; l += 128) {
      const float4 p = m4[l], q = m4[l + 64];
      const float4 u = x4[l], v = x4[l + 64];
      a0 += (p.x * u.x + p.y * u.y) + (p.z * u.z + p.w * u.w);
      a1 += (q.x * v.x + q.y * v.y) + (q.z * v.z + q.w * v.w);
      b0 += (p.x + p.y) + (p.z + p.w);
      b1 += (q.x + q.y) + (q.z + q.w);
      if (o) {
        o4[l] = make_float4(p.x * sc, p.y * sc, p.z * sc, p.w * sc);
        o4[l + 64] = make_float4(q.x * sc, q.y * sc, q.z * sc, q.w * sc);
      }
    }
    if (l < L4) {
      const float4 p = m4[l];
      const float4 u = x4[l];
      a0 += (p.x * u.x + p.y * u.y) + (p.z * u.z + p.w * u.w);
      b0 += (p.x + p.y) + (p.z + p.w);
      if (o) o4[l] = make_float4(p.x * sc, p.y * sc, p.z * sc, p.w * sc);
    }
    accx = a0 + a1;
    accm = b0 + b1;
  } else {
    for (int64_t l = lane; l < L; l += 64) {
      const float v = m[l];
      accx += v * xr[l];
      accm += v;
      if (o) o[l] = v * sc;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    accx += __shfl_down(accx, off, 64);
    accm += __shfl_down(accm, off, 64);
  }
  if (lane == 0) {
    if (mx) mx[r] = wrstd ? (accx - wmean[r % C] * accm) * wrstd[r % C] : accx;
    if (msum) msum[r] = accm;
  }
}

// ---- loss-Hessian square roots ------------------------------------------------------------------------------------------
// Cross entropy: p = softmax(logits[n, :]).  exact: S[v, n, c] = sqrt(p_v) (delta_vc - p_c) * scale  (V = C slices);
// sampled:  S[m, n, c] = (p_c - onehot[m, n, c]) * scale.   One workgroup per sample.
__global__ __launch_bounds__(256) void ce_sqrt_hessian_kernel(const float *__restrict__ logits, const float *__restrict__ onehot,
                                                              float *__restrict__ S, int64_t N, int64_t C, int64_t Vd, float scale) {
  extern __shared__ float p[];  // [C]
  __shared__ float red[256];
  const int64_t n = blockIdx.x;
  const float *z = logits + n * C;
  float mx = -INFINITY;
  for (int64_t c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, z[c]);
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  mx = red[0];
  __syncthreads();
  float sum = 0.f;
  for (int64_t c = threadIdx.x; c < C; c += 256) {
    const float e = expf(z[c] - mx);
    p[c] = e;
    sum += e;
  }
  red[threadIdx.x] = sum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float inv = 1.f / red[0];
  for (int64_t c = threadIdx.x; c < C; c += 256) p[c] *= inv;
  __syncthreads();
  for (int64_t i = threadIdx.x; i < Vd * C; i += 256) {
    const int64_t v = i / C, c = i - v * C;
    float val;
    if (onehot) val = (p[c] - onehot[(v * N + n) * C + c]) * scale;
    else val = sqrtf(p[v]) * ((v == c ? 1.f : 0.f) - p[c]) * scale;
    S[(v * N + n) * C + c] = val;
  }
}

} // namespace vivit

using namespace vivit;

static inline unsigned grid_for(int64_t total) {
  int64_t g = cdiv(total, 256);
  return (unsigned)(g > 262144 ? 262144 : (g < 1 ? 1 : g));
}

extern "C" {

int vivit_act_jac_t_f32(const float *M, const float *x, float *out, int64_t Vd, int64_t per_v, int kind, float param,
                        void *stream) {
  if (Vd < 0 || per_v < 0 || kind < 0 || kind > ACT_SELU) return VIVIT_E_BADARG;
  if (Vd == 0 || per_v == 0) return VIVIT_OK;
  if (!M || !x || !out) return VIVIT_E_BADARG;
  act_jac_t_kernel<<<grid_for(Vd * per_v), 256, 0, static_cast<hipStream_t>(stream)>>>(M, x, out, Vd * per_v, per_v, kind, param);
  return launch_status();
}

int vivit_channel_scale_f32(const float *M, const float *scale, float *out, int64_t rows, int64_t C, int64_t L, void *stream) {
  if (rows < 0 || C <= 0 || L <= 0) return VIVIT_E_BADARG;
  if (rows == 0) return VIVIT_OK;
  if (!M || !scale || !out) return VIVIT_E_BADARG;
  channel_scale_kernel<<<grid_for(rows * C * L), 256, 0, static_cast<hipStream_t>(stream)>>>(M, scale, out, rows * C * L, C, L);
  return launch_status();
}

static int pool_geom(PoolGeom &g, int64_t H, int64_t W, int64_t OH, int64_t OW, int64_t kh, int64_t kw, int64_t sh, int64_t sw,
                     int64_t ph, int64_t pw) {
  if (H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0) return VIVIT_E_BADARG;
  if (H * W > 0x7fffffffLL || OH * OW > 0x7fffffffLL) return VIVIT_E_UNSUPPORTED;
  g = PoolGeom{(int)H, (int)W, (int)OH, (int)OW, (int)kh, (int)kw, (int)sh, (int)sw, (int)ph, (int)pw};
  return VIVIT_OK;
}

// idx_ws: [planes * OH * OW] ints (planes = N * C of the layer input)
int vivit_maxpool2d_jac_t_f32(const float *M, const float *x, float *out, int *idx_ws, int64_t Vd, int64_t planes, int64_t H,
                              int64_t W, int64_t OH, int64_t OW, int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph,
                              int64_t pw, void *stream) {
  PoolGeom g;
  const int st0 = pool_geom(g, H, W, OH, OW, kh, kw, sh, sw, ph, pw);
  if (st0 != VIVIT_OK || Vd < 0 || planes < 0) return st0 != VIVIT_OK ? st0 : VIVIT_E_BADARG;
  if (Vd == 0 || planes == 0) return VIVIT_OK;
  if (!M || !x || !out || !idx_ws) return VIVIT_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  maxpool_argmax_kernel<<<grid_for(planes * OH * OW), 256, 0, s>>>(x, idx_ws, planes, g);
  pool_jac_t_kernel<true><<<grid_for(Vd * planes * H * W), 256, 0, s>>>(M, idx_ws, out, Vd * planes, planes, g, 1.f);
  return launch_status();
}

int vivit_avgpool2d_jac_t_f32(const float *M, float *out, int64_t rows_planes, int64_t H, int64_t W, int64_t OH, int64_t OW,
                              int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, void *stream) {
  PoolGeom g;
  const int st0 = pool_geom(g, H, W, OH, OW, kh, kw, sh, sw, ph, pw);
  if (st0 != VIVIT_OK || rows_planes < 0) return st0 != VIVIT_OK ? st0 : VIVIT_E_BADARG;
  if (rows_planes == 0) return VIVIT_OK;
  if (!M || !out) return VIVIT_E_BADARG;
  pool_jac_t_kernel<false><<<grid_for(rows_planes * H * W), 256, 0, static_cast<hipStream_t>(stream)>>>(
      M, nullptr, out, rows_planes, 1, g, 1.f / (float)(kh * kw));
  return launch_status();
}

int vivit_conv2d_jac_t_f32(const float *M, const float *weight, float *out, int64_t rows, int64_t Cin, int64_t H, int64_t W,
                           int64_t Cout, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int64_t sh, int64_t sw, int64_t ph,
                           int64_t pw, int64_t dh, int64_t dw, void *stream) {
  if (rows < 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || OH <= 0 || OW <= 0 || sh <= 0 || sw <= 0 ||
      ph < 0 || pw < 0 || dh <= 0 || dw <= 0)
    return VIVIT_E_BADARG;
  if (rows == 0) return VIVIT_OK;
  if (!M || !weight || !out) return VIVIT_E_BADARG;
  const size_t lds = (size_t)Cout * KH * KW * CIT * sizeof(float);
  const int64_t tiles = cdiv(H * W, 256);
  ConvGeom g{(int)Cin, (int)H, (int)W, (int)Cout, (int)KH, (int)KW, (int)OH, (int)OW, (int)sh, (int)sw, (int)ph, (int)pw, (int)dh, (int)dw};
  // the geometry must be that of a convolution (as the weight rule checks): every output position reads inside the padded input
  const bool conv_geom = (OH - 1) * sh + (KH - 1) * dh - ph < H + ph && (OW - 1) * sw + (KW - 1) * dw - pw < W + pw;
  // The scalar kernel (40-47 TFLOP/s on ResNet-32's layers) takes every shape whose filter slice fits its LDS budget; the
  // matrix-pipe kernel takes the rest (Cout KH KW > 1024: measured 107 against 122 us on 32 -> 32 @ 16 x 16, but 165 against
  // 119 us on 16 -> 16 @ 32 x 32 and half the speed on stride-2 layers, whose inserted zeros it multiplies).
  const bool scalar_ok = lds <= 64 * 1024 && rows * tiles <= 0x7fffffffLL && cdiv(Cin, CIT) <= 65535;
  if (!scalar_ok && conv_geom && rows <= 0x7fffffffLL && cdiv(Cin, CJM_IC) <= 65535 && H * W <= (1 << 24) && Cout * KH * KW <= (1 << 24)) {
    const ConvJtPlan pl = conv2d_jac_t_mfma_plan(g);
    if (pl.lds > 0) {
      static unsigned long long attr_done = 0;
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess) return VIVIT_E_LAUNCH;
      if (!(attr_done & (1ull << (dev & 63)))) {
        if (!ensure_dynamic_lds(reinterpret_cast<const void *>(conv2d_jac_t_mfma_kernel<1>), 152 * 1024, attr_done) ||
            !ensure_dynamic_lds(reinterpret_cast<const void *>(conv2d_jac_t_mfma_kernel<2>), 152 * 1024, attr_done))
          return VIVIT_E_LAUNCH;
        attr_done |= 1ull << (dev & 63);
      }
      const dim3 grid((unsigned)rows, (unsigned)cdiv(Cin, CJM_IC));
      if (pl.RT == 1)
        conv2d_jac_t_mfma_kernel<1><<<grid, pl.threads, pl.lds, static_cast<hipStream_t>(stream)>>>(M, weight, out, g, pl.CC, pl.S);
      else
        conv2d_jac_t_mfma_kernel<2><<<grid, pl.threads, pl.lds, static_cast<hipStream_t>(stream)>>>(M, weight, out, g, pl.CC, pl.S);
      return launch_status();
    }
  }
  if (!scalar_ok) return VIVIT_E_UNSUPPORTED;
  const int64_t gx = H * W <= 128 ? cdiv(rows, 256 / (H * W)) : rows * tiles;
  conv2d_jac_t_kernel<<<dim3((unsigned)gx, (unsigned)cdiv(Cin, CIT)), 256, lds, static_cast<hipStream_t>(stream)>>>(M, weight, out, g, rows);
  return launch_status();
}

int vivit_row_dot_f32(const float *M, const float *X, float *out, int64_t rows, int64_t rows_x, int64_t L, void *stream) {
  if (rows < 0 || L < 0 || (X && rows_x <= 0)) return VIVIT_E_BADARG;
  if (rows == 0) return VIVIT_OK;
  if (!M || !out) return VIVIT_E_BADARG;
  row_dot_kernel<<<(unsigned)cdiv(rows, 4), 256, 0, static_cast<hipStream_t>(stream)>>>(M, X, out, rows, X ? rows_x : 1, L);
  return launch_status();
}

int vivit_bn_eval_rules_f32(const float *M, const float *X, const float *scale, float *out, float *mx, float *msum, int64_t rows,
                            int64_t rows_x, int64_t C, int64_t L, const float *wmean, const float *wrstd, void *stream) {
  if (rows < 0 || L < 0 || rows_x <= 0 || C <= 0) return VIVIT_E_BADARG;
  if (rows == 0) return VIVIT_OK;
  if (!M || !X || (out && !scale) || (!out && !mx && !msum) || ((wmean == nullptr) != (wrstd == nullptr))) return VIVIT_E_BADARG;
  if (rows % C != 0 && (out || wrstd)) return VIVIT_E_BADARG;   // rows = (v, n, c) with c fastest
  bn_eval_rules_kernel<<<(unsigned)cdiv(rows, 4), 256, 0, static_cast<hipStream_t>(stream)>>>(M, X, scale, out, mx, msum, rows, rows_x, C, L,
                                                                                             wmean, wrstd);
  return launch_status();
}

int vivit_ce_sqrt_hessian_f32(const float *logits, const float *onehot, float *S, int64_t N, int64_t C, int64_t Vd, float scale,
                              void *stream) {
  if (N < 0 || C <= 0 || Vd <= 0) return VIVIT_E_BADARG;
  if (N == 0) return VIVIT_OK;
  if (!logits || !S || (!onehot && Vd != C)) return VIVIT_E_BADARG;
  if ((size_t)C * sizeof(float) > 60 * 1024 || N > 0x7fffffffLL) return VIVIT_E_UNSUPPORTED;
  ce_sqrt_hessian_kernel<<<(unsigned)N, 256, (size_t)C * sizeof(float), static_cast<hipStream_t>(stream)>>>(logits, onehot, S, N, C, Vd,
                                                                                                              scale);
  return launch_status();
}

} // extern "C"
