// Factor materialisation kernels: the per-sample parameter Jacobian-transpose products of Linear and Conv2d weights,
// V_t[v, n, *param] = d(out_n . M[v, n]) / d(param)   (BackPACK's `param_mjp(..., sum_batch=False)`, called by the
// reference at vivit/extensions/secondorder/vivit/base.py:84-92; Linear: einsum("vno,ni->vnoi"), Conv2d: unfold +
// einsum("vnol,nkl->vnok")).  Both are bound by writing V (4 n P bytes): the Linear kernel is a pure store stream, the
// Conv2d kernel gathers the input patch values on the fly (no im2col buffer) from the L1/L2-resident sample.
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
  for (int64_t r0 = 0; r0 < rows; r0 += step) {
    const int64_t rc = (rows - r0) < step ? rows - r0 : step;
    const dim3 grid((unsigned)rc, (unsigned)cdiv(Cin * KH * KW, 256), (unsigned)cdiv(Cout, CV_OT));
    conv2d_weight_mjp_kernel<<<grid, 256, 0, st>>>(M + r0 * Cout * OH * OW, x, V + r0 * outs, N, g);
  }
  return launch_status();
}

} // extern "C"
