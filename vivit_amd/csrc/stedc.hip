// Eigen-decomposition of a symmetric tridiagonal matrix T = (d, e) on gfx950.
//
//   values only : Sturm-sequence bisection, one eigenvalue index per thread, fp64 recurrence.
//   with vectors: divide and conquer (Cuppen's rank-one tearing, Gu-Eisenstat/Loewner weights):
//       leaves (<= 64 rows) by wavefront-cooperative implicit-shift QL (device_utils.h),
//       merges bottom-up, ALL merges of a tree level batched into the same launches.
//
// Design points:
//   * the eigenvector matrix is kept TRANSPOSED (row = eigenvector): deflation rotations and the
//     gather of non-deflated vectors then touch contiguous rows, and the merge product
//     Qt_new = U^T * Qt_gathered is one MFMA GEMM per merge (batched through device-side
//     GemmDesc descriptors, because the non-deflated count k is only known on the device);
//   * no host synchronisation anywhere: sizes that depend on the data (k, number of rotations)
//     stay on the device, grids are sized for the worst case and exit early;
//   * O(k^2) scalar work (secular equation, Loewner products, eigenvector formation) runs in fp64
//     on the vector ALU; the O(n k^2) work runs in fp32 on MFMA;
//   * deterministic: rank-based sorting, sequential deflation scan, fixed-order reductions.
#include "common.h"
#include "device_utils.h"
#include "eig_internal.h"

namespace vivit {

constexpr int LEAF = 64;

__host__ __device__ inline int64_t node_bound(int64_t q, int level, int64_t n, int64_t nl) {
  return ((q << level) * n) / nl;  // first row of node q at `level` (level 0 = leaves)
}

struct DcWs {
  float *dcur, *dnew;   // [n] eigenvalues of the current / next level (physical row order)
  float *z;             // [n] rank-one vector
  float *ds, *zs;       // [n] sorted copies (modified by the deflation scan)
  float *dk, *zk;       // [n] compacted non-deflated poles / weights
  float *rot;           // [n][4]  (tp, tq) as ints in [0],[1]; c, s in [2],[3]
  float *rho, *tol;     // [nmerge]
  int *order;           // [n] sorted position -> local physical row
  int *ndpos, *dfpos;   // [n] sorted positions of the non-deflated / deflated poles, in output order
  int *kcount, *nrot;   // [nmerge]
  int *org;             // [n] origin pole of each secular root
  double *mu, *zhat;    // [n]
  float **rowptr;       // [n] destination row of each sorted position (after gather)
  GemmDesc *desc;       // [nmerge]
  float *Qt0, *Qt1;     // [n][n] eigenvectors (rows), block diagonal per node, ping-pong
  float *G;             // [n][n] gathered non-deflated rows
  float *U;             // [n * smax] secular eigenvectors, U_q[j * s + i]
};

// ------------------------------------------------------------------------------------------
// values only: bisection on the Sturm count (thread m finds the m-th smallest eigenvalue)
// ------------------------------------------------------------------------------------------
// Multisection: 8 lanes per eigenvalue evaluate 8 interior shifts of the current bracket in one sweep over the
// recurrence, so a round shrinks the bracket 9-fold and 12 rounds replace 38 bisection steps (the sweep is a
// serial O(n) chain per shift with an fp64 division per step: one thread per eigenvalue left 40 960 threads on
// 256 CUs and cost 0.4 s at n = 40 960).  The bracket ends far below fp32 resolution: span * 9^-12 = 3.5e-12 span.
constexpr int SB_LANES = 8;    // shifts per eigenvalue and round
constexpr int SB_ROUNDS = 12;
__global__ __launch_bounds__(256) void stebz_kernel(const float *__restrict__ d, const float *__restrict__ e, int n,
                                                    float *__restrict__ w, double *__restrict__ w64) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  // Gershgorin interval (every block recomputes it: O(n) reads, fixed order)
  float lo = 3.0e38f, hi = -3.0e38f;
  for (int i = tid; i < n; i += 256) {
    const float r = (i > 0 ? fabsf(e[i - 1]) : 0.f) + (i + 1 < n ? fabsf(e[i]) : 0.f);
    lo = fminf(lo, d[i] - r);
    hi = fmaxf(hi, d[i] + r);
  }
  const float gl = -block_max(-lo, red, tid);
  const float gu = block_max(hi, red, tid);
  const int sub = tid & (SB_LANES - 1);                       // which interior shift
  const int m = blockIdx.x * (256 / SB_LANES) + tid / SB_LANES;  // which eigenvalue (ascending)
  const bool active = m < n;
  const int mm = active ? m : n - 1;  // idle lanes shadow the last eigenvalue: the shuffles below stay convergent
  const double span = fmax((double)gu - (double)gl, 1e-300);
  double a = (double)gl - 1e-7 * span - 1e-300, b = (double)gu + 1e-7 * span + 1e-300;
  const double pivmin = 1e-290;
  // nine rounds leave a bracket of 2.6e-9 span, a twentieth of the fp32 spacing at the largest eigenvalue: enough for the fp32
  // result; the fp64 shifts of the inverse iteration (w64) take all twelve
  const int rounds = w64 ? SB_ROUNDS : SB_ROUNDS - 3;
  for (int it = 0; it < rounds; ++it) {
    const double h = (b - a) / (double)(SB_LANES + 1);
    const double x = a + h * (double)(sub + 1);
    // count eigenvalues < x
    int cnt = 0;
    double q = (double)d[0] - x;
    if (fabs(q) < pivmin) q = -pivmin;
    cnt += q < 0.0;
    for (int i = 1; i < n; ++i) {
      const double ee = (double)e[i - 1];
      // 1 / q from the hardware approximation + one Newton step (relative error ~1e-15: the count only needs the sign of q;
      // the IEEE division sequence was most of the sweep)
      double rq = __builtin_amdgcn_rcp(q);
      rq = fma(fma(-q, rq, 1.0), rq, rq);
      q = (double)d[i] - x - ee * ee * rq;
      // |q| is kept inside [pivmin, 1e300]: e^2 / pivmin overflows for |e| > 1e9 and rcp(inf) = 0 would turn the Newton
      // step (and every later count) into NaN, where the IEEE division recovered with e^2 / inf = 0
      const double aq = fabs(q);
      q = aq < pivmin ? -pivmin : (aq > 1e300 ? copysign(1e300, q) : q);   // (a NaN stays a NaN)
      cnt += q < 0.0;
    }
    // shifts are ascending in `sub`, counts non-decreasing: the eigenvalue lies between the last shift with
    // cnt <= m and the first with cnt > m.  nle = number of this group's shifts with cnt <= m.
    int nle = cnt <= mm ? 1 : 0;
#pragma unroll
    for (int off = 1; off < SB_LANES; off <<= 1) nle += __shfl_xor(nle, off, SB_LANES);
    const double na = a + h * (double)nle;          // nle = 0 keeps a
    const double nb = a + h * (double)(nle + 1);    // nle = 8 gives a + 9 h = b
    a = nle > 0 ? na : a;
    b = nle < SB_LANES ? nb : b;
    if (!(b > a)) break;
  }
  if (active && sub == 0) {
    w[m] = (float)(0.5 * (a + b));
    if (w64) w64[m] = 0.5 * (a + b);  // unrounded, unscaled: the shifts of the inverse iteration (stein.hip)
  }
}

// ------------------------------------------------------------------------------------------
// leaves: one wavefront per leaf, QL with the eigenvector rows in LDS
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dc_leaf_kernel(const float *__restrict__ d, const float *__restrict__ e, int n, int nl,
                                                     float *__restrict__ dcur, float *__restrict__ Qt, int64_t ldq,
                                                     int32_t *__restrict__ info) {
  __shared__ float Z[LEAF * (LEAF + 1)];
  __shared__ float dl[LEAF], el[LEAF];
  const int lane = threadIdx.x;
  const int64_t lo = node_bound(blockIdx.x, 0, n, nl), hi = node_bound(blockIdx.x + 1, 0, n, nl);
  const int s = (int)(hi - lo);
  if (lane < s) {
    float dv = d[lo + lane];
    if (lane == 0 && lo > 0) dv -= fabsf(e[lo - 1]);          // rank-one tearing at the cuts
    if (lane == s - 1 && hi < n) dv -= fabsf(e[hi - 1]);
    dl[lane] = dv;
    el[lane] = (lane < s - 1) ? e[lo + lane] : 0.f;
    for (int c = 0; c < s; ++c) Z[lane * (LEAF + 1) + c] = (c == lane) ? 1.f : 0.f;
  }
  // norm of the whole tridiagonal matrix (fixed order: every leaf gets the same value)
  float tn = 0.f;
  for (int i = lane; i < n; i += 64) tn = fmaxf(tn, fabsf(d[i]) + (i + 1 < n ? fabsf(e[i]) : 0.f) + (i > 0 ? fabsf(e[i - 1]) : 0.f));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tn = fmaxf(tn, __shfl_xor(tn, off, 64));
  __syncthreads();
  const int nfail = ql_implicit(dl, el, s, lane < s ? Z + lane * (LEAF + 1) : nullptr, tn);
  __syncthreads();
  // row c of Qt = eigenvector c = column c of Z
  for (int c = 0; c < s; ++c)
    if (lane < s) Qt[(lo + c) * ldq + lo + lane] = Z[lane * (LEAF + 1) + c];
  if (lane < s) dcur[lo + lane] = dl[lane];
  if (lane == 0 && nfail > 0) atomicAdd(info, nfail);
}

// ------------------------------------------------------------------------------------------
// merge step 1: z = (last row of Q1 | sign * first row of Q2), normalised; rho; tolerance
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dc_setup_kernel(const float *__restrict__ e, int n, int nl, int level,
                                                       const float *__restrict__ Qt, int64_t ldq,
                                                       const float *__restrict__ dcur, DcWs ws) {
  __shared__ float red[4];
  const int q = blockIdx.x, tid = threadIdx.x;
  const int64_t lo = node_bound(q, level, n, nl), hi = node_bound(q + 1, level, n, nl);
  const int64_t mid = node_bound(2 * q + 1, level - 1, n, nl);
  const float rho0 = e[mid - 1];
  const float sgn = rho0 < 0.f ? -1.f : 1.f;
  float ss = 0.f, dmax = 0.f;
  for (int64_t r = lo + tid; r < hi; r += 256) {
    // component (mid-1) of child-1 eigenvectors, component mid of child-2 eigenvectors
    const float zv = (r < mid) ? Qt[r * ldq + (mid - 1)] : sgn * Qt[r * ldq + mid];
    ws.z[r] = zv;
    ss += zv * zv;
    dmax = fmaxf(dmax, fabsf(dcur[r]));
  }
  ss = block_sum(ss, red, tid);
  dmax = block_max(dmax, red, tid);
  const float zn = sqrtf(ss);
  const float inv = zn > 0.f ? 1.f / zn : 0.f;
  float zmax = 0.f;
  for (int64_t r = lo + tid; r < hi; r += 256) {
    const float zv = ws.z[r] * inv;
    ws.z[r] = zv;
    zmax = fmaxf(zmax, fabsf(zv));
  }
  zmax = block_max(zmax, red, tid);
  if (tid == 0) {
    ws.rho[q] = fabsf(rho0) * ss;
    ws.tol[q] = 8.f * EPS32 * fmaxf(dmax, zmax);
  }
}

// merge step 2: stable rank sort of the node's eigenvalues
__global__ __launch_bounds__(256) void dc_rank_kernel(int n, int nl, int level, const float *__restrict__ dcur, DcWs ws) {
  const int q = blockIdx.y;
  const int64_t lo = node_bound(q, level, n, nl), hi = node_bound(q + 1, level, n, nl);
  const int s = (int)(hi - lo);
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= s) return;
  const float dv = dcur[lo + r];
  const unsigned kv = sort_key(dv);
  int rk = 0;
  for (int k = 0; k < s; ++k) {
    const unsigned kk = sort_key(dcur[lo + k]);
    rk += (kk < kv || (kk == kv && k < r)) ? 1 : 0;
  }
  ws.order[lo + rk] = r;
  ws.ds[lo + rk] = dv;
  ws.zs[lo + rk] = ws.z[lo + r];
}

// merge step 3: deflation scan (sequential over the sorted poles; one thread per merge), then the
// wave compacts the surviving poles and assigns every sorted row its destination:
//   non-deflated pole p  -> row lo+p of G (input of the merge GEMM)
//   deflated pole u      -> row lo+k+u of the next level's eigenvector matrix (final as is)
__global__ __launch_bounds__(64) void dc_deflate_kernel(int n, int nl, int level, DcWs ws, float *__restrict__ Gbuf,
                                                        float *__restrict__ Qnext, int64_t ldq) {
  const int q = blockIdx.x;
  const int64_t lo = node_bound(q, level, n, nl), hi = node_bound(q + 1, level, n, nl);
  const int s = (int)(hi - lo);
  __shared__ int sh_k, sh_nd;
  float *ds = ws.ds + lo, *zs = ws.zs + lo;
  int *ndpos = ws.ndpos + lo, *dfpos = ws.dfpos + lo;
  {
    // The scan is sequential by nature (a rotation changes the pole the next test compares with), but it need not pay a memory
    // round trip per element (round 6: thread 0 walking z and d in global memory took 8.7 ms for the top merge at n = 40 960,
    // 22 ms per solve).  The wave reads 64 elements at a time, coalesced, and walks them in registers: element t of the chunk
    // is broadcast from lane t (v_readlane), the state of the last non-deflated pole (index, z, d) is wave-uniform, the few
    // results go out as fire-and-forget stores from lane 0.  Same tests, same order of the output lists, same arithmetic.
    const int lane = threadIdx.x;
    const float rho = ws.rho[q], tol = ws.tol[q];
    float zmax = 0.f;
    for (int t = lane; t < s; t += 64) zmax = fmaxf(zmax, fabsf(zs[t]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, o, 64));
    int k = 0, ndefl = 0, nrot = 0;
    if (!(rho * zmax > tol)) {
      for (int t = lane; t < s; t += 64) dfpos[t] = t;  // rank-one term negligible: all deflate
      ndefl = s;
    } else {
      int prev = -1;
      float zprev = 0.f, dprev = 0.f;
      for (int base = 0; base < s; base += 64) {
        const int mine = base + lane;
        const float zl = mine < s ? zs[mine] : 0.f, dl = mine < s ? ds[mine] : 0.f;
        const int cnt = s - base < 64 ? s - base : 64;
        for (int u = 0; u < cnt; ++u) {
          const int t = base + u;
          float zt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zl), u));
          float dt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dl), u));
          if (rho * fabsf(zt) <= tol) {
            if (lane == 0) dfpos[ndefl] = t;
            ++ndefl;
            continue;
          }
          if (prev >= 0) {
            float sv = zprev, cv = zt;
            const float tau = sqrtf(cv * cv + sv * sv);
            const float tdiff = dt - dprev;
            cv /= tau;
            sv = -sv / tau;
            if (fabsf(tdiff * cv * sv) <= tol) {
              // close poles: rotate so that z[prev] vanishes, prev deflates
              const float dp = dprev * cv * cv + dt * sv * sv;
              const float dn = dprev * sv * sv + dt * cv * cv;
              if (lane == 0) {
                zs[t] = tau;
                zs[prev] = 0.f;
                float *rr = ws.rot + 4 * (lo + nrot);
                reinterpret_cast<int *>(rr)[0] = prev;
                reinterpret_cast<int *>(rr)[1] = t;
                rr[2] = cv;
                rr[3] = sv;
                ds[prev] = dp;
                ds[t] = dn;
                dfpos[ndefl] = prev;
              }
              zt = tau;
              dt = dn;
              ++nrot;
              ++ndefl;
              --k;  // prev was the last tentatively non-deflated pole
            }
          }
          if (lane == 0) ndpos[k] = t;
          ++k;
          prev = t;
          zprev = zt;
          dprev = dt;
        }
      }
    }
    if (lane == 0) {
      ws.kcount[q] = k;
      ws.nrot[q] = nrot;
      sh_k = k;
      sh_nd = ndefl;
    }
  }
  __syncthreads();
  const int k = sh_k, ndefl = sh_nd;
  for (int p = threadIdx.x; p < k; p += 64) {
    const int t = ndpos[p];
    ws.dk[lo + p] = ds[t];
    ws.zk[lo + p] = zs[t];
    ws.rowptr[lo + t] = Gbuf + (lo + p) * ldq + lo;
  }
  for (int u = threadIdx.x; u < ndefl; u += 64) {
    const int t = dfpos[u];
    ws.rowptr[lo + t] = Qnext + (lo + k + u) * ldq + lo;
    ws.dnew[lo + k + u] = ds[t];
  }
}

// merge step 4: copy every sorted row to its destination, zero-filling the other child's columns
__global__ __launch_bounds__(256) void dc_gather_kernel(int n, int nl, int level, const float *__restrict__ Qcur,
                                                        int64_t ldq, DcWs ws) {
  const int q = blockIdx.z;
  const int64_t lo = node_bound(q, level, n, nl), hi = node_bound(q + 1, level, n, nl);
  const int64_t mid = node_bound(2 * q + 1, level - 1, n, nl);
  const int s = (int)(hi - lo), n1 = (int)(mid - lo);
  const int t = blockIdx.y;
  if (t >= s) return;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= s) return;
  const int r = ws.order[lo + t];
  const bool mine = (r < n1) ? (c < n1) : (c >= n1);
  const float v = mine ? Qcur[(lo + r) * ldq + lo + c] : 0.f;
  ws.rowptr[lo + t][c] = v;
}

// merge step 5: deflation rotations on the destination rows (sequential per column)
__global__ __launch_bounds__(256) void dc_rotate_kernel(int n, int nl, int level, DcWs ws) {
  const int q = blockIdx.y;
  const int64_t lo = node_bound(q, level, n, nl), hi = node_bound(q + 1, level, n, nl);
  const int s = (int)(hi - lo);
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= s) return;
  const int nrot = ws.nrot[q];
  for (int r = 0; r < nrot; ++r) {
    const float *rr = ws.rot + 4 * (lo + r);
    const int tp = reinterpret_cast<const int *>(rr)[0], tq = reinterpret_cast<const int *>(rr)[1];
    const float cv = rr[2], sv = rr[3];
    float *pp = ws.rowptr[lo + tp], *pq = ws.rowptr[lo + tq];
    const float x = pp[c], y = pq[c];
    pp[c] = cv * x + sv * y;
    pq[c] = -sv * x + cv * y;
  }
}

// ------------------------------------------------------------------------------------------
// merge step 6: secular equation  1 + rho sum_j z_j^2 / (d_j - lambda) = 0, one root per WAVEFRONT: the 64 lanes
// split every pole sum (one root per thread left a top-level merge of 2e4 roots with 2e4 threads on 256 CUs:
// 134 ms of 100 % latency); all lanes see bit-identical reduced values (xor butterfly, commutative adds), so the
// bracketing / Newton control flow stays wave-uniform.
// (fp64; bracket by geometric search from the nearer pole, then safeguarded Newton/bisection)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ inline void secular_eval(const float *__restrict__ dk, const float *__restrict__ zk, int k, double dorg,
                                    double rho, double x, int lane, double &g, double &gp) {
  double s = 0.0, sp = 0.0;
  for (int j = lane; j < k; j += 64) {
    const double zz = (double)zk[j];
    const double den = ((double)dk[j] - dorg) - x;
    const double t = zz / den;
    s += zz * t;
    sp += t * t;
  }
  s = wave_sum_f64(s);
  sp = wave_sum_f64(sp);
  g = 1.0 + rho * s;
  gp = rho * sp;
}

constexpr int SEC_ROOTS_PER_WG = 4;  // 256 threads = 4 wavefronts = 4 roots

__global__ __launch_bounds__(256) void dc_secular_kernel(int n, int nl, int level, DcWs ws) {
  const int q = blockIdx.y;
  const int64_t lo = node_bound(q, level, n, nl);
  const int k = ws.kcount[q];
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * SEC_ROOTS_PER_WG + (threadIdx.x >> 6);
  if (i >= k) return;
  const float *dk = ws.dk + lo, *zk = ws.zk + lo;
  const double rho = (double)ws.rho[q];
  double z2sum = 0.0;
  if (i == k - 1) {
    for (int j = lane; j < k; j += 64) z2sum += (double)zk[j] * (double)zk[j];
    z2sum = wave_sum_f64(z2sum);
  }
  const double lo_pole = (double)dk[i];
  const double hi_pole = (i + 1 < k) ? (double)dk[i + 1] : lo_pole + rho * z2sum;
  int o = i;
  double g, gp;
  if (i + 1 < k) {
    secular_eval(dk, zk, k, 0.0, rho, 0.5 * (lo_pole + hi_pole), lane, g, gp);
    o = (g >= 0.0) ? i : i + 1;
  }
  const double dorg = (double)dk[o];
  double a, b;  // bracket in mu = lambda - dorg with g(a) <= 0 <= g(b)
  if (o == i) {
    const double top = hi_pole - dorg;
    double x = (i + 1 < k) ? 0.5 * top : top;
    secular_eval(dk, zk, k, dorg, rho, x, lane, g, gp);
    b = (i + 1 < k) ? top : x;
    bool found_hi = false;
    int guard = 0;
    while (g > 0.0 && x > 0.0 && guard++ < 1100) {
      b = x; found_hi = true;
      x *= 0.5;
      secular_eval(dk, zk, k, dorg, rho, x, lane, g, gp);
    }
    a = x;
    if (!found_hi && !(i + 1 < k)) b = top;
  } else {
    const double bot = lo_pole - dorg;  // < 0
    double x = 0.5 * bot;
    secular_eval(dk, zk, k, dorg, rho, x, lane, g, gp);
    a = bot;
    int guard = 0;
    while (g < 0.0 && x < 0.0 && guard++ < 1100) {
      a = x;
      x *= 0.5;
      secular_eval(dk, zk, k, dorg, rho, x, lane, g, gp);
    }
    b = x;
  }
  // safeguarded Newton inside [a, b]
  double x = 0.5 * (a + b);
  for (int it = 0; it < 100; ++it) {
    secular_eval(dk, zk, k, dorg, rho, x, lane, g, gp);
    if (g > 0.0) b = x; else a = x;
    double xn = x - g / gp;
    if (!(xn > a && xn < b)) xn = 0.5 * (a + b);
    if (xn == x || (b - a) <= 4.4e-16 * fmax(fabs(a), fabs(b))) { x = xn; break; }
    x = xn;
  }
  if (lane == 0) {
    ws.org[lo + i] = o;
    ws.mu[lo + i] = x;
    ws.dnew[lo + i] = (float)(dorg + x);
  }
}

// merge step 7: Loewner weights  zhat_j^2 = prod_i (lam_i - d_j) / (rho prod_{i != j} (d_i - d_j))
__global__ __launch_bounds__(256) void dc_zhat_kernel(int n, int nl, int level, DcWs ws) {
  const int q = blockIdx.y;
  const int64_t lo = node_bound(q, level, n, nl);
  const int k = ws.kcount[q];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= k) return;
  const float *dk = ws.dk + lo;
  const int *org = ws.org + lo;
  const double *mu = ws.mu + lo;
  const double dj = (double)dk[j];
  double prod = (((double)dk[org[j]] - dj) + mu[j]) / (double)ws.rho[q];
  for (int i = 0; i < k; ++i) {
    if (i == j) continue;
    const double num = ((double)dk[org[i]] - dj) + mu[i];
    const double den = (double)dk[i] - dj;
    prod *= num / den;
  }
  const double zj = (double)ws.zk[lo + j];
  ws.zhat[lo + j] = (zj < 0.0 ? -1.0 : 1.0) * sqrt(fabs(prod));
}

// merge step 8: U[j][i] = zhat_j / (d_j - lam_i), columns normalised (thread per column i)
__global__ __launch_bounds__(256) void dc_buildu_kernel(int n, int nl, int level, DcWs ws, int64_t smax) {
  const int q = blockIdx.y;
  const int64_t lo = node_bound(q, level, n, nl), hi = node_bound(q + 1, level, n, nl);
  const int64_t s = hi - lo;
  const int k = ws.kcount[q];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= k) return;
  const float *dk = ws.dk + lo;
  const double *zh = ws.zhat + lo;
  const double dorg = (double)dk[ws.org[lo + i]], mui = ws.mu[lo + i];
  double nrm = 0.0;
  for (int j = 0; j < k; ++j) {
    const double v = zh[j] / (((double)dk[j] - dorg) - mui);
    nrm += v * v;
  }
  const double inv = 1.0 / sqrt(nrm);
  float *U = ws.U + lo * smax;
  for (int j = 0; j < k; ++j) {
    const double v = zh[j] / (((double)dk[j] - dorg) - mui);
    U[(int64_t)j * s + i] = (float)(v * inv);
  }
}

// merge step 9: GEMM descriptors  Qt_next[lo : lo+k, lo : hi] = U^T (k x k) * G[lo : lo+k, lo : hi]
__global__ void dc_plan_kernel(int n, int nl, int level, DcWs ws, float *__restrict__ Gbuf, float *__restrict__ Qnext,
                               int64_t ldq, int64_t smax, int nmerge) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nmerge) return;
  const int64_t lo = node_bound(q, level, n, nl), hi = node_bound(q + 1, level, n, nl);
  GemmDesc ds;
  ds.A = ws.U + lo * smax;            // LAY_M: A[k_idx = j][row = i]
  ds.B = Gbuf + lo * ldq + lo;        // LAY_M: B[k_idx = j][col = c]
  ds.C = Qnext + lo * ldq + lo;
  ds.M = ws.kcount[q];
  ds.N = hi - lo;
  ds.K = ws.kcount[q];
  ds.lda = hi - lo;
  ds.ldb = ldq;
  ds.ldc = ldq;
  ws.desc[q] = ds;
}

// ------------------------------------------------------------------------------------------
// final ordering and output
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dc_final_rank_kernel(int n, const float *__restrict__ dcur, int *__restrict__ order,
                                                            float *__restrict__ w, const float *__restrict__ scal) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const float dv = dcur[r];
  const unsigned kv = sort_key(dv);
  int rk = 0;
  for (int k = 0; k < n; ++k) {
    const unsigned kk = sort_key(dcur[k]);
    rk += (kk < kv || (kk == kv && k < r)) ? 1 : 0;
  }
  order[rk] = r;
  const float sigma = scal ? scal[1] : 1.f;
  w[rk] = dv / sigma;
}

// Z[i][p] = Qt[order[p]][i]   (32x32 tiles through LDS, coalesced on both sides)
__global__ __launch_bounds__(256) void dc_transpose_out_kernel(int n, const float *__restrict__ Qt, int64_t ldq,
                                                               const int *__restrict__ order, float *__restrict__ Z,
                                                               int64_t ldz) {
  __shared__ float t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int p0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, i = i0 + tx;
    t[r][tx] = (p < n && i < n) ? Qt[(int64_t)order[p] * ldq + i] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, p = p0 + tx;
    if (i < n && p < n) Z[(int64_t)i * ldz + p] = t[tx][r];
  }
}

__global__ void scale_w_kernel(float *__restrict__ w, int n, const float *__restrict__ scal) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] /= scal[1];
}

// info = n for non-finite input; VIVIT_INFO_PERSIST_TIMEOUT when a persistent kernel of this solve gave up (its bit in
// the sticky word, which is cleared here) -- the two are different failures and the host treats them differently
__global__ void finalize_info_kernel(int32_t *info, int n, const float *__restrict__ scal, int *__restrict__ tmo) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (tmo && *tmo != 0) {
    *info = VIVIT_INFO_PERSIST_TIMEOUT;
    *tmo = 0;
  } else if (scal && scal[2] != 0.f) {
    *info = n;
  }
}

// ------------------------------------------------------------------------------------------
static int dc_num_leaves(int64_t n) {
  int64_t nl = 1;
  while (n > nl * LEAF) nl *= 2;
  return (int)nl;
}

size_t stedc_workspace_bytes(int64_t n, bool vectors) {
  if (!vectors) return 256;
  const int64_t nl = dc_num_leaves(n);
  size_t b = 0;
  b += align_up(sizeof(float) * n, 256) * 8;                // dcur dnew z ds zs dk zk + spare
  b += align_up(sizeof(float) * 4 * n, 256);                // rot
  b += align_up(sizeof(float) * nl, 256) * 2;               // rho tol
  b += align_up(sizeof(int) * n, 256) * 4;                  // order org ndpos dfpos
  b += align_up(sizeof(int) * nl, 256) * 2;                 // kcount nrot
  b += align_up(sizeof(double) * n, 256) * 2;               // mu zhat
  b += align_up(sizeof(float *) * n, 256);                  // rowptr
  b += align_up(sizeof(GemmDesc) * nl, 256);                // desc
  b += align_up(sizeof(float) * n * n, 256) * 4;            // Qt0 Qt1 G U
  return b + 1024;
}

static DcWs dc_carve(void *base, int64_t n) {
  const int64_t nl = dc_num_leaves(n);
  char *p = static_cast<char *>(base);
  p = reinterpret_cast<char *>(align_up(reinterpret_cast<uintptr_t>(p), 256));
  auto take = [&](size_t bytes) {
    char *r = p;
    p += align_up(bytes, 256);
    return r;
  };
  DcWs ws;
  ws.dcur = (float *)take(sizeof(float) * n);
  ws.dnew = (float *)take(sizeof(float) * n);
  ws.z = (float *)take(sizeof(float) * n);
  ws.ds = (float *)take(sizeof(float) * n);
  ws.zs = (float *)take(sizeof(float) * n);
  ws.dk = (float *)take(sizeof(float) * n);
  ws.zk = (float *)take(sizeof(float) * n);
  take(sizeof(float) * n);
  ws.rot = (float *)take(sizeof(float) * 4 * n);
  ws.rho = (float *)take(sizeof(float) * nl);
  ws.tol = (float *)take(sizeof(float) * nl);
  ws.order = (int *)take(sizeof(int) * n);
  ws.org = (int *)take(sizeof(int) * n);
  ws.ndpos = (int *)take(sizeof(int) * n);
  ws.dfpos = (int *)take(sizeof(int) * n);
  ws.kcount = (int *)take(sizeof(int) * nl);
  ws.nrot = (int *)take(sizeof(int) * nl);
  ws.mu = (double *)take(sizeof(double) * n);
  ws.zhat = (double *)take(sizeof(double) * n);
  ws.rowptr = (float **)take(sizeof(float *) * n);
  ws.desc = (GemmDesc *)take(sizeof(GemmDesc) * nl);
  ws.Qt0 = (float *)take(sizeof(float) * n * n);
  ws.Qt1 = (float *)take(sizeof(float) * n * n);
  ws.G = (float *)take(sizeof(float) * n * n);
  ws.U = (float *)take(sizeof(float) * n * n);
  return ws;
}

// Divide and conquer.  On return *Qt_out points at the (unsorted, row = eigenvector) matrix and
// *d_out at the matching eigenvalues, both inside the workspace.
int stedc_dc_launch(const float *d, const float *e, int64_t n, void *wsbase, float **Qt_out, float **d_out,
                    int **order_scratch, int32_t *info, hipStream_t stream) {
  const int nl = dc_num_leaves(n);
  DcWs ws = dc_carve(wsbase, n);
  const int ni = (int)n;
  const int64_t ldq = n;
  float *Qcur = ws.Qt0, *Qnxt = ws.Qt1;
  dc_leaf_kernel<<<nl, 64, 0, stream>>>(d, e, ni, nl, ws.dcur, Qcur, ldq, info);
  int level = 0;
  for (int nm = nl / 2; nm >= 1; nm /= 2) {
    ++level;
    int64_t smax = 0;
    for (int q = 0; q < nm; ++q) {
      const int64_t s = node_bound(q + 1, level, n, nl) - node_bound(q, level, n, nl);
      if (s > smax) smax = s;
    }
    const unsigned gx = (unsigned)cdiv(smax, 256);
    dc_setup_kernel<<<nm, 256, 0, stream>>>(e, ni, nl, level, Qcur, ldq, ws.dcur, ws);
    dc_rank_kernel<<<dim3(gx, nm), 256, 0, stream>>>(ni, nl, level, ws.dcur, ws);
    dc_deflate_kernel<<<nm, 64, 0, stream>>>(ni, nl, level, ws, ws.G, Qnxt, ldq);
    dc_gather_kernel<<<dim3(gx, (unsigned)smax, nm), 256, 0, stream>>>(ni, nl, level, Qcur, ldq, ws);
    dc_rotate_kernel<<<dim3(gx, nm), 256, 0, stream>>>(ni, nl, level, ws);
    dc_secular_kernel<<<dim3((unsigned)cdiv(smax, SEC_ROOTS_PER_WG), nm), 256, 0, stream>>>(ni, nl, level, ws);
    dc_zhat_kernel<<<dim3(gx, nm), 256, 0, stream>>>(ni, nl, level, ws);
    dc_buildu_kernel<<<dim3(gx, nm), 256, 0, stream>>>(ni, nl, level, ws, smax);
    dc_plan_kernel<<<(unsigned)cdiv(nm, 64), 64, 0, stream>>>(ni, nl, level, ws, ws.G, Qnxt, ldq, smax, nm);
    const int st = gemm_batched_launch(LAY_M, LAY_M, ws.desc, nm, smax, smax, 1.f, 0.f, stream);
    if (st != VIVIT_OK) return st;
    float *tq = Qcur; Qcur = Qnxt; Qnxt = tq;
    float *td = ws.dcur; ws.dcur = ws.dnew; ws.dnew = td;
  }
  *Qt_out = Qcur;
  *d_out = ws.dcur;
  *order_scratch = ws.order;
  return launch_status();
}

int stebz_launch(const float *d, const float *e, int64_t n, float *w, const float *scal, hipStream_t stream,
                 double *w64) {
  stebz_kernel<<<(unsigned)cdiv(n, 256 / SB_LANES), 256, 0, stream>>>(d, e, (int)n, w, w64);
  if (scal) scale_w_kernel<<<(unsigned)cdiv(n, 256), 256, 0, stream>>>(w, (int)n, scal);
  return launch_status();
}

int info_finalize_launch(int32_t *info, int64_t n, const float *scal, hipStream_t stream) {
  finalize_info_kernel<<<1, 64, 0, stream>>>(info, (int)n, scal, persist_timeout_word(stream));
  return launch_status();
}

int dc_output_launch(int64_t n, const float *dcur, const float *Qt, int64_t ldq, int *order, float *w, float *Z,
                     int64_t ldz, const float *scal, int32_t *info, hipStream_t stream) {
  dc_final_rank_kernel<<<(unsigned)cdiv(n, 256), 256, 0, stream>>>((int)n, dcur, order, w, scal);
  if (Z)
    dc_transpose_out_kernel<<<dim3((unsigned)cdiv(n, 32), (unsigned)cdiv(n, 32)), 256, 0, stream>>>((int)n, Qt, ldq, order,
                                                                                                  Z, ldz);
  if (scal) finalize_info_kernel<<<1, 64, 0, stream>>>(info, (int)n, scal, persist_timeout_word(stream));
  return launch_status();
}

// Zs[s][:] = Qt[order[r0 + s]][:]   (rows = eigenvectors r0 .. r0+rows-1 in ascending eigenvalue order)
__global__ __launch_bounds__(256) void dc_gather_rows_kernel(int n, const float *__restrict__ Qt, int64_t ldq,
                                                             const int *__restrict__ order, int r0, float *__restrict__ Zs,
                                                             int64_t ldz) {
  const int s = blockIdx.y;
  const float *src = Qt + (int64_t)order[r0 + s] * ldq;
  float *dst = Zs + (int64_t)s * ldz;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i];
}

// sort + scaled eigenvalues (all n) and the sorted eigenvector rows [r0, r1) of Qt copied to Zs
int dc_rows_launch(int64_t n, const float *dcur, const float *Qt, int64_t ldq, int *order, float *w, float *Zs,
                   int64_t ldz, int64_t r0, int64_t r1, const float *scal, hipStream_t stream) {
  dc_final_rank_kernel<<<(unsigned)cdiv(n, 256), 256, 0, stream>>>((int)n, dcur, order, w, scal);
  for (int64_t r = r0; r < r1; r += 65535) {  // grid.y is limited to 65535 rows per launch
    const int64_t rc = (r1 - r < 65535) ? r1 - r : 65535;
    dc_gather_rows_kernel<<<dim3((unsigned)(cdiv(n, 1024) < 64 ? cdiv(n, 1024) : 64), (unsigned)rc), 256, 0, stream>>>(
        (int)n, Qt, ldq, order, (int)r, Zs + (r - r0) * ldz, ldz);
  }
  return launch_status();
}

// Zs[s][:] = Qt[order[sel[s]]][:]   (arbitrary selection of eigenvectors, ascending eigenvalue positions sel[s])
__global__ __launch_bounds__(256) void dc_gather_sel_kernel(int n, const float *__restrict__ Qt, int64_t ldq,
                                                            const int *__restrict__ order, const int *__restrict__ sel,
                                                            int s0, float *__restrict__ Zs, int64_t ldz) {
  const int s = s0 + blockIdx.y;
  const float *src = Qt + (int64_t)order[sel[s]] * ldq;
  float *dst = Zs + (int64_t)s * ldz;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i];
}

int dc_select_launch(int64_t n, const float *dcur, const float *Qt, int64_t ldq, int *order, float *wscratch,
                     const int *sel, int64_t K, float *Zs, int64_t ldz, hipStream_t stream) {
  dc_final_rank_kernel<<<(unsigned)cdiv(n, 256), 256, 0, stream>>>((int)n, dcur, order, wscratch, nullptr);
  for (int64_t s = 0; s < K; s += 65535) {
    const int64_t sc = (K - s < 65535) ? K - s : 65535;
    dc_gather_sel_kernel<<<dim3((unsigned)(cdiv(n, 1024) < 64 ? cdiv(n, 1024) : 64), (unsigned)sc), 256, 0, stream>>>(
        (int)n, Qt, ldq, order, sel, (int)s, Zs, ldz);
  }
  return launch_status();
}

int info_scal_launch(int32_t *info, int64_t n, const float *scal, hipStream_t stream) {
  if (scal) finalize_info_kernel<<<1, 64, 0, stream>>>(info, (int)n, scal, persist_timeout_word(stream));
  return launch_status();
}

} // namespace vivit

using namespace vivit;

extern "C" {

size_t vivit_stedc_f32_workspace_bytes(int64_t n, int want_vectors) {
  if (n <= 0) return 0;
  return stedc_workspace_bytes(n, want_vectors != 0);
}

int vivit_stedc_f32(float *d, float *e, int64_t n, float *w, float *Z, int64_t ldz, void *workspace,
                    size_t workspace_bytes, int32_t *info, void *stream) {
  if (n < 0 || !info) return VIVIT_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // every argument check comes before the first enqueue (a refused call leaves *info untouched)
  if (n > 0 && (!d || !w || (n > 1 && !e) || (Z && ldz < n))) return VIVIT_E_BADARG;
  if (n > 0x7fffffffLL / 4) return VIVIT_E_UNSUPPORTED;
  if (n > 0 && Z && (!workspace || workspace_bytes < stedc_workspace_bytes(n, true))) return VIVIT_E_WORKSPACE;
  if (hipMemsetAsync(info, 0, sizeof(int32_t), s) != hipSuccess) return VIVIT_E_LAUNCH;
  if (n == 0) return VIVIT_OK;
  if (!Z) return stebz_launch(d, e, n, w, nullptr, s);
  float *Qt, *dd;
  int *order;
  int st = stedc_dc_launch(d, e, n, workspace, &Qt, &dd, &order, info, s);
  if (st != VIVIT_OK) return st;
  return dc_output_launch(n, dd, Qt, n, order, w, Z, ldz, nullptr, info, s);
}

} // extern "C"
