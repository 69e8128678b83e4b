// Band -> tridiagonal reduction by bulge chasing (stage 2 of the two-stage tridiagonalisation).
//
// The symmetric band matrix (half bandwidth NB = 64) lives in a row-band layout with room for the
// bulge:  AB[i][j - i + 2*NB] = A[i][j]  for  i - 2*NB <= j <= i   (n x (2*NB+1) floats, 21 MB at
// n = 40 960: L2 / Infinity-Cache resident; inside the library rows have a stride of SB2ST_LDP = 132 floats).  Sweep s annihilates column s below the sub-diagonal
// with a Householder reflector on rows s+1 .. s+NB and chases the resulting bulge down the band:
// task (s, k) works on rows  R_k = [s+1+k*NB, s+1+(k+1)*NB):
//     (i)   k > 0: apply H(s,k-1) from the right to the block E = A[R_k, R_{k-1}]   (creates the bulge)
//     (ii)  new reflector H(s,k) from the first column of E (k = 0: from column s), E <- H E
//     (iii) D = A[R_k, R_k] <- H D H
// Task (s, k) only depends on (s, k-1) and (s-1, k+1), so all tasks with equal t = 2 s + k are
// independent: the host launches one kernel per wavefront step t (about 2 n launches of up to
// n / (2 NB - 1) workgroups; a dependent launch boundary costs less than an in-kernel grid barrier
// and cannot deadlock).  Each task is one 256-thread workgroup (four waves, each a quarter of the columns).
//
// The reflectors are kept for the back-transformation: v(s,k) at R2[s][R_k], tau at tau2[s][k].
#include <cstdlib>

#include "common.h"
#include "device_utils.h"
#include "eig_internal.h"

namespace vivit {

#ifndef SB2ST_VARIANT
#define SB2ST_VARIANT 0
#endif
constexpr int NB = 64;             // half bandwidth
constexpr int LDAB = SB2ST_LDP;    // band row stride inside the library: 2 NB + 1 entries + 3 floats of padding (eig_internal.h)
static_assert(LDAB >= 2 * NB + 1 + 3 && LDAB % 4 == 0, "a 16-byte store of a row's last entries needs three floats of padding");

// value of lane `l` (compile-time constant) as a wave-uniform scalar
__device__ __forceinline__ float rl(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

constexpr int LDT = NB + 4;  // LDS row stride: 16-byte aligned rows, conflict-free row-per-lane ds_read_b128

// value of lane `l` (wave-uniform, run-time) as a wave-uniform scalar
__device__ __forceinline__ float rlu(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// One task per 256-thread workgroup.  The four waves load the two 64 x 64 blocks from the band (one
// coalesced 256-byte segment per band row: row r of the band holds [E row r | lower D row r]
// contiguously; all 32 loads of a thread are in flight together) into LDS.  Thread (wave q, lane r) then
// owns columns 16 q .. 16 q + 15 of row r of E and of D in registers.  Every matrix-vector product is a
// 16-term partial sum per thread, completed across the four waves through LDS (two barriers: one for
// g = E pv, one for E^T v and D v together); the small vector work (reflector, tau, w) is done redundantly
// and bit-identically by every wave, so no broadcast barrier is needed.  (History: ten barrier-separated
// phases on LDS-resident blocks: 12 us per wavefront step; one wave with the blocks in registers and
// v_readlane broadcasts, no barrier: 11.5 us of which 6.7 us were that single wave's 512 FMAs + 450
// readlanes and 4.7 us the serialised loads - scripts/probe/run_variant_sb2st.py.)
struct Sb2stLds {
  float sE[NB * LDT];
  float sD[NB * LDT];
  float red[3][4][NB];
};

// COH (the persistent kernel): agent-scope (sc1) accesses -- loads that miss the L1 and see what other workgroups
// stored, stores that are written through.
template <bool COH>
__device__ __forceinline__ float band_ld(const float *p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COH>
__device__ __forceinline__ void band_st(float *p, float v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

struct Sb2stRows {
  float ev[NB / 4], dv[NB / 4];
};

// band rows r = wave, wave + 4, ...: every address is inside the band array whatever r, lane and k are, so the loads
// are unconditional (all in flight at once) and masked afterwards
template <bool COH>
__device__ __forceinline__ void sb2st_load(const float *__restrict__ AB, int c0, int L, int wave, int lane, Sb2stRows &t) {
#pragma unroll
  for (int rr = 0; rr < NB / 4; ++rr) {
    const int r = 4 * rr + wave;
    const float *row = AB + (int64_t)(c0 + (r < L ? r : 0)) * LDAB;
    t.ev[rr] = band_ld<COH>(row + (NB - r + lane));                                   // E[r][lane]
    t.dv[rr] = band_ld<COH>(row + (lane <= r ? 2 * NB - r + lane : 2 * NB));          // D[r][lane], lane <= r
  }
  __builtin_amdgcn_sched_barrier(0);   // keep the consumers behind ALL loads (hipcc hoists the first one otherwise)
}

// Blocks in registers -> LDS, the task's arithmetic, results back in LDS (rows of E and D as they go back to the band).
// Band rows rr0 <= rr < rr1 of this wave (r = 4 rr + wave) from registers into the LDS blocks (rows beyond L and, for the first task
// of a sweep, the whole E block are zero; D is completed to the full symmetric block).  The persistent kernel does this for all
// rows while it waits for the hand-over of the last one and repeats it for that row alone afterwards.
__device__ __forceinline__ void sb2st_stage(int k, int L, int wave, int lane, const Sb2stRows &t, Sb2stLds &lds, int rr0, int rr1) {
  float *sE = lds.sE, *sD = lds.sD;
#pragma unroll
  for (int rr = 0; rr < NB / 4; ++rr) {
    if (rr >= rr0 && rr < rr1) {
      const int r = 4 * rr + wave;
      const bool in = r < L;
      const float e1 = (in && k > 0) ? t.ev[rr] : 0.f, d1 = in ? t.dv[rr] : 0.f;
      sE[r * LDT + lane] = e1;
      if (lane <= r) {
        sD[r * LDT + lane] = d1;
        sD[lane * LDT + r] = d1;
      }
    }
  }
}

// The task's arithmetic on the staged blocks (sb2st_stage by every wave, then this), results back in LDS.
// x: column s of the band for k = 0.  (pv, ptau): reflector of task k - 1 in, of task k out.  Returns beta.
__device__ __forceinline__ float sb2st_core(int k, int L, int wave, int lane, float x, float &pv, float &ptau, Sb2stLds &lds) {
  float *sE = lds.sE, *sD = lds.sD;
  auto &red = lds.red;
  const int q0 = 16 * wave;
  __syncthreads();

  float er[16], d[16];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    const float4 a = *reinterpret_cast<const float4 *>(sE + lane * LDT + q0 + 4 * c4);
    er[4 * c4] = a.x; er[4 * c4 + 1] = a.y; er[4 * c4 + 2] = a.z; er[4 * c4 + 3] = a.w;
    const float4 b = *reinterpret_cast<const float4 *>(sD + lane * LDT + q0 + 4 * c4);
    d[4 * c4] = b.x; d[4 * c4 + 1] = b.y; d[4 * c4 + 2] = b.z; d[4 * c4 + 3] = b.w;
  }
  float g = 0.f;
  if (k > 0) {
    // (i) g = ptau E pv;  E <- E - g pv^T
    float pvq[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) pvq[j] = rlu(pv, q0 + j);
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) part += er[j] * pvq[j];
    red[0][wave][lane] = part;
    __syncthreads();
    g = ptau * ((red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]));
#pragma unroll
    for (int j = 0; j < 16; ++j) er[j] -= g * pvq[j];
    x = sE[lane * LDT] - g * rl(pv, 0);   // first column of the updated E
  }
  // (ii) Householder reflector from x (every wave, identically)
  const float ssq = wave_sum_dpp(lane >= 1 ? x * x : 0.f);
  const float alpha = rl(x, 0);
  float tau = 0.f, beta = alpha, scal = 0.f;
  if (ssq > 0.f) {
    beta = -copysignf(sqrt_nr(alpha * alpha + ssq), alpha);
    tau = (beta - alpha) * rcp_nr(beta);
    scal = rcp_nr(alpha - beta);
  }
  const float v = lane < L ? (lane == 0 ? 1.f : x * scal) : 0.f;
  float vq[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) vq[j] = rlu(v, q0 + j);
  if (k > 0) {
    // z = tau E'^T v with E' = E - g pv^T (E still unchanged in LDS): this wave's 16 rows, lane = column
    float zp = 0.f, gv = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      zp += vq[j] * sE[(q0 + j) * LDT + lane];
      gv += vq[j] * rlu(g, q0 + j);
    }
    red[1][wave][lane] = zp - pv * gv;
  }
  {
    // (iii) p = tau D v: this wave's 16 columns, lane = row
    float pp = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) pp += d[j] * vq[j];
    red[2][wave][lane] = pp;
  }
  __syncthreads();
  if (k > 0) {
    const float z = tau * ((red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]));
#pragma unroll
    for (int j = 0; j < 16; ++j) er[j] -= v * rlu(z, q0 + j);
    if (wave == 0 && lane < L) er[0] = (lane == 0) ? beta : 0.f;  // exact zeros below the new sub-band entry
  }
  // w = p - tau/2 (p.v) v;  D -= v w^T + w v^T
  const float p = tau * ((red[2][0][lane] + red[2][1][lane]) + (red[2][2][lane] + red[2][3][lane]));
  const float pdotv = wave_sum_dpp(p * v);
  const float w = p - 0.5f * tau * pdotv * v;
#pragma unroll
  for (int j = 0; j < 16; ++j) d[j] -= v * rlu(w, q0 + j) + w * vq[j];
  // results back to LDS (rows)
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    if (k > 0)
      *reinterpret_cast<float4 *>(sE + lane * LDT + q0 + 4 * c4) = make_float4(er[4 * c4], er[4 * c4 + 1], er[4 * c4 + 2], er[4 * c4 + 3]);
    *reinterpret_cast<float4 *>(sD + lane * LDT + q0 + 4 * c4) = make_float4(d[4 * c4], d[4 * c4 + 1], d[4 * c4 + 2], d[4 * c4 + 3]);
  }
  pv = v;
  ptau = tau;
  __syncthreads();
  return beta;
}

// band rows rr0 <= rr < rr1 of this wave (r = 4 rr + wave) from LDS back to the band: lower part of D, and E
// Sixteen bytes per lane, two band rows per instruction (lanes 0..31: row 4 (2 j) + wave, lanes 32..63: row 4 (2 j + 1) + wave): a
// row's results are one contiguous segment [E row r (NB floats) | lower part of D row r (r + 1 floats)] that ends at the row's
// diagonal entry, the last entry of the band row; lane h writes floats 4 h .. 4 h + 3 of it, and what the last store of a row
// writes beyond the diagonal entry lands in the row's three floats of padding.  (The addresses are 4-byte aligned only.)  A
// task's 128 single-float store instructions per workgroup kept the write-through queue full for 2.5 us (timing-only builds, all
// on 256 workgroups with nobody waiting: 298 ms per chase, 171 ms without the bulk stores, 235 ms with these: SB2ST_PVAR notes
// in DESIGN.md section 8); 32 sixteen-byte ones take half of that.
template <bool COH>
__device__ __forceinline__ void band_st4(float *p, float4 v) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v w = {v.x, v.y, v.z, v.w};
  if constexpr (COH) __asm__ volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
  else __asm__ volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(w) : "memory");
}
template <bool COH>
__device__ __forceinline__ void sb2st_store(float *__restrict__ AB, int c0, int L, int k, int wave, int lane, int rr0, int rr1,
                                            const Sb2stLds &lds) {
  const int half = lane >> 5, s4 = 4 * (lane & 31);
#pragma unroll
  for (int j = 0; j < NB / 8; ++j) {
    const int rr = 2 * j + half, r = 4 * rr + wave;
    if (rr >= rr0 && rr < rr1 && r < L && (s4 < NB ? k > 0 : s4 - NB <= r)) {
      const float *src = s4 < NB ? lds.sE + r * LDT + s4 : lds.sD + r * LDT + (s4 - NB);
      band_st4<COH>(AB + (int64_t)(c0 + r) * LDAB + (NB - r + s4), *reinterpret_cast<const float4 *>(src));
    }
  }
}

// one launch per wavefront step t = 2 s + k (the fallback: VIVIT_SB2ST_PERSIST=0, or n < 960)
__global__ __launch_bounds__(256) void sb2st_task_kernel(float *__restrict__ AB, int n, int t, int s_lo,
                                                         float *__restrict__ R2, int64_t ldr, float *__restrict__ tau2,
                                                         int nk, int rmod) {
  __shared__ __attribute__((aligned(16))) Sb2stLds lds;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = s_lo + blockIdx.x;
  const int k = t - 2 * s;
  const int c0 = s + 1 + k * NB;
  const int L = (n - c0) < NB ? (n - c0) : NB;
  if (k < 0 || L <= 0 || s > n - 3) return;
  float pv = 0.f, ptau = 0.f, x = 0.f;
  if (k > 0) {
    pv = R2[(int64_t)(s % rmod) * ldr + (c0 - NB + lane)];
    ptau = tau2[(int64_t)s * nk + (k - 1)];
  } else {
    x = lane < L ? AB[(int64_t)(c0 + lane) * LDAB + (2 * NB - 1 - lane)] : 0.f;  // column s of the band
  }
  Sb2stRows rows;
  sb2st_load<false>(AB, c0, L, wave, lane, rows);
  sb2st_stage(k, L, wave, lane, rows, lds, 0, NB / 4);
  const float beta = sb2st_core(k, L, wave, lane, x, pv, ptau, lds);
  if (wave == 0) {
    if (k == 0 && lane < L) AB[(int64_t)(c0 + lane) * LDAB + (2 * NB - 1 - lane)] = (lane == 0) ? beta : 0.f;
    if (lane < L) R2[(int64_t)(s % rmod) * ldr + c0 + lane] = pv;
    if (lane == 0) tau2[(int64_t)s * nk + k] = ptau;
  }
  sb2st_store<false>(AB, c0, L, k, wave, lane, 0, NB / 4, lds);
}

// ---- the whole chase as ONE persistent launch.  A workgroup takes a sweep and runs its tasks k = 0, 1, .. in order,
// the reflector staying in registers.  What a task waits for is split by band row: rows 0..62 of task (s, k) are rows
// 1..63 of (s - 1, k), only its last row is the first row of (s - 1, k + 1).  Sweep s - 1 therefore publishes two
// counters (one 32-bit word in the spare last column of tau2's row: A << 16 | B):
//   A = k + 1: the FIRST row of its task k is in memory (stored first, right behind the arithmetic);
//   B = k + 1: all rows of its task k are.
// Task (s, k) requests its 64 rows as soon as B >= k + 1 (true long ago: it is one level behind), polls A >= k + 2 while
// they fly, then fetches the last row again.  The dependent chain of a wavefront step is then: first row stored ->
// counter -> poll -> one row loaded -> arithmetic -> first row stored, with everything else off the path.
// Band traffic is agent-scope (sc1) both ways.  (Plain stores inside an XCD with a write-through only at the hand-over
// to the next XCD are NOT safe: bytes that no later sweep rewrites -- the zeros a sweep leaves in the first column of E
// -- keep their line dirty in that XCD's L2, and an sc1 load that hits a dirty line returns its stale remainder after
// other XCDs have rewritten it; and they were slower than write-through stores throughout: 607 against 493 ms.)
// Sweeps are dealt to the XCDs in blocks of 32 consecutive sweeps (a ticket counter per XCD, taken in order: a waiting
// sweep's predecessor is always running or done, whatever is resident), so most hand-overs stay inside one L2.
// Spins are bounded (2 s): the grid always drains.  The arrival of all workgroups (they learn the set of XCDs from each
// other) goes through the gate of device_utils.h:persist_arrive: an attempt that is not fully resident within 2 s aborts
// with nothing written and the retry queued behind it runs (sytrd_persist.hip has the protocol); a second abort or a stall
// later on poisons d with NaN AND raises PERSIST_TMO_SB2ST in the sticky failure word: the solve ends with
// info = VIVIT_INFO_PERSIST_TIMEOUT, a status of its own.
struct Sb2stCtl {
  int ticket[8];
  int arrive[2], state[2], mask[2];   // per attempt: arrival counter, gate state, XCDs present
  int dead, pad;
};

__global__ __launch_bounds__(256) void sb2st_persist_kernel(float *__restrict__ AB, int n, float *__restrict__ R2, int64_t ldr,
                                                            float *__restrict__ tau2, int nk, int rmod, Sb2stCtl *ctl, int attempt,
                                                            int fault) {
  if (attempt == 1 && __hip_atomic_load(&ctl->state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != PERSIST_ABORT) return;
  __shared__ __attribute__((aligned(16))) Sb2stLds lds;
  __shared__ int s_info[4];   // 0: sweep, 1: dead, 2: index of the XCD among those present, 3: number of XCDs
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto prog = [&](int s) { return reinterpret_cast<int *>(tau2 + (int64_t)s * nk + (nk - 1)); };
  // ---- who is where: every workgroup registers its XCD, all wait for all (once)
  if (tid == 0) {
    int xcc;
    __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15;
    __hip_atomic_fetch_or(&ctl->mask[attempt], 1 << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool go = persist_arrive(&ctl->arrive[attempt], &ctl->state[attempt], (int)gridDim.x,
                                   ((fault >> attempt) & 1) ? 0ull : PERSIST_TIMEOUT_TICKS);
    const int mask = __hip_atomic_load(&ctl->mask[attempt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_info[1] = go ? 0 : 2;
    s_info[2] = __builtin_popcount(mask & ((1 << xcc) - 1));
    s_info[3] = __builtin_popcount(mask);
    // aborted at the gate: nothing has been written; only the second abort fails the solve
    if (!go && attempt == 1) __hip_atomic_store(&ctl->dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_info[1] == 2) return;
  const int xi = s_info[2], nx = s_info[3];
  // thread 0: wait until counter `A` (hi = true) or `B` of sweep sp has reached `need`
  int seenA = 0, seenB = 0;
  auto wait_for = [&](int sp, bool hi, int need) {
    int &seen = hi ? seenA : seenB;
    if (seen >= need) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int spins = 0;
    while (true) {
      const int raw = __hip_atomic_load(prog(sp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      seenA = raw >> 16;
      seenB = raw & 0xffff;
      if (seen >= need) break;
      if ((++spins & 1023) == 0 && (__builtin_amdgcn_s_memrealtime() - t0 > PERSIST_TIMEOUT_TICKS ||
                                    __hip_atomic_load(&ctl->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        s_info[1] = 1;
        __hip_atomic_store(&ctl->dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  };
  while (!s_info[1]) {
    // ---- next sweep of this XCD: ticket i -> block (i / 32) * nx + xi, sweep 32 * block + i % 32
    if (tid == 0) {
      const int i = __hip_atomic_fetch_add(&ctl->ticket[xi], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_info[0] = 32 * ((i >> 5) * nx + xi) + (i & 31);
    }
    __syncthreads();
    const int s = s_info[0];
    if (s > n - 3) break;
    const int ks = (n - s - 2) / NB + 1;                   // tasks of this sweep:  s + 1 + k NB < n
    const int kprev = s > 0 ? (n - s - 1) / NB + 1 : 0;    // tasks of sweep s - 1 (>= ks)
    seenA = seenB = 0;
    float pv = 0.f, ptau = 0.f;
    for (int k = 0; k < ks; ++k) {
      const int c0 = s + 1 + k * NB;
      const int L = (n - c0) < NB ? (n - c0) : NB;
#if !(defined(SB2ST_PVAR) && SB2ST_PVAR == 2)   // (2: timing only, wrong results: nobody waits -- the throughput without the chain)
      if (tid == 0 && s > 0) wait_for(s - 1, false, k + 1);
#endif
      __syncthreads();   // (also: the previous task's LDS reads are done)
      if (s_info[1]) break;
      Sb2stRows rows;
      sb2st_load<true>(AB, c0, L, wave, lane, rows);
      // into LDS while the hand-over of the last row is awaited: off the chain of the wavefront step (the stale copy of that row goes
      // along and is replaced below)
      sb2st_stage(k, L, wave, lane, rows, lds, 0, NB / 4);
#if !(defined(SB2ST_PVAR) && SB2ST_PVAR == 2)
      if (tid == 0 && s > 0) wait_for(s - 1, true, k + 2 < kprev ? k + 2 : kprev);
#endif
      __syncthreads();
      if (s_info[1]) break;
      if (wave == 3 && L == NB) {   // the last row again, now that its owner has stored it
        const float *row = AB + (int64_t)(c0 + NB - 1) * LDAB;
        rows.ev[NB / 4 - 1] = band_ld<true>(row + (1 + lane));
        rows.dv[NB / 4 - 1] = band_ld<true>(row + (NB + 1 + lane));
        sb2st_stage(k, L, wave, lane, rows, lds, NB / 4 - 1, NB / 4);
      }
      float x = 0.f;
      if (k == 0) x = lane < L ? band_ld<true>(AB + (int64_t)(c0 + lane) * LDAB + (2 * NB - 1 - lane)) : 0.f;  // column s of the band
#if defined(SB2ST_PVAR) && SB2ST_PVAR == 3   // timing only (wrong results): counter A goes out BEFORE the arithmetic -- the step without it
      if (tid == 0) __hip_atomic_store(prog(s), ((k + 1) << 16) | k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
#if defined(SB2ST_PVAR) && SB2ST_PVAR == 1   // timing only (wrong results): no arithmetic, the hand-over chain alone
      float beta = x;
      lds.sE[tid] = rows.ev[0]; lds.sD[tid] = rows.dv[0];
      __syncthreads();
#else
      const float beta = sb2st_core(k, L, wave, lane, x, pv, ptau, lds);
#endif
      // ---- first row out, counter A
      if (wave == 0) {
        if (k == 0 && lane == 0) band_st<true>(AB + (int64_t)c0 * LDAB + (2 * NB - 1), beta);
        sb2st_store<true>(AB, c0, L, k, wave, lane, 0, 1, lds);
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(prog(s), ((k + 1) << 16) | k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the rest of column s (zeros), the reflector for the back-transformation
        if (k == 0 && lane >= 1 && lane < L) band_st<true>(AB + (int64_t)(c0 + lane) * LDAB + (2 * NB - 1 - lane), 0.f);
        if (lane < L) R2[(int64_t)(s % rmod) * ldr + c0 + lane] = pv;
        if (lane == 0) tau2[(int64_t)s * nk + k] = ptau;
        sb2st_store<true>(AB, c0, L, k, wave, lane, 1, NB / 4, lds);
      } else {
        sb2st_store<true>(AB, c0, L, k, wave, lane, 0, NB / 4, lds);
      }
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's band rows are in memory
      __syncthreads();
      if (tid == 0) __hip_atomic_store(prog(s), ((k + 1) << 16) | (k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(256) void sb2st_zero_kernel(float *__restrict__ tau2, int n, int nk) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n - 1) tau2[(int64_t)i * nk + (nk - 1)] = 0.f;                  // progress counters
  if (i < nk) tau2[(int64_t)(n - 1) * nk + i] = 0.f;                       // control block
}

__global__ void sb2st_poison_kernel(const Sb2stCtl *ctl, int n, float *__restrict__ d, int *__restrict__ tmo) {
  if (ctl->dead) {
    d[0] = __builtin_nanf("");
    __hip_atomic_fetch_or(tmo, PERSIST_TMO_SB2ST, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (n < 0) printf("sb2st persist: mask 0x%x 0x%x dead %d tickets %d %d %d %d %d %d %d %d\n", ctl->mask[0], ctl->mask[1], ctl->dead, ctl->ticket[0], ctl->ticket[1],
                    ctl->ticket[2], ctl->ticket[3], ctl->ticket[4], ctl->ticket[5], ctl->ticket[6], ctl->ticket[7]);
}

__global__ __launch_bounds__(256) void sb2st_extract_kernel(const float *__restrict__ AB, int n, float *__restrict__ d,
                                                            float *__restrict__ e) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  d[i] = AB[(int64_t)i * LDAB + 2 * NB];
  if (i + 1 < n) e[i] = AB[(int64_t)(i + 1) * LDAB + 2 * NB - 1];
  else e[i] = 0.f;
}

// the caller's band (rows of 2 NB + 1) in the library's row stride (public entry point only)
__global__ __launch_bounds__(256) void sb2st_pad_kernel(const float *__restrict__ AB, int64_t n, float *__restrict__ ABp) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * LDAB) return;
  const int64_t i = idx / LDAB;
  const int c = (int)(idx - i * LDAB);
  ABp[idx] = c <= 2 * NB ? AB[i * (2 * NB + 1) + c] : 0.f;
}

int sb2st_num_levels(int64_t n) { return (int)cdiv(n, NB) + 1; }
int64_t sb2st_ring_rows(int64_t n) { const int64_t need = n / NB + 64; return need < n ? need : n; }

// Reduce the band AB (destroyed) to tridiagonal (d, e).  R2: [r2rows][ldr] reflector storage, sweep s
// uses row s % r2rows (r2rows = n keeps every reflector for the back-transformation; a ring of
// SB2ST_RING rows is enough for the chase itself when only eigenvalues are wanted);
// tau2: [n][sb2st_num_levels(n)].
static bool sb2st_persist_enabled() {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("VIVIT_SB2ST_PERSIST");
    on = e ? atoi(e) : 1;
  }
  return on != 0 && persist_override() != 0;
}

int sb2st_launch(float *AB, int64_t n, float *d, float *e, float *R2, int64_t ldr, int64_t r2rows, float *tau2,
                 hipStream_t stream) {
  const int ni = (int)n;
  const int nk = sb2st_num_levels(n);
  // persistent form: the progress counters sit in tau2's last column (never a task: k < nk - 1), the control block in
  // its unused last row (sweeps end at n - 3)
  // (all 256 workgroups must be co-resident for the kernel's arrival barrier: one per CU of an unpartitioned 8-XCD part,
  // as for the other persistent kernels; a CPX partition or a CU-masked device takes the launch chain)
  const bool persist = sb2st_persist_enabled() && n >= 960 && (size_t)nk * sizeof(float) >= sizeof(Sb2stCtl) &&
                       device_cu_count() >= 256;
  int *tmo = persist ? persist_timeout_word(stream) : nullptr;
  if (n >= 3 && persist && tmo) {
    Sb2stCtl *ctl = reinterpret_cast<Sb2stCtl *>(tau2 + (n - 1) * nk);
    sb2st_zero_kernel<<<(unsigned)cdiv(n, 256), 256, 0, stream>>>(tau2, ni, nk);
    // two attempts: the second returns at once unless the first aborted at its arrival gate (nothing written by then)
    for (int attempt = 0; attempt < 2; ++attempt)
      sb2st_persist_kernel<<<256, 256, 0, stream>>>(AB, ni, R2, ldr, tau2, nk, (int)r2rows, ctl, attempt, persist_fault());
    sb2st_extract_kernel<<<(unsigned)cdiv(n, 256), 256, 0, stream>>>(AB, ni, d, e);
    sb2st_poison_kernel<<<1, 1, 0, stream>>>(ctl, getenv("VIVIT_SB2ST_DEBUG") ? -ni : ni, d, tmo);
    return launch_status();
  }
  if (n >= 3) {
    const int64_t tmax = 2 * (n - 3) + nk;
    for (int64_t t = 0; t <= tmax; ++t) {
      int64_t s_hi = t / 2;
      if (s_hi > n - 3) s_hi = n - 3;
      int64_t s_lo = 0;
      const int64_t num = t * NB + 1 - n;
      if (num >= 0) s_lo = num / (2 * NB - 1) + 1;
      if (s_lo > s_hi) continue;
      sb2st_task_kernel<<<(unsigned)(s_hi - s_lo + 1), 256, 0, stream>>>(AB, ni, (int)t, (int)s_lo, R2, ldr, tau2, nk, (int)r2rows);
    }
  }
  sb2st_extract_kernel<<<(unsigned)cdiv(n, 256), 256, 0, stream>>>(AB, ni, d, e);
  return launch_status();
}

} // namespace vivit

using namespace vivit;

extern "C" {

int vivit_sb2st_half_bandwidth(void) { return NB; }

size_t vivit_sb2st_f32_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  // tau2 + the band in the library's own row stride
  return align_up(sizeof(float) * (size_t)n * (size_t)sb2st_num_levels(n), 256) + sizeof(float) * (size_t)n * (size_t)LDAB + 512;
}

// AB: [n][2*NB+1] row-band layout (see top of file), destroyed.  d: [n], e: [n-1 (n allocated)].
// R2: [n][n] reflector rows (required).  workspace: tau2.
int vivit_sb2st_f32(float *AB, int64_t n, float *d, float *e, float *R2, void *workspace, size_t workspace_bytes,
                    void *stream) {
  if (n < 1 || !AB || !d || !e || !R2) return VIVIT_E_BADARG;
  if (!workspace || workspace_bytes < vivit_sb2st_f32_workspace_bytes(n)) return VIVIT_E_WORKSPACE;
  float *tau2 = reinterpret_cast<float *>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  float *ABp = reinterpret_cast<float *>(align_up(reinterpret_cast<uintptr_t>(tau2 + (size_t)n * (size_t)sb2st_num_levels(n)), 256));
  hipStream_t s = static_cast<hipStream_t>(stream);
  sb2st_pad_kernel<<<(unsigned)cdiv(n * LDAB, 256), 256, 0, s>>>(AB, n, ABp);
  return sb2st_launch(ABp, n, d, e, R2, n, n, tau2, s);
}

} // extern "C"
