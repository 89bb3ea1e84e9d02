// blocked_chol.hpp — blocked Cholesky with border rows, tiles resident in MFMA accumulators (fp64).
//
// Third generation of the two factorisations on the update path (profiles/r01 history: Householder
// TSQR 440 us x4 -> element-wise register elimination 64 us -> this, 34 us).  Both are "factor a
// symmetric k x k matrix A = L L^T and push nb border rows B through it (B <- B L^-T)":
//   compression  (REF: StateHelper::measurement_compress_inplace, PL/state/StateHelper.cpp:602-614)
//       A = equilibrated Gram matrix of [H r],  border = the (H^T r) row   ->  R = L^T, z
//   EKF update   (REF: StateHelper::EKFUpdate, StateHelper.cpp:94-173)
//       A = S = H P H^T + R,  border = [M ; res^T] ((n+1) x r)           ->  W^T = [M ; res^T] L^-T
//       (then K M^T = W^T W and dx = W^T y, ekf_dc_kernel)
//
// Work decomposition (16-column panels, one wave per 16-row strip):
//   * NT waves own the symmetric row strips: tiles (t, c <= t), each held TRANSPOSED as a
//     v_mfma_f64_16x16x4 accumulator (reg q of lane l = A[16t + (l&15)][16c + (l>>4) + 4q]).
//     Held that way, an accumulator register is bit-for-bit a valid MFMA B operand, and row j of a
//     tile is one k-slab of one register, so panel results feed the next product with no shuffles.
//   * one more wave owns a border strip (16 border rows); workgroup g takes strip g and
//     redundantly factors A, so border work spreads over CUs with no inter-workgroup sync.
//   panel p:  (a) the wave owning the diagonal tile eliminates its 16 pivots one MFMA each
//                 (diag_chain) and publishes every step through LDS;
//             (b) every strip below applies the steps to its own panel tile as they appear
//                 (strip_chain, spin on an LDS counter), ending with X^T = (tile L_d^-T)^T;
//                 symmetric strips publish X through LDS in per-lane order (the same lane reads it
//                 back as an A operand);
//             (c) one barrier, then the trailing update  tile(t,c)^T -= L(c,p) X_t^T  (4 MFMA / tile).
//   The pivot chain is the critical path: 16 dependent steps per panel.  Strip -> wave assignment
//   pairs heavy and light strips on the waves that share a SIMD (w, w+4: tools/ubench/simd_map.hip).
// Pivots <= tau mark a numerically dependent column: its column of L is zero (compression: the
// gauge directions an MSCKF Jacobian cannot observe; EKF: tau = 0, "not PSD").
#pragma once
#include <hip/hip_runtime.h>

#include "mfma_tile.hpp"
#include "wave_ops.hpp"

namespace plv {


__device__ __forceinline__ double readlane_f64(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Optional phase stamps (tools/ubench/bchol_time.hip defines PLV_BCHOL_TIMING): s_memtime per wave.
#ifdef PLV_BCHOL_TIMING
__device__ long long *g_bchol_stamps = nullptr;  // [waves][64]
#define BC_STAMP(id)                                                                            \
  do {                                                                                          \
    if (g_bchol_stamps && (threadIdx.x & 63) == 0 && blockIdx.x == 0)                           \
      g_bchol_stamps[(threadIdx.x >> 6) * 64 + (id)] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define BC_STAMP(id)
#endif

typedef double d2 __attribute__((ext_vector_type(2)));

// Volatile LDS accesses with the address space spelled out: through a generic pointer the compiler
// keeps volatile accesses as flat_load / flat_store (it does not infer address spaces for them).
#define PLV_LDS __attribute__((address_space(3)))
template <class T> __device__ __forceinline__ void lds_vstore(T *p, T v) { *(volatile PLV_LDS T *)(p) = v; }
template <class T> __device__ __forceinline__ T lds_vload(const T *p) { return *(const volatile PLV_LDS T *)(p); }

// PLV_BC_FLAGS (tools/ubench/bchol_time.hip): the barrier behind every panel replaced by per-strip ready flags.
// A strip's trailing update waits only for the panel tiles it multiplies with, the next diagonal tile only for its own strip, and a
// heavy strip may run a panel behind: the step buffer is then kept by panel parity like the panel tiles and the pivots, and two
// counters per panel say when a parity's buffers may be overwritten.
#ifdef PLV_BC_FLAGS
#define PLV_BC_TSP 2
#else
#define PLV_BC_TSP 1
#endif
#ifndef PLV_BC_NTL
#define PLV_BC_NTL 12
#endif
template <int NTL>
struct BcLdsT {
  double Lp[2][NTL][64][4];  // published panel tiles of the symmetric strips (per-lane order), by panel parity (NT <= NTL)
  double Ts[PLV_BC_TSP][16][64][2];  // step j of the running factorisation: {register dump holding row j, masked -1/pivot}
  double rs[2][16];        // the pivots l_jj^2 of the panel (<= 0: dead column), by panel parity; readers take 1 / sqrt themselves
  int step_flag;           // 16 * panel + steps published so far
  int rs_flag;             // panels whose rs[] is published
  int bad;
  int n_amb;               // pivots the factorisation could not tell from zero (diag_chain's `amb` band)
#ifdef PLV_BC_FLAGS
  int x_flag[NTL];         // strip t: panels whose tile X_t is published in Lp
  int cdone[NTL];          // panel p: followers that have finished its chain (they no longer read Ts / rs of that parity)
  int tdone[NTL];          // panel p: followers that have finished its trailing update (they no longer read Lp of that parity)
#endif
};
typedef BcLdsT<PLV_BC_NTL> BcLds;    // the update's factorisations (up to 192 columns); the gate inside the Jacobian launches uses BcLdsT<2>

#ifdef PLV_BC_FLAGS
// bounded spin on an LDS word (a flag that is never set must not hang the device: the factorisation is then marked bad)
template <class LDS>
__device__ __forceinline__ void bc_wait_ge(LDS &lds, const int *w, int want) {
  int spins = 0;
  while (__builtin_amdgcn_readfirstlane(lds_vload(w)) < want) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 22)) {
      lds.bad = 2;
      break;
    }
  }
}
#endif

// 1/x: v_rcp_f64 + two Newton steps (the sequence the compiler's IEEE division starts with, without
// the scale / fixup instructions that only matter outside the pivots' range).
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  return r;
}

__device__ __forceinline__ double rsqrt_nr(double x) {
  double r = __builtin_amdgcn_rsq(x);
  double e = fma(-x * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  e = fma(-x * r, r, 1.0);
  return fma(0.5 * r, e, r);
}

// ---- the sequential part: 16 pivots of the diagonal tile ---------------------------------------
// T is the symmetric 16x16 diagonal tile in MFMA accumulator layout.  Row j of an accumulator sits in
// the 16 lanes of k-slab (j&3) of register j>>2, which is where the A and B operands of
// v_mfma_f64_16x16x4 want it, so one elimination step is "scale the register by -1/pivot masked to
// that slab, one MFMA" (only one operand needs the mask).  Each step is published through LDS
// ({register, masked multiplier}, then a step counter; LDS executes a wave's instructions in order)
// and every strip below applies it to its own panel tile while this wave computes the next pivot
// (strip_chain).  The pivot chain is the critical path of the whole kernel; measured per step on
// MI355X (tools/ubench/diag_step.hip): 153 ticks for this form, +95 if the same wave also carries
// an identity border to get L_d^-1, 600 for a one-lane-per-row v_readlane formulation.
// amb > 0 (compression): a pivot 0 < |pv| < amb is counted in lds.n_amb — on a unit-diagonal Gram matrix it is the square of a
// relative singular value below sqrt(amb), which the Gram matrix carries with a relative error of eps / |pv| or worse: whether such a
// column lives or dies (tau) is decided by rounding.  Exactly zero pivots (columns no row touches) are not counted.
template <bool STORE_L, class LDS>
__device__ __forceinline__ void diag_chain(d4 T, LDS &lds, double tau, int p, d4 &cap, double amb = 0.0) {
  const int lane = threadIdx.x & 63;
  const int lq = lane >> 4;
  double mask01[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) mask01[q] = (lq == q) ? 1.0 : 0.0;
  bool dead_any = false;
  int n_amb = 0;
  // The pivot of step j + 1 does not wait for the MFMA of step j: element (j+1, j+1) after that step is
  // fma(T[j][j+1] * (-1/pivot_j), T[j][j+1], T[j+1][j+1]) — exactly the one product the MFMA adds into it — so it is formed from
  // two lanes read BEFORE the MFMA is issued, and its reciprocal (rcp + two Newton steps) is computed in the shadow of the MFMA's
  // latency.  The step chain is then "MFMA result -> two lane reads -> scale -> MFMA" (measured: 410 -> ~170 cycles per step).
  double pv = readlane_f64(T[0], 0);
  bool live = pv > tau;  // uniform
  double ninv = live ? -rcp_nr(pv) : 0.0;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const int kk = jj & 3, rq = jj >> 2;
    if (jj == 0) BC_STAMP(40);
    if (jj == 8) BC_STAMP(41);
    dead_any |= !live;
    n_amb += (amb > 0.0 && pv != 0.0 && fabs(pv) < amb) ? 1 : 0;
    const double nm = ninv * mask01[kk];
    const double trow = T[rq];
    lds_vstore(reinterpret_cast<d2 *>(&lds.Ts[p & (PLV_BC_TSP - 1)][jj][lane][0]), d2{trow, nm});
    lds_vstore(&lds.step_flag, 16 * p + jj + 1);
#ifndef PLV_BC_NO_SCHED
    __builtin_amdgcn_sched_barrier(0);  // publish now: the scheduler would sink all 16 stores below the chain
#endif
    double b = 0.0, c = 0.0;
    if (jj < 15) {
      b = readlane_f64(trow, 16 * kk + jj + 1);                                   // T[j][j+1]
      c = readlane_f64(T[(jj + 1) >> 2], 16 * ((jj + 1) & 3) + jj + 1);           // T[j+1][j+1] before this step
    }
    if (STORE_L) cap[rq] = (lq == kk) ? trow : cap[rq];
    T = __builtin_amdgcn_mfma_f64_16x16x4f64(trow * nm, trow, T, 0, 0, 0);
    // (in the shadow of the MFMA latency) the pivot for the strips — they take 1 / sqrt of it themselves at the end of the panel,
    // four per lane in parallel, instead of ten dependent operations per step on this wave — then the next pivot and its reciprocal
    lds_vstore(&lds.rs[p & 1][jj], live ? pv : 0.0);
    if (jj < 15) {
      pv = fma(b * ninv, b, c);
      live = pv > tau;
      ninv = live ? -rcp_nr(pv) : 0.0;
    }
  }
  BC_STAMP(42);
  lds_vstore(&lds.rs_flag, p + 1);
  if (dead_any && lane == 0) lds.bad = 1;
  if (n_amb && lane == 0) lds.n_amb += n_amb;  // (one diagonal wave per panel, panels in sequence: no race)
}

// ---- every strip below the diagonal tile follows the chain on its own (transposed) panel tile ----
// W[c][i] -= a_c * (W[j][i] / pivot_j): A operand = the published register (slab j&3 holds a), B
// operand = own register scaled by the published masked multiplier.  Row j is captured before it is
// eliminated; scaled by 1/l_jj at the end it is row j of X^T = (tile L_d^-T)^T.
template <class LDS>
__device__ __forceinline__ d4 strip_chain(d4 W, LDS &lds, int p) {
  const int lane = threadIdx.x & 63;
  const int lq = lane >> 4;
  d4 cap = {0, 0, 0, 0};
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const int kk = jj & 3, rq = jj >> 2;
    const int want = 16 * p + jj + 1;
    d2 d;
#ifndef PLV_BC_SLEEP
#define PLV_BC_SLEEP 1
#endif
    for (;;) {  // flag first, then the data: both loads are in flight together
      const int f = lds_vload(&lds.step_flag);
      d = lds_vload(reinterpret_cast<const d2 *>(&lds.Ts[p & (PLV_BC_TSP - 1)][jj][lane][0]));
      if (__builtin_amdgcn_readfirstlane(f) >= want) break;
      __builtin_amdgcn_s_sleep(PLV_BC_SLEEP);  // a spinning wave must not take issue slots and LDS cycles from the chain
    }
    // (polling only the 4-byte flag and reading the step's data once afterwards, or sleeping longer, changes nothing measurable:
    // tools/ubench/bchol_time.hip, 80.0 k ticks per factorisation either way)
    const double brow = W[rq];
    cap[rq] = (lq == kk) ? brow : cap[rq];
    W = __builtin_amdgcn_mfma_f64_16x16x4f64(d[0], brow * d[1], W, 0, 0, 0);
  }
  while (__builtin_amdgcn_readfirstlane(lds_vload(&lds.rs_flag)) < p + 1) __builtin_amdgcn_s_sleep(1);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double pvq = lds_vload(&lds.rs[p & 1][lq + 4 * q]);
    cap[q] *= pvq > 0.0 ? rsqrt_nr(pvq) : 0.0;
  }
  return cap;
}

// -DPLV_BC_FOLLOW_BARRIER (round 5; NOT the default): the same sixteen steps applied once the diagonal wave has
// published ALL of them (blocked_chol then puts a barrier in between): no polling, the step records come out of LDS eight at a time
// ahead of the MFMAs that use them.  Same operations in the same order as strip_chain: the same bits (fingerprint in
// tools/ubench/bchol_time.hip).  The premise was that the strips' polling slows the pivot chain down (340-390 cycles per step in the
// kernel against 170 in tools/ubench/diag_step.hip).  Measured with stamps at r = 104, n = 121 (profiles/r05/bchol_time.txt): the
// diagonal wave's chain takes 5.7-5.8 k cycles per panel with the strips asleep at the barrier and 5.8-6.6 k with them polling — the
// step's own dependent sequence (four lane reads, ~11 double-precision operations and a reciprocal behind an MFMA result: ~360
// cycles) is what it costs, not the company — and putting the strips' 2 k cycles behind it instead of beside it makes the kernel
// SLOWER: 33.3 us against 30.8 us per launch back to back.  Kept for the record and for the next attempt, which has to shorten the
// step itself.
template <class LDS>
__device__ __forceinline__ d4 strip_follow(d4 W, LDS &lds, int p) {
  const int lane = threadIdx.x & 63;
  const int lq = lane >> 4;
  d4 cap = {0, 0, 0, 0};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    d2 d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = lds_vload(reinterpret_cast<const d2 *>(&lds.Ts[p & (PLV_BC_TSP - 1)][8 * h + u][lane][0]));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int jj = 8 * h + u, kk = jj & 3, rq = jj >> 2;
      const double brow = W[rq];
      cap[rq] = (lq == kk) ? brow : cap[rq];
      W = __builtin_amdgcn_mfma_f64_16x16x4f64(d[u][0], brow * d[u][1], W, 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double pvq = lds_vload(&lds.rs[p & 1][lq + 4 * q]);
    cap[q] *= pvq > 0.0 ? rsqrt_nr(pvq) : 0.0;
  }
  return cap;
}

template <int NT, class Ops, class LDS>
__device__ __forceinline__ void blocked_chol(Ops &ops, LDS &lds, int k, int nb, double tau, int strip, double amb = 0.0) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // wave-uniform roles
  const int li = lane & 15, lq = lane >> 4;
  const int ntk = (k + 15) >> 4;
  const bool is_sym = wave < NT;
  // strip t has (t - p) trailing tiles at panel p and the border strip ntk - 1 - p; waves w, w+4, w+8 share
  // a SIMD (one matrix pipe), so heavy strips are paired with light ones
  constexpr int map7[8] = {6, 5, 4, 1, 0, 2, 3, 7};
  constexpr int map8[8] = {1, 7, 6, 5, 0, 2, 3, 4};
  const int t = NT == 7 ? map7[wave & 7] : (NT == 8 ? map8[wave & 7] : wave);
  const int bs = strip;  // one border strip per workgroup, carried by wave NT (further waves only keep the barriers)
  const bool active = is_sym ? (t < ntk) : (wave == NT && bs * 16 < nb);
#ifdef PLV_BC_FLAGS
  const int nbord = bs * 16 < nb ? 1 : 0;  // followers of panel p: the symmetric strips below it and the border strip
#endif
  // acc[j] holds tile (strip, p + j): the array is rotated after every panel, and a finished panel tile is stored
  // right away instead of being kept to the end.
  d4 acc[NT];
#pragma unroll
  for (int c = 0; c < NT; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = c * 16 + lq + 4 * q;
      double v = 0.0;
      if (active && c < ntk) {
        if (is_sym)
          v = c <= t ? ops.sym_raw(t * 16 + li, cc) : 0.0;
        else
          v = ops.border_raw(bs * 16 + li, cc);
      }
      acc[c][q] = v;
    }
#ifdef PLV_BC_FLAGS
  if (threadIdx.x < NT) lds.x_flag[threadIdx.x] = lds.cdone[threadIdx.x] = lds.tdone[threadIdx.x] = 0;
#endif
  ops.scales_ready();  // (compression: column scales into LDS + barrier; the raw loads above are already in flight)
#ifdef PLV_BC_FLAGS
  __syncthreads();
#endif
#pragma unroll
  for (int c = 0; c < NT; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = c * 16 + lq + 4 * q;
      if (active && c < ntk) {
        if (is_sym)
          acc[c][q] = c <= t ? ops.sym_fix(t * 16 + li, cc, acc[c][q]) : 0.0;
        else
          acc[c][q] = ops.border_fix(bs * 16 + li, cc, acc[c][q]);
      }
    }
  BC_STAMP(0);
  // The panel loop is unrolled: the same body as a real loop (rotating the accumulators makes it loop
  // invariant) gave wrong results for NT = 4 with hipcc 7.2 (dx off by 1e-3; correct for NT = 2, 7, 8) and was
  // not faster, so the suspected instruction-cache effect was not the limiter either.
#pragma unroll
  for (int p = 0; p < NT; ++p)
    if (p < ntk) {
    const bool below = active && (!is_sym || t > p);
    d4 x = {0, 0, 0, 0};
    BC_STAMP(1 + 5 * p);
#ifdef PLV_BC_PRIO
    // the wave on the critical path gets the issue slots and the matrix pipe of its SIMD first: the diagonal wave of this panel, and
    // (PLV_BC_PRIO >= 2) the strip that holds the next diagonal tile
    if (is_sym && t == p)
      __builtin_amdgcn_s_setprio(3);
    else if (PLV_BC_PRIO >= 2 && is_sym && t == p + 1)
      __builtin_amdgcn_s_setprio(2);
    else
      __builtin_amdgcn_s_setprio(0);
#endif
    if (is_sym && t == p) {
      d4 cap = {0, 0, 0, 0};
#ifdef PLV_BC_FLAGS
      if (p >= 2) bc_wait_ge(lds, &lds.cdone[p - 2], ntk - 1 - (p - 2) + nbord);  // the followers of panel p - 2 are out of this parity's step buffer and pivots
#endif
      diag_chain<Ops::kStoreL>(acc[0], lds, tau, p, cap, amb);
      BC_STAMP(43 + p);
      if (Ops::kStoreL) {  // rows of L_d: cap[q] = l_ic * l_cc with i = li, c = lq + 4q
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = lq + 4 * q;
          const double pvc = lds.rs[p & 1][c];
          if (c <= li) ops.store_sym(p * 16 + li, p * 16 + c, cap[q] * (pvc > 0.0 ? rsqrt_nr(pvc) : 0.0));
        }
      }
    }
#if !defined(PLV_BC_FLAGS) && defined(PLV_BC_FOLLOW_BARRIER)
    __syncthreads();  // the panel's sixteen steps and its pivots are in LDS: the strips apply them in one go (strip_follow)
#endif
#ifdef PLV_BC_SOLO
    if (false) {
#else
    if (!(is_sym && t == p) && below) {
#endif
#if !defined(PLV_BC_FLAGS) && defined(PLV_BC_FOLLOW_BARRIER)
      x = strip_follow(acc[0], lds, p);
#else
      x = strip_chain(acc[0], lds, p);
#endif
      BC_STAMP(43 + p);
#ifdef PLV_BC_FLAGS
      // the strip that holds the next diagonal tile: that tile needs nothing but this strip's own X (which the lane that would read
      // it back from Lp holds in x already) — updated first, before anything is published
      if (is_sym && t == p + 1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x[s], x[s], acc[1], 0, 0, 0);
      }
      if (lane == 0) __hip_atomic_fetch_add((PLV_LDS int *)&lds.cdone[p], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (is_sym) {
        if (p >= 2) bc_wait_ge(lds, &lds.tdone[p - 2], ntk - 1 - (p - 2) + nbord);  // nobody reads this parity's panel tiles any more
        lds_vstore(reinterpret_cast<d4 *>(&lds.Lp[p & 1][t][lane][0]), x);
        lds_vstore(&lds.x_flag[t], p + 1);
      }
#else
      if (is_sym) *reinterpret_cast<d4 *>(&lds.Lp[p & 1][t][lane][0]) = x;
#endif
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // panel p of this strip is final: out it goes
        const int cc = p * 16 + lq + 4 * q;
        if (is_sym)
          ops.store_sym(t * 16 + li, cc, x[q]);
        else
          ops.store_border(bs * 16 + li, cc, x[q]);
      }
    }
    BC_STAMP(2 + 5 * p);
#ifndef PLV_BC_FLAGS
    __syncthreads();
#endif
    BC_STAMP(3 + 5 * p);
    if (below) {
      const int ntrail = (is_sym ? t : ntk - 1) - p;  // tiles (strip, p+1 .. p+ntrail) live in acc[1 .. ntrail]
#pragma unroll
      for (int c0 = 1; c0 < NT; c0 += 3) {  // three panel tiles per LDS round trip
        if (c0 <= ntrail) {
          d4 a[3];
#pragma unroll
          for (int u = 0; u < 3; ++u)
            if (c0 + u < NT && c0 + u <= ntrail) {
#ifdef PLV_BC_FLAGS
              if (is_sym && p + c0 + u == t) {
                a[u] = x;  // (its own tile: the lane holds what it would read back)
              } else {
                bc_wait_ge(lds, &lds.x_flag[p + c0 + u], p + 1);
                a[u] = lds_vload(reinterpret_cast<const d4 *>(&lds.Lp[p & 1][p + c0 + u][lane][0]));
              }
#else
              a[u] = *reinterpret_cast<const d4 *>(&lds.Lp[p & 1][p + c0 + u][lane][0]);
#endif
            }
#pragma unroll
          for (int u = 0; u < 3; ++u)
            if (c0 + u < NT && c0 + u <= ntrail) {
#ifdef PLV_BC_FLAGS
              if (is_sym && t == p + 1) continue;  // (done above)
#endif
#pragma unroll
              for (int s = 0; s < 4; ++s)
                acc[c0 + u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[u][s], x[s], acc[c0 + u], 0, 0, 0);
            }
        }
      }
#ifdef PLV_BC_FLAGS
      if (lane == 0) __hip_atomic_fetch_add((PLV_LDS int *)&lds.tdone[p], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    }
#pragma unroll
    for (int i = 0; i + 1 < NT; ++i) acc[i] = acc[i + 1];
    BC_STAMP(4 + 5 * p);
  }
  BC_STAMP(50);
#ifdef PLV_BC_FLAGS
  __syncthreads();  // (what follows the factorisation in the kernels expects every wave to be through)
#endif
}

}  // namespace plv
