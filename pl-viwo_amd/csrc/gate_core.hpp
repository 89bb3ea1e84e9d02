// gate_core.hpp — the chi2 gate of one batch entry inside the workgroup that built and projected its rows
// (REF: UpdaterStatistics::get_chi2, PL/update/UpdaterStatistics.cpp:94-117, + the gate of UpdaterCamera.cpp:237-245 / :406-419).
//
// The update chain used to reach the gate in three launches: [triangulation + Jacobians + null space] -> chi2_t_kernel (T = H' Ps for
// every entry as one tile grid) -> chi2_gate_kernel (S = T H'^T + sigma^2 I, bordered Cholesky, verdict, stack).  All of it is per
// entry, so the Jacobian launch's workgroup can go on: its projected block H' (mp <= 32 rows, k <= GATE_KMAX columns) is still in
// LDS, Ps is read straight from the covariance through the column map (L2-resident, one pass), T stays in LDS.  One launch to the
// gate: two kernel boundaries and the T round trip through memory less.  Same arithmetic as the two kernels it
// replaces (MFMA tile products with the same operand order, blocked_chol<2>), so chi2 agrees with them to the last bits.
#pragma once
#include "blocked_chol.hpp"
#include "gate_stage.hpp"
#include "mfma_tile.hpp"
#include "wave_ops.hpp"

#ifndef GATE_STAMP
#define GATE_STAMP(id)
#endif

namespace plv {

// LDS of the gate: this block (it may overlay whatever the launch no longer needs when the gate starts), the prior block
// Ps = P[cols, cols] as its upper triangle — gate_ps_doubles(k) doubles, filled early by gate_stage_prior — and T [GATE_MMAX][GATE_TLD].
struct GateLds {
  double S[GATE_MMAX * (GATE_MMAX + 1)];
  BcLdsT<2> bc;
  double ybuf[64];
  double passflag;
};

__device__ __forceinline__ double gate_wave_sum(double v) {  // (the butterfly of chi2_gate_kernel: same bits)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct GateOps {  // the bordered factorisation of S with r as the border row: y = L^-1 r
  static constexpr bool kStoreL = false;
  const double *S;  // LDS, stride GATE_MMAX + 1
  const double *r;  // LDS, stride rstr
  int rstr, mp;
  double *y;
  __device__ __forceinline__ double sym_raw(int i, int c) const {
    const int hi = min(max(i, c), mp - 1), lo = min(min(i, c), mp - 1);
    return S[lo * (GATE_MMAX + 1) + hi];  // REF: selfadjointView<Upper>
  }
  __device__ __forceinline__ double border_raw(int, int c) const { return r[(size_t)min(c, mp - 1) * rstr]; }
  __device__ __forceinline__ void scales_ready() const {}
  __device__ __forceinline__ double sym_fix(int i, int c, double g) const { return (i >= mp || c >= mp) ? (i == c ? 1.0 : 0.0) : g; }
  __device__ __forceinline__ double border_fix(int b, int c, double g) const { return (b == 0 && c < mp) ? g : 0.0; }
  __device__ __forceinline__ void store_sym(int, int, double) const {}
  __device__ __forceinline__ void store_border(int b, int c, double v) const {
    if (b == 0 && c < mp) y[c] = v;
  }
};

__host__ __device__ inline int gate_ps_doubles(int k) { return k * (k + 1) / 2 + (k + 1) / 2 + 2; }  // triangle + the column map (ints)
__device__ __forceinline__ int gate_tri(int a, int b) {  // position of Ps(a, b) in the packed upper triangle
  const int hi = max(a, b), lo = min(a, b);
  return hi * (hi + 1) / 2 + lo;
}
// Stages Ps = P[cols, cols] (the covariance block T = H' Ps contracts with) into LDS as a packed triangle, by `nthreads` threads
// (tid = 0 .. nthreads - 1, a multiple of 64) that have nothing else to do while the entry is being triangulated: every element one
// load through the column map, issued eight rows at a time.  Round 4a read the B operands of T straight from memory at the gate: 25
// gather loads per tile column, each lane-scattered over ~10 cache lines — 8 us of the gate's 20.  The caller puts a barrier
// between this and gate_tail.
__device__ __forceinline__ void gate_stage_prior(const GateStage &g, double *Ps, const int *cols_g, int k, int tid, int nthreads) {
  int *cols_l = reinterpret_cast<int *>(Ps + k * (k + 1) / 2);
  for (int i = tid; i < k; i += nthreads) cols_l[i] = cols_g[i];
  const int w = tid >> 6, nw = nthreads >> 6, lane = tid & 63;
  for (int h0 = w; h0 < k; h0 += 8 * nw) {  // rows h0, h0 + nw, .. of this wave, eight per round; columns lane, lane + 64 (<= row)
    double v[8][2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int hi = min(h0 + u * nw, k - 1);
      const double *row = g.P + (size_t)cols_g[hi] * g.ldp;
#pragma unroll
      for (int c = 0; c < 2; ++c) v[u][c] = row[cols_g[min(lane + 64 * c, hi)]];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int hi = h0 + u * nw;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int lo = lane + 64 * c;
        if (hi < k && lo <= hi) Ps[hi * (hi + 1) / 2 + lo] = v[u][c];
      }
    }
  }
}

// Called by ALL 256 threads of the workgroup of entry f (block-uniform arguments).  X: the entry's block in LDS, row-major with
// ncol = fdim + k + 1 columns; rows `shift` .. rows - 1 hold the projected system [.. | H' | r] (shift = fdim when the null space was
// applied, rows = 0 for an entry the selection did not take).  cols_g: the column map (k entries).
// Ps: the staged prior block (gate_stage_prior, complete and visible: a barrier in between), T: GATE_MMAX x GATE_TLD doubles.
__device__ __forceinline__ void gate_tail(const GateStage &g, GateLds &L, const double *Ps, double *T, int f, const double *X, int ncol, int fdim,
                                          int shift, int rows, int k, const int *cols_g) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
  const int mp = rows - fdim;
  const bool valid = shift == fdim && mp >= 1 && mp <= GATE_MMAX && rows >= g.min_rows;  // block-uniform
  if (f == 0 && threadIdx.x == 0 && g.n_acc_next) *g.n_acc_next = 0;
  double chi = NAN, nrm2 = 0.0;
  if (valid) {
    const double *Hp = X + (size_t)shift * ncol + fdim;  // H'(i, a) = Hp[i * ncol + a], r(i) = Hp[i * ncol + k]
    if (threadIdx.x == 0) {  // (read behind the two barriers below)
      L.bc.bad = 0;
      L.bc.step_flag = 0;
      L.bc.rs_flag = 0;
      L.bc.n_amb = 0;
    }
    const int mt = (mp + 15) >> 4, kt = (k + 15) >> 4;
    // T = H' Ps from LDS: H' in the entry's block, Ps(a, b) in the staged triangle.  Per tile the MFMA sequence is chi2_t_kernel's
    // (k ascending, four k per step): the same bits.  What sets the pace here is the number of instructions around the MFMAs, not
    // the matrix pipe (a wave issues one every 5-8 cycles): a wave therefore owns tile COLUMNS — the B operands of a column block
    // (a triangle lookup each) are formed once and serve every row tile — B carries the zero beyond k (A then needs no clamp: what
    // lies behind column k of a row of the block is finite), and the MFMAs run in groups of four without a test in between.
    const int ngroups = (k + 15) >> 4;
    for (int tj = wave; tj < kt; tj += 4) {
      const int jb = min(tj * 16 + li, k - 1);
      double bv[GATE_KMAX / 4];
#pragma unroll
      for (int u = 0; u < GATE_KMAX / 4; ++u) {
        const int kk = 4 * u + lq;
        const double x = Ps[kk < k ? gate_tri(kk, jb) : 0];
        bv[u] = kk < k ? x : 0.0;
      }
      for (int ti = 0; ti < mt; ++ti) {
        const double *Hr = Hp + (size_t)min(ti * 16 + li, mp - 1) * ncol + lq;
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int gq = 0; gq < GATE_KMAX / 16; ++gq)
          if (gq < ngroups) {  // (uniform)
            double av[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) av[u] = (16 * gq + 4 * u + lq < k) ? Hr[16 * gq + 4 * u] : 0.0;  // (columns beyond k: whatever LDS holds there must not reach the MFMA, 0 * NaN = NaN)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[4 * gq + u], acc, 0, 0, 0);
          }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = ti * 16 + lq + 4 * q, j = tj * 16 + li;
          if (i < mp && j < k) T[i * GATE_TLD + j] = acc[q];
        }
      }
    }
    __syncthreads();
    GATE_STAMP(6);
    // S = T H'^T + sigma2 I, tiles on and above the diagonal
    for (int t = wave; t < mt * mt; t += 4) {
      const int ti = t / mt, tj = t - ti * mt;
      if (tj < ti) continue;
      const double *Tr = T + (size_t)min(ti * 16 + li, mp - 1) * GATE_TLD;
      const double *Hr = Hp + (size_t)min(tj * 16 + li, mp - 1) * ncol;
      d4 acc = {0, 0, 0, 0};
      auto fa = [&](int, int kk) { return Tr[kk]; };
      auto fb = [&](int kk, int) { return Hr[kk]; };
      acc = mfma_tile_f64_pipe<16>(fa, fb, k, acc);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = ti * 16 + lq + 4 * q, j = tj * 16 + li;
        if (i < mp && j < mp) L.S[i * (GATE_MMAX + 1) + j] = acc[q] + (i == j ? g.sigma2 : 0.0);
      }
    }
    __syncthreads();
    GATE_STAMP(7);
    if (wave == 0) {
      const double rv = lane < mp ? Hp[(size_t)lane * ncol + k] : 0.0;
      nrm2 = gate_wave_sum(rv * rv);
    }
    GateOps ops{L.S, Hp + k, ncol, mp, L.ybuf};
    blocked_chol<2>(ops, L.bc, mp, 1, 0.0, 0);
    __syncthreads();
    GATE_STAMP(8);
    if (wave == 0) {
      const double y = lane < mp ? L.ybuf[lane] : 0.0;
      chi = gate_wave_sum(y * y);
      if (L.bc.bad) chi = NAN;
    }
  }
  if (wave == 0 && lane == 0) {
    g.chi2[f] = chi;
    if (g.dec) g.dec[3 * f] = valid ? chi : NAN, g.dec[3 * f + 1] = (valid && mp < g.q95_n) ? g.chi2_mult * g.q95[mp] : NAN, g.dec[3 * f + 2] = valid ? sqrt(nrm2) : NAN;
    bool pass = valid && !isnan(chi);
    if (pass && g.res_norm_gate > 0.0) pass = sqrt(nrm2) < g.res_norm_gate;
    if (pass) pass = (mp < g.q95_n) && (chi < g.chi2_mult * g.q95[mp]);
    g.accepted[f] = pass ? 1 : 0;
    // an entry with more projected rows than the gate holds cannot be judged here: -1 (the host's wait reports PLV_E_CAPACITY instead
    // of losing the measurement silently; the launcher's row hint is meant to keep such a batch on the separate kernels)
    const bool overflow = shift == fdim && mp > GATE_MMAX;
    if (g.acc_rows) g.acc_rows[f] = pass ? mp : (overflow ? -1 : 0);
    if (g.h_accepted) g.h_accepted[f] = pass ? 1 : 0;
    if (g.h_acc_rows) g.h_acc_rows[f] = pass ? mp : (overflow ? -1 : 0);
    if (pass && g.n_acc) atomicAdd(g.n_acc, 1);
    L.passflag = pass ? 1.0 : 0.0;
  }
  if (g.probe_dst) {
    for (int i = threadIdx.x; i < g.probe_stride_a; i += blockDim.x) g.probe_dst[(size_t)f * g.probe_stride_a + i] = g.probe_src[(size_t)f * g.probe_stride_a + i];
    for (int i = threadIdx.x; i < g.probe_stride_b; i += blockDim.x)
      g.probe_dst[(size_t)g.probe_off_b + (size_t)f * g.probe_stride_b + i] = g.probe_src[(size_t)g.probe_off_b + (size_t)f * g.probe_stride_b + i];
  }
  if (g.stack) {
    __syncthreads();
    const bool pass = L.passflag != 0.0;
    if (!pass && g.stack_accepted_only) return;
    const double *Hp = X + (size_t)shift * ncol + fdim;
    double *dst = g.stack + (size_t)f * g.mp_max;
    // (the consumer of an accepted-only stack walks acc_rows rows of every entry — gram_direct_kernel — and the Householder route zeroes
    // the rest of a slot itself, stack_zero_rejected_kernel: the padding rows of an entry are not written — 40 % of the launch's writes)
    const int rows_w = g.stack_accepted_only ? mp : g.mp_max;
    for (int i = threadIdx.x & 31; i < rows_w; i += 32)      // (32 rows x 8 columns per pass: no integer division per element)
      for (int j = threadIdx.x >> 5; j <= k; j += 8) {
        double v = 0.0;
        if (pass && i < mp) v = Hp[(size_t)i * ncol + j];  // (column k of the block is r)
        dst[(size_t)j * g.lds + i] = v;
      }
  }
}

}  // namespace plv
