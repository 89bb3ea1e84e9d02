// update_kernels.hip — gfx950 kernels of the EKF-update half of the hot path (fp64).
//
//   nullspace_kernel   K11  StateHelper::nullspace_project_inplace   REF: PL/state/StateHelper.cpp:616-651
//   chi2_gate_kernel   K12  UpdaterStatistics::get_chi2 + the gate   REF: PL/update/UpdaterStatistics.cpp:94-117,
//                                                                         PL/update/cam/UpdaterCamera.cpp:237-263
//   qr_accum_kernel    K13  StateHelper::measurement_compress_inplace REF: StateHelper.cpp:602-614,653-672
//   ekf_*_kernel       K14  StateHelper::EKFUpdate                   REF: StateHelper.cpp:94-173
//
// Design notes (DESIGN.md has the long form):
//  * wave = 64 lanes; every dense contraction goes through v_mfma_f64_16x16x4_f64 tiles, one
//    16x16 output tile per wave, operands read straight from L2-resident global memory or LDS
//    (the matrices are ~100x100: nothing here is HBM-bound, everything is latency-bound);
//  * the Givens nullspace keeps the reference's exact rotation order, one thread per column with
//    the pivot column replicated, so no intra-pass communication is needed;
//  * the compression is a Householder TSQR: a row chunk lives in registers (16 rows x 1 column
//    per thread), the running R factor in LDS.
#include <algorithm>

#include "plv_ctx.hpp"
#include "blocked_chol.hpp"
#include "mfma_tile.hpp"
#include "nullspace_core.hpp"
#include "update_kernels.hpp"

namespace plv {

// ------------------------------------------------------------------------------------------
// One wave computes one 16x16 fp64 tile  acc(i,j) += sum_k a(i,k) * b(k,j)  with
// v_mfma_f64_16x16x4_f64.  Operand lane map: lane l feeds A[i = l&15][k = l>>4] and
// B[k = l>>4][j = l&15]; result lane map: col = l&15, row = (l>>4) + 4*reg.
// All 64 lanes must call this together (EXEC all ones); accessors return 0 out of range.
template <class FA, class FB>
__device__ __forceinline__ d4 mfma_tile_f64(FA a, FB b, int K, d4 acc) {
  const int lane = threadIdx.x & 63;
  const int ij = lane & 15, kq = lane >> 4;
  int k0 = 0;
  for (; k0 + 16 <= K; k0 += 16) {  // operand loads of four k-steps issued ahead of the MFMAs
    double av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      av[u] = a(ij, k0 + 4 * u + kq);
      bv[u] = b(k0 + 4 * u + kq, ij);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
  }
  for (; k0 < K; k0 += 4) {
    const int kk = k0 + kq;
    const bool in = kk < K;
    const double av = in ? a(ij, kk) : 0.0;
    const double bv = in ? b(kk, ij) : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
  return acc;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Eigen::JacobiRotation<double>::makeGivens, real case (Eigen/src/Jacobi/Jacobi.h).
__device__ __forceinline__ void make_givens(double p, double q, double &c, double &s) {
  if (q == 0.0) {
    c = p < 0.0 ? -1.0 : 1.0;
    s = 0.0;
  } else if (p == 0.0) {
    c = 0.0;
    s = q < 0.0 ? 1.0 : -1.0;
  } else if (fabs(p) > fabs(q)) {
    double t = q / p;
    double u = sqrt(1.0 + t * t);
    if (p < 0.0) u = -u;
    c = 1.0 / u;
    s = -t * c;
  } else {
    double t = p / q;
    double u = sqrt(1.0 + t * t);
    if (q < 0.0) u = -u;
    s = -1.0 / u;
    c = -t * s;
  }
}

// ------------------------------------------------------------------------------------------
// K11: one workgroup per feature.  X = [Hf | Hx | res] (rows x ncol) staged row-major in LDS,
// one thread per column.  For pivot column n the rotation sequence m = rows-1 .. n+1 is the
// reference's; (c,s) are recomputed by every thread from a read-only copy of the pivot column,
// which reproduces bit-for-bit what the column's owner computes, so threads never wait on each
// other inside a pass.  The chain of rows-1 dependent rotations per pivot column is the critical path
// (a register-resident variant without the LDS round trips measured the same 29-32 us).
__global__ void __launch_bounds__(256) nullspace_kernel(int fdim, int k, int ld, const int *__restrict__ rows_arr,
                                                        double *__restrict__ Hf, double *__restrict__ Hx,
                                                        double *__restrict__ res, int F, GatherArgs g, int shift) {
  extern __shared__ double smem[];
  if ((int)blockIdx.x >= F) {  // independent work riding on the same launch: the dense covariance gathers
    gather_cov_block(g, blockIdx.x - F);
    return;
  }
  const int f = blockIdx.x;
  const int rows = rows_arr[f];
  const int ncol = fdim + k + 1;
  double *X = smem;               // [ld][ncol]
  double *piv = smem + ld * ncol; // [ld]
  double *gHf = Hf + (size_t)f * fdim * ld;
  double *gHx = Hx + (size_t)f * k * ld;
  double *gres = res + (size_t)f * ld;
  if (rows <= fdim) return;

  // thread per column: `rows` contiguous doubles of its column, eight loads in flight at a time; a wave then covers one
  // contiguous stretch of the batch and the LDS writes of a row are conflict-free
  for (int j = threadIdx.x; j < ncol; j += blockDim.x) {
    const double *src = j < fdim ? gHf + j * ld : (j < fdim + k ? gHx + (size_t)(j - fdim) * ld : gres);
    for (int i0 = 0; i0 < rows; i0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[min(i0 + u, rows - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u < rows) X[(i0 + u) * ncol + j] = v[u];
    }
  }
  nullspace_rotate(X, piv, rows, ncol, fdim);
  // write back: Hf whole (upper-triangular now), Hx / res shifted up by `shift` rows (= fdim for the nullspace
  // projection; 0 keeps the initialising rows as well: StateHelper::initialize, StateHelper.cpp:391-405)
  const int mp = rows - shift;
  for (int j = threadIdx.x; j < ncol; j += blockDim.x) {
    const bool isf = j < fdim;
    double *dst = isf ? gHf + j * ld : (j < fdim + k ? gHx + (size_t)(j - fdim) * ld : gres);
    const int cnt = isf ? rows : mp, off = isf ? 0 : shift;
    for (int i0 = 0; i0 < cnt; i0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = X[(min(i0 + u, cnt - 1) + off) * ncol + j];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u < cnt) dst[i0 + u] = v[u];
    }
  }
}

// ------------------------------------------------------------------------------------------
// K12: one workgroup (4 waves) per feature.  mp = rows[f] - fdim_off projected rows.
//   T = H' * P[cols,cols]   (MFMA tiles, T kept in LDS)
//   S = T * H'^T + sigma2 I (MFMA tiles, S in LDS)
//   Cholesky of the bordered matrix [S r; r^T .] by wave 0 (lane = row): the border row is
//   y = L^-1 r, so chi2 = |y|^2  (== r^T S^-1 r, what the reference gets from S.inverse()).
// If `stack` is non-null the gate is applied and the accepted system is copied (rejected: zeros)
// into rows [f*mp_max, (f+1)*mp_max) of the stacked matrix [H | r] (col-major, ld = lds).

#define CHI2_MAXM 63

// Ops of blocked_chol for one feature's bordered system: S (upper triangle valid, LDS, pitch 65) and the
// residual as the single border row; the border comes back as y = L^-1 r.
struct Chi2Ops {
  static constexpr bool kStoreL = false;
  const double *S;  // LDS
  const double *r;  // global
  int mp;
  double *y;        // LDS [64]
  __device__ __forceinline__ double sym_raw(int i, int c) const {
    const int hi = min(max(i, c), mp - 1), lo = min(min(i, c), mp - 1);
    return S[lo * 65 + hi];  // REF: selfadjointView<Upper>
  }
  __device__ __forceinline__ double border_raw(int, int c) const { return r[min(c, mp - 1)]; }
  __device__ __forceinline__ void scales_ready() const {}
  __device__ __forceinline__ double sym_fix(int i, int c, double g) const { return (i >= mp || c >= mp) ? (i == c ? 1.0 : 0.0) : g; }
  __device__ __forceinline__ double border_fix(int b, int c, double g) const { return (b == 0 && c < mp) ? g : 0.0; }
  __device__ __forceinline__ void store_sym(int, int, double) const {}
  __device__ __forceinline__ void store_border(int b, int c, double v) const {
    if (b == 0 && c < mp) y[c] = v;
  }
};

// NT = number of 16-row strips of S (2 for mp <= 32, 4 up to 63); blockDim = 64 * max(4, NT + 1).
template <int NT>
__global__ void __launch_bounds__(64 * (NT + 1 > 4 ? NT + 1 : 4)) chi2_gate_kernel(Chi2Args a) {
  extern __shared__ double smem[];
  __shared__ BcLds lds;
  __shared__ double ybuf[64];
  __shared__ double passflag;
  const int f = blockIdx.x;
  const int rows_f = a.rows[f];
  const int mp = rows_f - a.fdim_off;
  const int k = a.k, ld = a.ld;
  const int mt = (max(mp, 1) + 15) >> 4;
  double *S = smem;  // [64][65]
  const double *H = a.Hx + (size_t)f * k * ld;
  const double *r = a.res + (size_t)f * ld;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int trow = lane >> 4, tcol = lane & 15;

  const bool valid = mp >= 1 && mp <= 16 * NT && mp <= CHI2_MAXM && rows_f >= a.min_rows;  // block-uniform
  double chi = NAN;
  double nrm2 = 0.0;
  if (valid) {
    // T = H' Ps was produced for every feature by chi2_t_kernel (col-major, ld)
    const double *Tg = a.T + (size_t)f * k * ld;
    // S = T H'^T + sigma2 I, upper tiles (REF: selfadjointView<Upper>)
    for (int t = wave; t < mt * mt; t += nw) {
      const int ti = t / mt, tj = t - ti * mt;
      if (tj < ti) continue;
      d4 acc = {0, 0, 0, 0};
      // rows beyond mp only feed entries that are not stored: clamp instead of branching (loads stay in flight)
      const double *Tr = Tg + min(ti * 16 + tcol, mp - 1), *Hr = H + min(tj * 16 + tcol, mp - 1);
      auto fa = [&](int, int kk) { return Tr[kk * ld]; };
      auto fb = [&](int kk, int) { return Hr[kk * ld]; };
      acc = mfma_tile_f64_pipe<16>(fa, fb, k, acc);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int i = ti * 16 + trow + 4 * q, j = tj * 16 + tcol;
        if (i < mp && j < mp) S[i * 65 + j] = acc[q] + (i == j ? a.sigma2 : 0.0);
      }
    }
    if (threadIdx.x == 0) {
      lds.bad = 0;
      lds.step_flag = 0;
      lds.rs_flag = 0;
    }
    __syncthreads();
    if (wave == 0) {
      const double rv = lane < mp ? r[lane] : 0.0;
      nrm2 = wave_sum(rv * rv);
    }
    // bordered Cholesky: the border row ends as y = L^-1 r, chi2 = |y|^2 (== r^T S^-1 r of the reference)
    Chi2Ops ops{S, r, mp, ybuf};
    blocked_chol<NT>(ops, lds, mp, 1, 0.0, 0);
    __syncthreads();
    if (wave == 0) {
      const double y = lane < mp ? ybuf[lane] : 0.0;
      chi = wave_sum(y * y);
      if (lds.bad) chi = NAN;
    }
  }
  if (wave == 0 && lane == 0) {
    a.chi2[f] = chi;
    if (a.dec) a.dec[3 * f] = valid ? chi : NAN, a.dec[3 * f + 1] = (valid && mp < a.q95_n) ? a.chi2_mult * a.q95[mp] : NAN, a.dec[3 * f + 2] = valid ? sqrt(nrm2) : NAN;
    passflag = 0.0;
    if (a.stack) {
      bool pass = valid && !isnan(chi);
      if (pass && a.res_norm_gate > 0.0) pass = sqrt(nrm2) < a.res_norm_gate;
      if (pass) pass = (mp < a.q95_n) && (chi < a.chi2_mult * a.q95[mp]);
      a.accepted[f] = pass ? 1 : 0;
      if (a.acc_rows) a.acc_rows[f] = pass ? mp : 0;
      if (a.h_accepted) a.h_accepted[f] = pass ? 1 : 0;
      if (a.h_acc_rows) a.h_acc_rows[f] = pass ? mp : 0;
      if (pass && a.n_acc) atomicAdd(a.n_acc, 1);
      passflag = pass ? 1.0 : 0.0;
    }
  }
  if (a.probe_dst) {
    for (int i = threadIdx.x; i < a.probe_stride_a; i += blockDim.x) a.probe_dst[(size_t)f * a.probe_stride_a + i] = a.probe_src[(size_t)f * a.probe_stride_a + i];
    for (int i = threadIdx.x; i < a.probe_stride_b; i += blockDim.x)
      a.probe_dst[(size_t)a.probe_off_b + (size_t)f * a.probe_stride_b + i] = a.probe_src[(size_t)a.probe_off_b + (size_t)f * a.probe_stride_b + i];
  }
  if (a.stack) {
    __syncthreads();
    const bool pass = passflag != 0.0;
    if (!pass && a.stack_accepted_only) return;  // (1.2 MB of zeros per update that nothing reads: rocprofv3 WRITE_SIZE, profiles/r03)
    double *dst = a.stack + (size_t)f * a.mp_max;
    const int rows_w = a.stack_accepted_only ? mp : a.mp_max;  // (padding rows of an accepted-only stack are never read: gate_core.hpp)
    for (int idx = threadIdx.x; idx < rows_w * (k + 1); idx += blockDim.x) {
      int j = idx / rows_w, i = idx - j * rows_w;
      double v = 0.0;
      if (pass && i < mp) v = j < k ? H[j * ld + i] : r[i];
      dst[(size_t)j * a.lds + i] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// K13: Householder TSQR.  Workgroup w reduces rows [w*rows_per_wg, (w+1)*rows_per_wg) of the
// m x nc matrix A (col-major, lda; the last column is the residual) to an nc x nc upper
// triangle R_w, written row-major-free as col-major [nc x nc] at out + w*nc (ld = ldo), so the
// outputs of all workgroups form the next level's input matrix.
// Thread t = 4*c + part owns, for column c, rows part*16 .. part*16+15 of the current 64-row
// chunk in registers; the running R lives in LDS (row-major nc x nc).
#define QR_RPT 16
#define QR_PARTS 4
#define QR_CHUNK (QR_RPT * QR_PARTS)

__global__ void __launch_bounds__(1024) qr_accum_kernel(const double *__restrict__ A, int lda, int m, int nc,
                                                        int rows_per_wg, double *__restrict__ out, int ldo,
                                                        int sign_fix) {
  extern __shared__ double smem[];
  double *R = smem;              // [nc][nc] row-major
  double *v = R + nc * nc;       // [QR_CHUNK] reflector tail
  double *hdr = v + QR_CHUNK;    // [0]=tau  [1]=unused
  const int t = threadIdx.x;
  const int c = t >> 2, part = t & 3;
  const bool active = c < nc;
  const int row_begin = blockIdx.x * rows_per_wg;
  const int row_end = min(m, row_begin + rows_per_wg);

  for (int idx = t; idx < nc * nc; idx += blockDim.x) R[idx] = 0.0;
  __syncthreads();

  for (int r0 = row_begin; r0 < row_end; r0 += QR_CHUNK) {
    double b[QR_RPT];
#pragma unroll
    for (int i = 0; i < QR_RPT; ++i) {
      int row = r0 + part * QR_RPT + i;
      b[i] = (active && row < row_end) ? A[(size_t)c * lda + row] : 0.0;
    }
    for (int j = 0; j < nc; ++j) {
      // ---- reflector for column j from [R_jj ; chunk column j]
      if (c == j) {
        double ss = 0.0;
#pragma unroll
        for (int i = 0; i < QR_RPT; ++i) ss += b[i] * b[i];
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        const double alpha = R[j * nc + j];
        double tau = 0.0, scale = 0.0, beta = alpha;
        if (ss > 0.0) {
          beta = sqrt(alpha * alpha + ss);
          if (alpha >= 0.0) beta = -beta;
          tau = (beta - alpha) / beta;
          scale = 1.0 / (alpha - beta);
        }
#pragma unroll
        for (int i = 0; i < QR_RPT; ++i) v[part * QR_RPT + i] = b[i] * scale;
        if (part == 0) {
          hdr[0] = tau;
          R[j * nc + j] = beta;
        }
      }
      __syncthreads();
      const double tau = hdr[0];
      if (active && c > j && tau != 0.0) {
        double vv[QR_RPT];
        double w = 0.0;
#pragma unroll
        for (int i = 0; i < QR_RPT; ++i) {
          vv[i] = v[part * QR_RPT + i];
          w += vv[i] * b[i];
        }
        w += __shfl_xor(w, 1, 64);
        w += __shfl_xor(w, 2, 64);
        const double rjc = R[j * nc + c];
        w += rjc;
        const double tw = tau * w;
        if (part == 0) R[j * nc + c] = rjc - tw;
#pragma unroll
        for (int i = 0; i < QR_RPT; ++i) b[i] -= tw * vv[i];
      }
      __syncthreads();
    }
  }
  // ---- write R (col-major into the next level's matrix); optionally make diag >= 0
  for (int idx = t; idx < nc * nc; idx += blockDim.x) {
    int i = idx / nc, j = idx - i * nc;
    double val = j >= i ? R[i * nc + j] : 0.0;
    if (sign_fix && R[i * nc + i] < 0.0) val = -val;
    out[(size_t)j * ldo + (size_t)blockIdx.x * nc + i] = val;
  }
}

// ------------------------------------------------------------------------------------------
// K13b (round 6b): the same factor by a BLOCKED Householder QR in ONE workgroup, in place.  The tree above spends a column step (a
// barrier or two, ~1 us) per column, per 64-row chunk and per level — 1260 steps for a 750 x 105 stack — although the stack is only
// seven times taller than wide.  Here a column step spans all rows at once: panels of 16 columns live in the registers of the 512
// threads (thread t holds rows d0 + t, d0 + t + 512, ..), a step's norm is one wave sum + eight partials, its dot products with the
// panel's later columns go through LDS transposed (every thread writes its 15 partial products, 32 lanes add 512 of them per column)
// — 105 steps in all —, and the columns behind the panel are updated with the panel's block reflector I - V T V^T on MFMA tiles
// (W = V^T A_tile summed over the waves' row ranges, T^T W, A_tile -= V (T^T W)); V stays where the panel was (unit diagonal
// implied), R ends in the top nc rows.  Same reflectors as the unblocked kernel (LAPACK dgeqrt's arithmetic), so the factor agrees
// with it to rounding.  m <= 512 * RP rows.  Measured (tools/tsqr_ab.py, events around the launches, one box): 750 x 105 599 us
// against 1400 for the tree's three launches, 300 x 105 396 against 955, 900 x 135 ~1100 against 2350, 1500 x 105 ~1100 against 1900;
// of the 599: the 105 column steps 230 (2.2 us each: sixteen sums over 512 threads through LDS, a square root and two divisions),
// V^T V and T 72, the eight passes over the trailing columns ~300 (every round of loads from the L2 is ~1 us for a lone workgroup).
#define HQ_T 512
template <int RP>
__global__ void __launch_bounds__(HQ_T) hqr_kernel(double *__restrict__ A, int lda, int m, int nc, int sign_fix, double *__restrict__ Vg, int ldv, const int *__restrict__ m_dev,
           double *__restrict__ Rout, int ldro, const int *__restrict__ first_m_dev, int first_m, int first_cap) {
  // Several workgroups (a stack of more rows than one holds in registers): the rows are split evenly, workgroup b factors its share in
  // place and leaves its R — nc rows — at rows b * nc of Rout; a second launch factors the stacked R's.  m_dev: the rows that hold
  // anything (a stack gathered on the device), at least nc of them taken (rows of zeros change no factor).
  // (the launch over the stacked R's: nothing to do when the first launch's rows fitted one workgroup — that one left the final R)
  if (first_m_dev && min(max(*first_m_dev, nc), first_m) <= first_cap) return;
  {
    const int m_tot = m_dev ? min(max(*m_dev, nc), m) : m;
    const int cap = HQ_T * RP, nb_act = (m_tot + cap - 1) / cap, bs = (((m_tot + nb_act - 1) / nb_act) + 15) & ~15;
    if (sign_fix == 2) sign_fix = nb_act == 1 ? 1 : 0;  // (first of two launches: final when it is alone)
    const int b = blockIdx.x, r_lo = b * bs, r_hi = min(m_tot, r_lo + bs);
    if (b >= nb_act) {  // nothing for this workgroup: an R of zeros
      for (int idx = threadIdx.x; idx < nc * nc; idx += HQ_T) Rout[(size_t)(idx / nc) * ldro + (size_t)b * nc + idx % nc] = 0.0;
      return;
    }
    A += r_lo;
    m = r_hi - r_lo;
    Vg += (size_t)b * ldv * 16;
  }
  __shared__ double buf[16 * HQ_T];  // a step's partial products [column][thread]; the MFMA phases' partial tiles [wave][256]
  __shared__ double wv[2][16], prow[2][16], tau_s[16], sg[208];  // (wv, prow: by the parity of the column step)
  __shared__ double VtV[16][17], Tm[16][17], Wm[4][16][17], W2[4][16][17];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int np = (nc + 15) >> 4;
  for (int p = 0; p < np; ++p) {
    const int c0 = 16 * p, d0 = c0, bw = min(16, nc - c0), mp = m - d0;
    double a[RP][16];
#pragma unroll
    for (int q = 0; q < RP; ++q) {
      const int row = d0 + t + HQ_T * q;
#pragma unroll
      for (int c = 0; c < 16; ++c) a[q][c] = (row < m && c < bw) ? A[(size_t)(c0 + c) * lda + row] : 0.0;
    }
    // (slot 0 of the register row is always the step's pivot column: a finished column goes to memory and the others move up, so the
    // step is ONE piece of code with fixed register indices — unrolled sixteen times it was 80 KB of instructions, more than the
    // instruction cache holds, and a step took 3.5 us; slots beyond the panel's width hold zeros)
    for (int j = 0; j < bw; ++j) {
      const int d = d0 + j;
      // one round of sums for the step: x_0 = sum over the rows below the pivot of a_j^2, x_c = the same rows' a_j a_c (c = 1 .. 15);
      // the pivot row's own entries go through LDS as they are.  With v = a_j * scale below the pivot and 1 on it, v^T a_c is
      // a_dc + scale * x_c: one barrier less per step than the norm first and the products with v afterwards.
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        double pc = 0.0;
#pragma unroll
        for (int q = 0; q < RP; ++q)
          if (d0 + t + HQ_T * q > d) pc += a[q][0] * a[q][c];  // (rows beyond m hold zeros)
        buf[c * HQ_T + t] = pc;
      }
      if (t == j) {
#pragma unroll
        for (int c = 0; c < 16; ++c) prow[j & 1][c] = a[0][c];
      }
      __syncthreads();
      {
        const int c = t >> 5, sgm = t & 31;
        const double *src = buf + c * HQ_T + sgm;  // (lanes side by side: every read of the wave covers all banks once)
        double x = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) x += src[32 * u];
        x += dpp_mov_f64<0x111>(x);
        x += dpp_mov_f64<0x112>(x);
        x += dpp_mov_f64<0x114>(x);
        x += dpp_mov_f64<0x118>(x);
        x += dpp_mov_f64<0x142>(x);  // (lane 31: rows 0 + 1 of the wave, lane 63: rows 2 + 3 — the wave's two columns)
        if ((lane & 31) == 31) wv[j & 1][c] = x;
      }
      __syncthreads();
      const double ss = wv[j & 1][0], alpha = prow[j & 1][0];
      double tau = 0.0, scale = 0.0, beta = alpha;
      if (ss > 0.0) {
        beta = sqrt(alpha * alpha + ss);
        if (alpha >= 0.0) beta = -beta;
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
      }
      double v[RP];
#pragma unroll
      for (int q = 0; q < RP; ++q) {
        const int row = d0 + t + HQ_T * q;
        v[q] = row > d ? a[q][0] * scale : (row == d ? 1.0 : 0.0);
      }
      // the finished column to memory (R above and on the diagonal, v below), the others one slot up with the reflector applied
#pragma unroll
      for (int q = 0; q < RP; ++q) {
        const int row = d0 + t + HQ_T * q;
        if (row < m) {
          A[(size_t)(c0 + j) * lda + row] = row > d ? v[q] : (row == d ? beta : a[q][0]);
          Vg[(size_t)j * ldv + (row - d0)] = v[q];  // (the reflector with its unit diagonal and the zeros above it: the MFMA phases' operand)
        }
      }
#pragma unroll
      for (int c = 1; c < 16; ++c) {
        const double tw = tau * (prow[j & 1][c] + scale * wv[j & 1][c]);
#pragma unroll
        for (int q = 0; q < RP; ++q) a[q][c - 1] = a[q][c] - tw * v[q];
      }
#pragma unroll
      for (int q = 0; q < RP; ++q) a[q][15] = 0.0;
      if (t == 0) tau_s[j] = tau;
      // (two barriers a step: the next step writes buf before its first barrier, behind this step's second, which every reader of
      //  buf has passed; prow and wv alternate by the step's parity)
    }
    __syncthreads();
    if (c0 + 16 >= nc) break;  // (the last panel: nothing behind it)
    const int ck = ((mp + 7) / 8 + 3) & ~3;  // rows of a wave's range (a multiple of the MFMA's four k)
    const int off0 = wave * ck, rb0 = d0 + off0, rk = max(0, min(ck, m - rb0));
    const double *vcol = Vg + (size_t)(lane & 15) * ldv;  // this lane's column of V (operand maps: lane & 15 = the tile's row / column)
    {  // ---- VtV = V^T V, then T
      d4 acc = {0, 0, 0, 0};
      auto fa = [&](int, int kk) { return vcol[off0 + kk]; };
      auto fb = [&](int kk, int) { return vcol[off0 + kk]; };
      acc = mfma_tile_f64_pipe<8>(fa, fb, rk, acc);
#pragma unroll
      for (int q = 0; q < 4; ++q) buf[wave * 256 + ((lane >> 4) + 4 * q) * 16 + (lane & 15)] = acc[q];
      __syncthreads();
      if (t < 256) {
        double x = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) x += buf[w * 256 + t];
        VtV[t >> 4][t & 15] = x;
      }
      __syncthreads();
      if (t < 16) {  // row t of T (upper triangular): T(0:j, j) = -tau_j T(0:j, 0:j) V(:, 0:j)^T v_j
        for (int jj = 0; jj < 16; ++jj) {
          double x = 0.0;
          if (jj == t) x = jj < bw ? tau_s[jj] : 0.0;
          if (jj > t && jj < bw) {
            double s_ = 0.0;
            for (int l = t; l < jj; ++l) s_ += Tm[t][l] * VtV[l][jj];
            x = -tau_s[jj] * s_;
          }
          Tm[t][jj] = x;
        }
      }
      __syncthreads();
    }
    // the columns behind the panel, four tiles (64 columns) a pass: the loads of a pass are in flight together — V once, the tiles
    // beside it — and a pass has one set of barriers (a tile a pass took 17 us of mostly waiting at 750 rows)
    for (int tc0 = c0 + 16; tc0 < nc; tc0 += 64) {
      const int ng = min(4, (nc - tc0 + 15) >> 4);  // tiles of this pass
      {  // W_g = V^T A_tile_g over the waves' row ranges
        d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        const double *acol[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acol[g] = A + (size_t)min(tc0 + 16 * g + (lane & 15), nc - 1) * lda + rb0;  // (beyond the last column: it again, never used)
        const int kq = lane >> 4;
        for (int k0 = 0; k0 < rk; k0 += 16) {  // (four k-steps of four rows: 4 + 16 loads in flight)
          double av[4], bv[4][4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int kk = k0 + 4 * u + kq, kc = min(kk, rk - 1);
            const double x = vcol[off0 + kc];
            av[u] = kk < rk ? x : 0.0;
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[g][u] = acol[g][kc];
          }
#pragma unroll
          for (int g = 0; g < 4; ++g)
            if (g < ng)
#pragma unroll
              for (int u = 0; u < 4; ++u) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[g][u], acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int q = 0; q < 4; ++q) buf[(wave * 4 + g) * 256 + ((lane >> 4) + 4 * q) * 16 + (lane & 15)] = acc[g][q];
      }
      __syncthreads();
      for (int e = t; e < 1024; e += HQ_T) {  // (tile g = e >> 8, entry e & 255)
        double x = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) x += buf[(w * 4 + (e >> 8)) * 256 + (e & 255)];
        Wm[e >> 8][(e >> 4) & 15][e & 15] = x;
      }
      __syncthreads();
      for (int e = t; e < 1024; e += HQ_T) {  // W2_g = T^T W_g
        const int g = e >> 8, i = (e >> 4) & 15, jj = e & 15;
        double x = 0.0;
        for (int l = 0; l <= i; ++l) x += Tm[l][i] * Wm[g][l][jj];
        W2[g][i][jj] = x;
      }
      __syncthreads();
      for (int rb = wave; rb * 16 < mp; rb += 8) {  // A_tile_g -= V W2_g, sixteen rows a time
        const int r0 = d0 + rb * 16;
        const int ri = min(rb * 16 + (lane & 15), mp - 1);  // (rows beyond the last: a valid row again, its results are not stored)
        double av[4], old[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) av[u] = Vg[(size_t)(4 * u + (lane >> 4)) * ldv + ri];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = min(r0 + (lane >> 4) + 4 * q, m - 1), col = min(tc0 + 16 * g + (lane & 15), nc - 1);
            old[g][q] = A[(size_t)col * lda + row];
          }
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (g < ng) {
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], W2[g][4 * u + (lane >> 4)][lane & 15], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int row = r0 + (lane >> 4) + 4 * q, col = tc0 + 16 * g + (lane & 15);
              if (row < m && col < nc) A[(size_t)col * lda + row] = old[g][q] - acc[q];
            }
          }
      }
      __syncthreads();
    }
  }
  // ---- R alone in the top nc rows: zeros below the diagonal, optionally rows with a negative diagonal negated; with several
  // workgroups: into this workgroup's rows of Rout instead
  __syncthreads();
  for (int i = t; i < nc; i += HQ_T) sg[i] = (sign_fix && A[(size_t)i * lda + i] < 0.0) ? -1.0 : 1.0;
  __syncthreads();
  for (int idx = t; idx < nc * nc; idx += HQ_T) {
    const int j = idx / nc, i = idx - j * nc;  // (column j, row i)
    double *e = A + (size_t)j * lda + i;
    const double x = i > j ? 0.0 : (sg[i] < 0.0 ? -*e : *e);
    if (Rout)
      Rout[(size_t)j * ldro + (size_t)blockIdx.x * nc + i] = x;
    else
      *e = x;
  }
}

// ------------------------------------------------------------------------------------------
// Dense gathers of the covariance blocks the update contracts with, so that no MFMA operand load
// goes through a dependent index load:  Pc = P[cols, :] (k x n, row-major), Ps = P[cols, cols]
// (k x k, row-major), inv[state] = position of that state in cols or -1.
__global__ void __launch_bounds__(256) gather_cov_kernel(GatherArgs g) { gather_cov_block(g, blockIdx.x); }

// T = Hx' * Ps for every feature at once: one 16x16 tile per wave over (feature, row tile, col tile).
__global__ void __launch_bounds__(256) chi2_t_kernel(Chi2Args a, int mt_max) {
  if (a.n_acc && blockIdx.x == 0 && threadIdx.x == 0) {
    a.n_acc[0] = 0;  // counted by chi2_gate_kernel, the next launch
    a.n_acc[2] = 0;  // ambiguous pivots of the compression (status block word 3), written by bchol_compress_kernel when it runs
  }
  const int kt = (a.k + 15) >> 4;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int per_f = mt_max * kt;
  const int f = wid / per_f;
  if (f >= a.F) return;
  const int t = wid - f * per_f, ti = t / kt, tj = t - ti * kt;
  const int mp = a.rows[f] - a.fdim_off;
  if (ti * 16 >= mp) return;
  const int k = a.k, ld = a.ld;
  const double *H = a.Hx + (size_t)f * k * ld;
  double *T = a.T + (size_t)f * k * ld;
  const int lane = threadIdx.x & 63;
  d4 acc = {0, 0, 0, 0};
  const double *Hr = H + min(ti * 16 + (lane & 15), mp - 1);  // rows/cols beyond the range only feed unstored entries
  const double *Pq = a.Ps + min(tj * 16 + (lane & 15), k - 1);
  auto fa = [&](int, int kk) { return Hr[kk * ld]; };
  auto fb = [&](int kk, int) { return Pq[(size_t)kk * k]; };
  acc = mfma_tile_f64_pipe<16>(fa, fb, k, acc);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = ti * 16 + (lane >> 4) + 4 * q, j = tj * 16 + (lane & 15);
    if (i < mp && j < k) T[(size_t)j * ld + i] = acc[q];
  }
}

// ------------------------------------------------------------------------------------------
// K14 pieces.  Mt = H * P[cols, :]  (r x n) == (P[:,cols] H^T)^T  == M_a^T of the reference.
__global__ void __launch_bounds__(256) ekf_mt_kernel(const double *__restrict__ H, int ldh, int r, int k,
                                                     const int *__restrict__ cols, const double *__restrict__ P,
                                                     int ldp, int n, double *__restrict__ Mt, int ldm,
                                                     int *__restrict__ flag, const int *__restrict__ skip) {
  if (flag && blockIdx.x == 0 && threadIdx.x == 0) {
    flag[0] = 0;  // update status word, set by the kernels that follow
  }
  if (skip && *skip == 0) return;
  const int tr_n = (r + 15) >> 4, tn_n = (n + 15) >> 4;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= tr_n * tn_n) return;
  const int tr = tile / tn_n, tn = tile - tr * tn_n;
  const int lane = threadIdx.x & 63;
  d4 acc = {0, 0, 0, 0};
  const double *Hr = H + min(tr * 16 + (lane & 15), r - 1);
  const double *Pq = P + min(tn * 16 + (lane & 15), n - 1);  // P = Pc (k x n)
  auto fa = [&](int, int kk) { return Hr[(size_t)kk * ldh]; };
  auto fb = [&](int kk, int) { return Pq[(size_t)kk * ldp]; };
  acc = mfma_tile_f64_pipe<16>(fa, fb, k, acc);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int i = tr * 16 + (lane >> 4) + 4 * q, j = tn * 16 + (lane & 15);
    if (i < r && j < n) {
      Mt[(size_t)j * ldm + i] = acc[q];
      const int pos = cols[j];  // cols = inverse map here: position of state j among the measured columns
      if (pos >= 0) Mt[(size_t)(n + 1 + pos) * ldm + i] = acc[q];  // compact copy Mt[:, cols] behind Mt and y
    }
  }
}

// S = Mt[:, cols] * H^T + diag(R)   (r x r, col-major ld = lds_)
__global__ void __launch_bounds__(256) ekf_s_kernel(const double *__restrict__ Mt, int ldm, const double *__restrict__ H,
                                                    int ldh, int r, int k, const int *__restrict__ cols,
                                                    const double *__restrict__ Rdiag, double *__restrict__ S, int lds_, const int *__restrict__ skip) {
  if (skip && *skip == 0) return;
  const int tn = (r + 15) >> 4;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= tn * tn) return;
  const int ti = tile / tn, tj = tile - ti * tn;
  const int lane = threadIdx.x & 63;
  d4 acc = {0, 0, 0, 0};
  const double *Mr = Mt + min(ti * 16 + (lane & 15), r - 1);  // Mt = compact Mt[:, cols]
  const double *Hr = H + min(tj * 16 + (lane & 15), r - 1);
  auto fa = [&](int, int kk) { return Mr[(size_t)kk * ldm]; };
  auto fb = [&](int kk, int) { return Hr[(size_t)kk * ldh]; };
  acc = mfma_tile_f64_pipe<16>(fa, fb, k, acc);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int i = ti * 16 + (lane >> 4) + 4 * q, j = tj * 16 + (lane & 15);
    if (i < r && j < r) S[(size_t)j * lds_ + i] = acc[q] + (i == j ? (Rdiag ? Rdiag[i] : 1.0) : 0.0);
  }
}

// ekf_mt_kernel and ekf_s_kernel in one launch.  The first mt_blocks workgroups form Mt = H Pc tile by tile; every further workgroup
// owns one 16-row strip of S: it first forms its strip of H Ps (= the columns `cols` of Mt, the same sums in the same order, kept in
// LDS) and multiplies it with H^T for the tiles on and above the diagonal — S no longer waits for Mt, and the compact copy of
// Mt[:, cols] is never written.
// Whitened update (dense_kernels.hip "whitened update"): workgroups from `first` on each own 16 columns b of the state and form, with
// H = M^T (Lt), "Ps" = G, g:   GP = G P[cols, b] (LDS, and stored for the C1 tiles of the next launch: blocked_chol.hip WhitenC1),
// Y0[:, b] = M^T GP,   d0[b] = P[cols, b]^T g.
struct WhitenPanels {
  const double *P;  // the covariance, both triangles valid
  int ldp, n;
  const int *cols;  // (may sit in pinned host memory: read once into LDS)
  const double *g;
  double *Y0;  // border layout of the factorisation kernels: Y0[b * k + c]
  double *GP;  // GP[b * k + c]
  double *d0;
  const int *use_m;  // device word (the prior factor's count of near-dependent pivots): 0 = this update does not take the factor form, nothing to do here
  int first;   // first workgroup of this part; < 0: none
};

__global__ void __launch_bounds__(512) ekf_ms_kernel(const double *__restrict__ H, int ldh, int r, int k, const double *__restrict__ Pc,
                                                     int ldp, int n, const double *__restrict__ Ps, double *__restrict__ Mt, int ldm,
                                                     const double *__restrict__ Rdiag, double *__restrict__ S, int lds_,
                                                     int mt_blocks, int *__restrict__ flag, const int *__restrict__ skip, int h_upper,
                                                     WhitenPanels wp) {
  // h_upper (whitened update: H = Lp^T): H(i, kk) = 0 for kk < i, so a 16-row strip's sums start at its own first column — the
  // skipped terms are exact zeros, the results the same bits, about half of the tile products gone
  __shared__ double strip[192 * 17];  // (H Ps)[16 rows][k], element (i, kk) at kk * 17 + i
  if (flag && blockIdx.x == 0 && threadIdx.x == 0) {
    flag[0] = 0;  // update status word, set by the kernels that follow
  }
  if (skip && *skip == 0) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, nw = blockDim.x >> 6;  // (4 or 8 waves)
  if ((int)blockIdx.x < mt_blocks) {
    const int tr_n = (r + 15) >> 4, tn_n = (n + 15) >> 4;
    const int tile = blockIdx.x * nw + wave;
    if (tile >= tr_n * tn_n) return;
    const int tr = tile / tn_n, tn = tile - tr * tn_n;
    d4 acc = {0, 0, 0, 0};
    const int k0 = h_upper ? min(tr * 16, k) : 0;
    const double *Hr = H + min(tr * 16 + li, r - 1) + (size_t)k0 * ldh;
    const double *Pq = Pc + min(tn * 16 + li, n - 1) + (size_t)k0 * ldp;  // Pc = P[cols, :] (k x n)
    auto fa = [&](int, int kk) { return Hr[(size_t)kk * ldh]; };
    auto fb = [&](int kk, int) { return Pq[(size_t)kk * ldp]; };
    acc = mfma_tile_f64_pipe<16>(fa, fb, k - k0, acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tr * 16 + (lane >> 4) + 4 * q, j = tn * 16 + li;
      if (i < r && j < n) Mt[(size_t)j * ldm + i] = acc[q];
    }
    return;
  }
  const int kt = (k + 15) >> 4, tr_n = (r + 15) >> 4;
  if (wp.first >= 0 && (int)blockIdx.x >= wp.first) {
    __shared__ int scols[192];
    if ((wp.use_m[0] | wp.use_m[1]) == 0) return;
    const int b0 = ((int)blockIdx.x - wp.first) * 16, bl = min(b0 + li, wp.n - 1);
    if (threadIdx.x < 192) scols[threadIdx.x] = wp.cols[min((int)threadIdx.x, k - 1)];
    __syncthreads();
    for (int tj = wave; tj < kt; tj += nw) {  // GP = G Pc[:, b]
      d4 acc = {0, 0, 0, 0};
      const double *Gr = Ps + min(tj * 16 + li, k - 1);
      auto fa = [&](int, int kk) { return Gr[(size_t)kk * k]; };
      auto fb = [&](int kk, int) { return wp.P[(size_t)scols[kk] * wp.ldp + bl]; };
      acc = mfma_tile_f64_pipe<16>(fa, fb, k, acc);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = tj * 16 + (lane >> 4) + 4 * q;
        if (c < k) strip[c * 17 + li] = acc[q];
        if (c < k && b0 + li < wp.n) wp.GP[(size_t)(b0 + li) * k + c] = acc[q];
      }
    }
    if (wave == nw - 1) {  // d0[b] = sum_c Pc(c, b) g[c]: four partial sums per column, in a fixed order
      double sum = 0.0;
      for (int c = lane >> 4; c < k; c += 4) sum += wp.P[(size_t)scols[c] * wp.ldp + bl] * wp.g[c];
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      if (lane < 16 && b0 + li < wp.n) wp.d0[b0 + li] = sum;
    }
    __syncthreads();
    for (int tj = wave; tj < kt; tj += nw) {  // Y0[:, b] = M^T GP  (M^T(c, kk) = 0 for kk < c)
      d4 acc = {0, 0, 0, 0};
      const int k0 = min(tj * 16, k);
      const double *Hr = H + min(tj * 16 + li, k - 1) + (size_t)k0 * ldh;
      const double *sp = strip + k0 * 17;
      auto fa = [&](int, int kk) { return Hr[(size_t)kk * ldh]; };
      auto fb = [&](int kk, int) { return sp[kk * 17 + li]; };
      acc = mfma_tile_f64_pipe<16>(fa, fb, k - k0, acc);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = tj * 16 + (lane >> 4) + 4 * q;
        if (c < k && b0 + li < wp.n) wp.Y0[(size_t)(b0 + li) * k + c] = acc[q];
      }
    }
    return;
  }
  const int ti = blockIdx.x - mt_blocks;
  {
    const int k0 = h_upper ? min(ti * 16, k) : 0;
    const double *Hr = H + min(ti * 16 + li, r - 1) + (size_t)k0 * ldh;
    for (int tj = wave; tj < kt; tj += nw) {
      d4 acc = {0, 0, 0, 0};
      const double *Pq = Ps + min(tj * 16 + li, k - 1) + (size_t)k0 * k;  // Ps = P[cols, cols] (k x k)
      auto fa = [&](int, int kk) { return Hr[(size_t)kk * ldh]; };
      auto fb = [&](int kk, int) { return Pq[(size_t)kk * k]; };
      acc = mfma_tile_f64_pipe<16>(fa, fb, k - k0, acc);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = tj * 16 + li;
        if (j < k) strip[j * 17 + (lane >> 4) + 4 * q] = acc[q];
      }
    }
  }
  __syncthreads();
  for (int tj = ti + wave; tj < tr_n; tj += nw) {
    d4 acc = {0, 0, 0, 0};
    const int k0 = h_upper ? min(tj * 16, k) : 0;  // (H(j, kk) = 0 for kk < j)
    const double *Hc = H + min(tj * 16 + li, r - 1) + (size_t)k0 * ldh;
    const double *sp = strip + k0 * 17;
    auto fa = [&](int, int kk) { return sp[kk * 17 + li]; };
    auto fb = [&](int kk, int) { return Hc[(size_t)kk * ldh]; };
    acc = mfma_tile_f64_pipe<16>(fa, fb, k - k0, acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = ti * 16 + (lane >> 4) + 4 * q, j = tj * 16 + li;
      if (i < r && j < r) S[(size_t)j * lds_ + i] = acc[q] + (i == j ? (Rdiag ? Rdiag[i] : 1.0) : 0.0);
    }
  }
}

// Cholesky S = L L^T from the UPPER triangle of S (REF: `S.selfadjointView<Upper>().llt()`), one
// workgroup, S staged in LDS.  Root-free right-looking elimination (one barrier per column), the
// square roots are taken at the end.  Output L (lower, col-major, ld = ldl).  flag |= 2 on a
// non-positive pivot.
__global__ void __launch_bounds__(1024) ekf_chol_kernel(const double *__restrict__ S, int lds_, int r, double *__restrict__ L,
                                                        int ldl, int *__restrict__ flag) {
  extern __shared__ double smem[];
  const int rp = r | 1;  // odd pitch: conflict-free column walks
  double *A = smem;      // A[i*rp + j], i >= j used
  for (int idx = threadIdx.x; idx < r * r; idx += blockDim.x) {
    int j = idx / r, i = idx - j * r;
    if (i >= j) A[i * rp + j] = S[(size_t)i * lds_ + j];  // lower(i,j) := upper(j,i)
  }
  __syncthreads();
  bool bad = false;
  for (int j = 0; j < r; ++j) {
    const double d = A[j * rp + j];
    if (!(d > 0.0)) bad = true;
    const double inv = 1.0 / d;
    const int nrem = r - 1 - j;
    // trailing update of the lower triangle: (i, c) with j < c <= i < r
    for (int idx = threadIdx.x; idx < nrem * nrem; idx += blockDim.x) {
      int ci = idx / nrem, ii = idx - ci * nrem;
      int c = j + 1 + ci, i = j + 1 + ii;
      if (i >= c) A[i * rp + c] -= A[i * rp + j] * A[c * rp + j] * inv;
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < r * r; idx += blockDim.x) {
    int j = idx / r, i = idx - j * r;
    double val = 0.0;
    if (i >= j) {
      double sq = sqrt(A[j * rp + j]);
      val = (i == j) ? sq : A[i * rp + j] / sq;
    }
    L[(size_t)j * ldl + i] = val;
  }
  if (bad && threadIdx.x == 0) atomicOr(flag, 2);
}

// W = L^-1 * [Mt | res]  — one wave per right-hand side, L staged in LDS (row-major), lanes split
// each row's dot product.  Column i < n of W is (K M^T) 's factor: diag(K M^T)_i = |W[:,i]|^2, so the
// reference's negative-diagonal test (StateHelper.cpp:143-152) is evaluated here: flag |= 1.
// Column n is y = L^-1 res.
__global__ void __launch_bounds__(256) ekf_trsm_kernel(const double *__restrict__ L, int ldl, int r,
                                                       const double *__restrict__ Mt, int ldm, int n,
                                                       const double *__restrict__ res, const double *__restrict__ P,
                                                       int ldp, double *__restrict__ W, int ldw, int *__restrict__ flag) {
  extern __shared__ double smem[];
  const int rp = r | 1;
  double *Ls = smem;  // Ls[q*rp + p]
  for (int idx = threadIdx.x; idx < r * r; idx += blockDim.x) {
    int p = idx / r, q = idx - p * r;
    Ls[q * rp + p] = L[(size_t)p * ldl + q];
  }
  __syncthreads();
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col > n) return;
  const int lane = threadIdx.x & 63;
  const double *b = col < n ? Mt + (size_t)col * ldm : res;
  // x distributed: lane l holds x[l], x[l+64], x[l+128], x[l+192]   (r <= 256)
  double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
  double ssq = 0.0;
  for (int q = 0; q < r; ++q) {
    double part = 0.0;
    if (lane < q) part += Ls[q * rp + lane] * x0;
    if (lane + 64 < q) part += Ls[q * rp + lane + 64] * x1;
    if (lane + 128 < q) part += Ls[q * rp + lane + 128] * x2;
    if (lane + 192 < q) part += Ls[q * rp + lane + 192] * x3;
    const double sum = wave_sum(part);
    const double xq = (b[q] - sum) / Ls[q * rp + q];
    ssq += xq * xq;
    const int slot = q >> 6, owner = q & 63;
    if (lane == owner) {
      if (slot == 0) x0 = xq;
      else if (slot == 1) x1 = xq;
      else if (slot == 2) x2 = xq;
      else x3 = xq;
      W[(size_t)col * ldw + q] = xq;
    }
  }
  if (col < n && lane == 0) {
    if (P[(size_t)col * ldp + col] - ssq < 0.0) atomicOr(flag, 1);
  }
}

// P -= W^T W (upper triangle, then mirrored: REF StateHelper.cpp:155-156) and dx = W^T y,
// only when flag == 0.  Tiles with tj >= ti; the last blocks compute dx.
__global__ void __launch_bounds__(256) ekf_apply_kernel(const double *__restrict__ W, int ldw, int r, int n,
                                                        double *__restrict__ P, int ldp, double *__restrict__ dx,
                                                        const int *__restrict__ flag) {
  if (*flag != 0) return;
  const int tn = (n + 15) >> 4;
  const int ntri = tn * (tn + 1) / 2;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wid < ntri) {
    // unrank (ti <= tj) from wid
    int ti = 0, rem = wid;
    while (rem >= tn - ti) {
      rem -= tn - ti;
      ++ti;
    }
    const int tj = ti + rem;
    d4 acc = {0, 0, 0, 0};
    auto fa = [&](int i, int kk) { int c = ti * 16 + i; return c < n ? W[(size_t)c * ldw + kk] : 0.0; };
    auto fb = [&](int kk, int j) { int c = tj * 16 + j; return c < n ? W[(size_t)c * ldw + kk] : 0.0; };
    acc = mfma_tile_f64(fa, fb, r, acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int i = ti * 16 + (lane >> 4) + 4 * q, j = tj * 16 + (lane & 15);
      if (i < n && j < n && i <= j) {
        double val = P[(size_t)j * ldp + i] - acc[q];
        P[(size_t)j * ldp + i] = val;
        P[(size_t)i * ldp + j] = val;
      }
    }
  } else {
    // dx: one wave per 64 state entries... simple: wave w handles states [w*64, w*64+64)
    const int w = wid - ntri;
    const int i = w * 64 + lane;
    if (i < n) {
      const double *y = W + (size_t)n * ldw;
      double s = 0.0;
      for (int q = 0; q < r; ++q) s += W[(size_t)i * ldw + q] * y[q];
      dx[i] = s;
    }
  }
}

// ========================================================================================== launchers
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

int gather_args(plv_ctx *ctx, const double *d_P, int n, int ldp, const int *d_cols, int k, GatherArgs &g) {
  int rc;
  if ((rc = ctx->d_Pc.reserve((size_t)k * n * 8)) || (rc = ctx->d_Ps.reserve((size_t)k * k * 8)) ||
      (rc = ctx->d_inv.reserve((size_t)n * 4)))
    return rc;
  g = GatherArgs{d_P, ldp, n, d_cols, k, ctx->d_Pc.as<double>(), ctx->d_Ps.as<double>(), ctx->d_inv.as<int>()};
  ++ctx->gather_stamp;  // the gathered blocks are about to be rewritten
  return PLV_OK;
}

// `d_P` non-null: the covariance gathers for (P, cols) ride on the same launch (n, ldp, d_cols describe them)
int launch_nullspace(plv_ctx *ctx, int F, int fdim, int k, int ld, const int *d_rows, double *d_Hf, double *d_Hx,
                     double *d_res, const double *d_P, int n, int ldp, const int *d_cols, int shift) {
  size_t shm = (size_t)(ld * (fdim + k + 1) + ld) * sizeof(double);
  if (shm > 160 * 1024) {
    set_last_error("nullspace: feature block of %zu bytes exceeds LDS", shm);
    return PLV_E_CAPACITY;
  }
  GatherArgs g{};
  int extra = 0;
  if (d_P) {
    int rc = gather_args(ctx, d_P, n, ldp, d_cols, k, g);
    if (rc) return rc;
    extra = cdiv(std::max(k * n, std::max(k * k, n)), 256);
  }
  PLV_HIP_CHECK(ensure_dyn_smem((const void *)nullspace_kernel, (int)shm));
  ProfScope ps(ctx->prof, "nullspace_kernel", ctx->stream);
  hipLaunchKernelGGL(nullspace_kernel, dim3(F + extra), dim3(256), shm, ctx->stream, fdim, k, ld, d_rows, d_Hf, d_Hx, d_res, F, g,
                     shift < 0 ? fdim : shift);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

int launch_gather_cov(plv_ctx *ctx, const double *d_P, int n, int ldp, const int *d_cols, int k) {
  GatherArgs g{};
  int rc = gather_args(ctx, d_P, n, ldp, d_cols, k, g);
  if (rc) return rc;
  ProfScope ps(ctx->prof, "gather_cov_kernel", ctx->stream);
  hipLaunchKernelGGL(gather_cov_kernel, dim3(cdiv(std::max(k * n, std::max(k * k, n)), 256)), dim3(256), 0, ctx->stream, g);
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

// chi2 of F features.  Requires launch_gather_cov for the same (P, cols) beforehand.
int launch_chi2(plv_ctx *ctx, int F, const Chi2Args &a_in, int max_mp) {
  if (max_mp > CHI2_MAXM) {
    set_last_error("chi2: %d projected rows per feature exceeds %d", max_mp, CHI2_MAXM);
    return PLV_E_CAPACITY;
  }
  Chi2Args a = a_in;
  int rc;
  if ((rc = ctx->d_T.reserve_units((size_t)F, (size_t)std::max(ctx->cfg.num_features, 64), (size_t)a.k * a.ld * 8))) return rc;  // (sized for the peak pool: DevBuf::reserve_units)
  a.F = F;
  a.Ps = ctx->d_Ps.as<double>();
  a.T = ctx->d_T.as<double>();
  const int mt = (max_mp + 15) / 16, kt = (a.k + 15) / 16;
  {
    ProfScope ps(ctx->prof, "chi2_t_kernel", ctx->stream);
    hipLaunchKernelGGL(chi2_t_kernel, dim3(cdiv(F * mt * kt, 4)), dim3(256), 0, ctx->stream, a, mt);
  }
  size_t shm = (size_t)(65 * 65 + 8) * sizeof(double);
  ProfScope ps(ctx->prof, "chi2_gate_kernel", ctx->stream);
  // static LDS of blocked_chol (48 KB) + the dynamic S tile exceed the 64 KB default
  if (max_mp <= 32) {
    PLV_HIP_CHECK(ensure_dyn_smem((const void *)chi2_gate_kernel<2>, (int)shm));
    hipLaunchKernelGGL(chi2_gate_kernel<2>, dim3(F), dim3(256), shm, ctx->stream, a);
  } else {
    PLV_HIP_CHECK(ensure_dyn_smem((const void *)chi2_gate_kernel<4>, (int)shm));
    hipLaunchKernelGGL(chi2_gate_kernel<4>, dim3(F), dim3(320), shm, ctx->stream, a);
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

// Reduces the m x nc col-major matrix at d_A (lda) to an nc x nc upper triangle with
// non-negative diagonal at d_out (ld = nc... returned in *ld_out).  d_A and d_tmp are ping-pong
// workspaces, both overwritten.  Returns the device pointer holding the result in *result.
int launch_tsqr(plv_ctx *ctx, double *d_A, int lda, int m, int nc, double *d_tmp, size_t tmp_elems, double **result,
                int *ld_out, const int *m_dev) {
  // (round 6b) blocked, in place, one workgroup per 2048 rows (PLV_KNOB_TSQR_TREE: the tree of rounds 1-6a); a stack less than one
  // and a half times as tall as wide stays with the tree: one pass of one workgroup there, 269 against 322 us at 120 x 105
  {
    const int cap = HQ_T * 4, nb = m <= HQ_T * 2 ? 1 : (m + cap - 1) / cap;
    const int ldv = ((nb == 1 ? m : cap) + 63) & ~63;  // the panels' reflectors, explicit (16 columns), live in the workspace
    const size_t v_elems = (size_t)nb * ldv * 16, r_elems = nb > 1 ? (size_t)nb * nc * nc : 0;
    const int ldv2 = (nb * nc + 63) & ~63;
    const size_t need = v_elems + r_elems + (nb > 1 ? (size_t)ldv2 * 16 : 0);
    const bool tree_holds = ((size_t)nc * nc + QR_CHUNK + 8) * sizeof(double) <= 160 * 1024 && nc * QR_PARTS <= 1024;  // (its R lives in LDS: ~140 columns)
    if ((2 * m >= 3 * nc || m_dev || !tree_holds) && m >= nc && nc <= 208 && nb <= 8 && need <= tmp_elems && !(plv::knob(plv::PLV_KNOB_TSQR_TREE) && tree_holds)) {
      ProfScope ps(ctx->prof, "hqr_kernel", ctx->stream);
      double *Vg = d_tmp, *Rst = d_tmp + v_elems, *Vg2 = Rst + r_elems;
      const int ldro = nb * nc;
      if (nb == 1 && m <= HQ_T * 2)
        hipLaunchKernelGGL(hqr_kernel<2>, dim3(1), dim3(HQ_T), 0, ctx->stream, d_A, lda, m, nc, 1, Vg, ldv, m_dev, (double *)nullptr, 0, (const int *)nullptr, 0, 0);
      else
        hipLaunchKernelGGL(hqr_kernel<4>, dim3(nb), dim3(HQ_T), 0, ctx->stream, d_A, lda, m, nc, nb == 1 ? 1 : 2, Vg, ldv, m_dev, nb > 1 ? Rst : (double *)nullptr,
                           ldro, (const int *)nullptr, 0, 0);
      if (nb > 1) {  // the stacked R's (nb * nc rows: at most 8 x 208); returns at once when the rows fitted the first launch's first workgroup
        if (ldro <= HQ_T * 2)
          hipLaunchKernelGGL(hqr_kernel<2>, dim3(1), dim3(HQ_T), 0, ctx->stream, Rst, ldro, ldro, nc, 1, Vg2, ldv2, (const int *)nullptr, (double *)nullptr, 0, m_dev, m, cap);
        else
          hipLaunchKernelGGL(hqr_kernel<4>, dim3(1), dim3(HQ_T), 0, ctx->stream, Rst, ldro, ldro, nc, 1, Vg2, ldv2, (const int *)nullptr, (double *)nullptr, 0, m_dev, m, cap);
      }
      PLV_HIP_CHECK(hipGetLastError());
      *result = nb > 1 ? Rst : d_A;
      *ld_out = nb > 1 ? ldro : lda;
      return PLV_OK;
    }
    // (else the tree below: it sizes its launches on the host and takes the bound m; rows beyond *m_dev are zeros)
  }
  size_t shm = ((size_t)nc * nc + QR_CHUNK + 8) * sizeof(double);
  if (shm > 160 * 1024 || nc * QR_PARTS > 1024) {
    set_last_error("tsqr: %d columns exceed the LDS-resident R capacity", nc);
    return PLV_E_CAPACITY;
  }
  PLV_HIP_CHECK(ensure_dyn_smem((const void *)qr_accum_kernel, (int)shm));
  int threads = cdiv(nc * QR_PARTS, 64) * 64;
  double *src = d_A, *dst = d_tmp;
  size_t cap_src = (size_t)lda * nc, cap_dst = tmp_elems;
  int src_ld = lda, rows = m;
  // rows per workgroup: a multiple of the chunk, at least 2*nc so every level at least halves
  int rpw = cdiv(2 * nc, QR_CHUNK) * QR_CHUNK;
  for (;;) {
    int W = cdiv(rows, rpw);
    if (W < 1) W = 1;
    int ldo = W * nc;
    if ((size_t)ldo * nc > cap_dst) {
      set_last_error("tsqr: workspace too small (%d x %d > %zu)", ldo, nc, cap_dst);
      return PLV_E_CAPACITY;
    }
    {
      ProfScope ps(ctx->prof, "qr_accum_kernel", ctx->stream);
      hipLaunchKernelGGL(qr_accum_kernel, dim3(W), dim3(threads), shm, ctx->stream, src, src_ld, rows, nc, rpw, dst, ldo,
                         W == 1 ? 1 : 0);
    }
    PLV_HIP_CHECK(hipGetLastError());
    if (W == 1) {
      *result = dst;
      *ld_out = ldo;
      return PLV_OK;
    }
    rows = W * nc;
    src_ld = ldo;
    double *t = src;
    src = dst;
    dst = t;
    size_t tc = cap_src;
    cap_src = cap_dst;
    cap_dst = tc;
  }
}

// Mt = H P[cols,:] and S = Mt[:,cols] H^T + R in one launch (S: tiles on and above the diagonal).
void launch_ekf_ms(plv_ctx *ctx, const double *d_P, int n, int ldp, const double *d_H, int r, int k, int ldh,
                   const int *d_cols, const double *d_Rdiag, double *Mt, int ldm, double *S, bool gathered, int *d_flag) {
  if (!gathered) (void)launch_gather_cov(ctx, d_P, n, ldp, d_cols, k);
  ProfScope ps(ctx->prof, "ekf_ms_kernel", ctx->stream);
  const int mt_blocks = cdiv(cdiv(r, 16) * cdiv(n, 16), 4);
  if (k > 192) {  // (the strip buffer of the fused kernel; launch_ekf's callers stay below: PLV_E_CAPACITY earlier)
    hipLaunchKernelGGL(ekf_mt_kernel, dim3(mt_blocks), dim3(256), 0, ctx->stream, d_H, ldh, r, k, ctx->d_inv.as<int>(),
                       ctx->d_Pc.as<double>(), n, n, Mt, ldm, d_flag, ctx->skip_word);
    hipLaunchKernelGGL(ekf_s_kernel, dim3(cdiv(cdiv(r, 16) * cdiv(r, 16), 4)), dim3(256), 0, ctx->stream, Mt + (size_t)(n + 1) * ldm, ldm,
                       d_H, ldh, r, k, d_cols, d_Rdiag, S, r, ctx->skip_word);
    return;
  }
  hipLaunchKernelGGL(ekf_ms_kernel, dim3(mt_blocks + cdiv(r, 16)), dim3(256), 0, ctx->stream, d_H, ldh, r, k, ctx->d_Pc.as<double>(), n, n,
                     ctx->d_Ps.as<double>(), Mt, ldm, d_Rdiag, S, r, mt_blocks, d_flag, ctx->skip_word, 0, WhitenPanels{nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, -1});
}

// Whitened route: B = M^T G M + I (k x k, upper tiles) and c = M^T g — ekf_ms_kernel with H := M^T (Lt), "Ps" := G (full
// symmetric) and "Pc" := g as a k x 1 block — and, in further workgroups of the same launch, the products with the covariance
// columns (WhitenPanels): GP = G P[cols, :], Y0 = M^T GP, d0 = P[:, cols] g.
void launch_whiten_b(plv_ctx *ctx, const double *Lt, int k, const double *Gs, const double *gv, double *cv, double *B, int *d_flag,
                     const double *d_P, int ldp, int n, const int *d_cols, double *Y0, double *GP, double *d0, const int *use_m) {
  ProfScope ps(ctx->prof, "ekf_ms_kernel", ctx->stream);
  const int mt_blocks = cdiv(cdiv(k, 16), 8), first = mt_blocks + cdiv(k, 16);  // eight waves a workgroup: one round of tiles per phase up to k = 128
  hipLaunchKernelGGL(ekf_ms_kernel, dim3(first + cdiv(n, 16)), dim3(512), 0, ctx->stream, Lt, k, k, k, gv, 1, 1, Gs, cv, k,
                     (const double *)nullptr, B, k, mt_blocks, d_flag, ctx->skip_word, 1, WhitenPanels{d_P, ldp, n, d_cols, gv, Y0, GP, d0, use_m, first});
}

// The EKF kernels on device-resident operands.  d_P is n x n (ldp).  On return *d_flag holds
// 0 (updated), bit0 (negative diagonal), bit1 (S not positive definite).
int launch_ekf(plv_ctx *ctx, double *d_P, int n, int ldp, const double *d_H, int r, int k, int ldh, const int *d_cols,
               const double *d_res, const double *d_Rdiag, double *d_dx, int *d_flag, bool gathered) {
  int rc;
  const int ldm = r, ldw = r;
  if ((rc = ctx->d_Mt.reserve((size_t)r * (n + 1 + k) * 8)) || (rc = ctx->d_S.reserve((size_t)r * r * 8 * 2)) ||
      (rc = ctx->d_W.reserve((size_t)r * (n + 1) * 8)))
    return rc;
  double *Mt = ctx->d_Mt.as<double>(), *S = ctx->d_S.as<double>(), *L = S + (size_t)r * r, *W = ctx->d_W.as<double>();
  size_t shm = (size_t)r * (r | 1) * sizeof(double);
  if (shm > 160 * 1024) {
    set_last_error("ekf: r=%d exceeds the LDS-resident Cholesky capacity", r);
    return PLV_E_CAPACITY;
  }
  launch_ekf_ms(ctx, d_P, n, ldp, d_H, r, k, ldh, d_cols, d_Rdiag, Mt, ldm, S, gathered, d_flag);
  PLV_HIP_CHECK(ensure_dyn_smem((const void *)ekf_chol_kernel, (int)shm));
  PLV_HIP_CHECK(ensure_dyn_smem((const void *)ekf_trsm_kernel, (int)shm));
  {
    ProfScope ps(ctx->prof, "ekf_chol_kernel", ctx->stream);
    hipLaunchKernelGGL(ekf_chol_kernel, dim3(1), dim3(1024), shm, ctx->stream, S, r, r, L, r, d_flag);
  }
  {
    ProfScope ps(ctx->prof, "ekf_trsm_kernel", ctx->stream);
    hipLaunchKernelGGL(ekf_trsm_kernel, dim3(cdiv(n + 1, 4)), dim3(256), shm, ctx->stream, L, r, r, Mt, ldm, n, d_res, d_P,
                       ldp, W, ldw, d_flag);
  }
  {
    ProfScope ps(ctx->prof, "ekf_apply_kernel", ctx->stream);
    int tn = cdiv(n, 16);
    int waves = tn * (tn + 1) / 2 + cdiv(n, 64);
    hipLaunchKernelGGL(ekf_apply_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, ctx->stream, W, ldw, r, n, d_P, ldp, d_dx,
                       d_flag);
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

}  // namespace plv
