// nullspace_core.hpp — the Givens null-space projection of one feature's [Hf | Hx | res] block held row-major in LDS, shared by
// nullspace_kernel (update_kernels.hip) and the fused jacobian_nullspace_kernel (jacobian_kernels.hip); plus the dense covariance
// gathers that ride on either launch.   REF: PL-VIWO/src/state/StateHelper.cpp:616-651
#pragma once
#include <hip/hip_runtime.h>

#include "wave_ops.hpp"

namespace plv {

// Same rotation on the latency-critical path of the nullspace kernel: the chain of rows-1 dependent
// rotations per pivot column is what bounds that kernel, and an IEEE divide + sqrt + divide is ~42
// dependent fp64 instructions.  v_rcp_f64 / v_rsq_f64 + two Newton steps each give the same values
// to within 1-2 ulp in ~17 (the rotation stays orthonormal to rounding: c^2 + s^2 = 1 +- 2 eps).
__device__ __forceinline__ double rcp_newton(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}
__device__ __forceinline__ double rsqrt_newton(double x) {
  double r = __builtin_amdgcn_rsq(x);
  double e = fma(-x * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  e = fma(-x * r, r, 1.0);
  return fma(0.5 * r, e, r);
}
__device__ __forceinline__ void make_givens_fast(double p, double q, double &c, double &s) {
  if (q == 0.0) {
    c = p < 0.0 ? -1.0 : 1.0;
    s = 0.0;
  } else if (p == 0.0) {
    c = 0.0;
    s = q < 0.0 ? 1.0 : -1.0;
  } else if (fabs(p) > fabs(q)) {
    const double t = q * rcp_newton(p);
    double iu = rsqrt_newton(fma(t, t, 1.0));
    if (p < 0.0) iu = -iu;
    c = iu;
    s = -t * c;
  } else {
    const double t = p * rcp_newton(q);
    double iu = rsqrt_newton(fma(t, t, 1.0));
    if (q < 0.0) iu = -iu;
    s = -iu;
    c = -t * s;
  }
}

// Dense gathers of the covariance blocks the update contracts with, so that no MFMA operand load goes
// through a dependent index load:  Pc = P[cols, :] (k x n, row-major), Ps = P[cols, cols] (k x k), inv[state] =
// position of that state in cols or -1.  256 threads per block; runs as its own launch or as extra
// blocks of nullspace_kernel (independent work, one launch less on the update stream).
struct GatherArgs {
  const double *P;
  int ldp, n;
  const int *cols;
  int k;
  double *Pc, *Ps;
  int *inv;
};
__device__ __forceinline__ void gather_cov_block(const GatherArgs &g, int block) {
  const int idx = block * 256 + threadIdx.x;
  const int n = g.n, k = g.k;
  if (idx < k * n) {
    const int kk = idx / n, j = idx - kk * n;
    g.Pc[idx] = g.P[(size_t)g.cols[kk] * g.ldp + j];  // P symmetric: row cols[kk] == column cols[kk]
  }
  if (idx < k * k) {
    const int kk = idx / k, c = idx - kk * k;
    g.Ps[idx] = g.P[(size_t)g.cols[kk] * g.ldp + g.cols[c]];
  }
  if (idx < n) {
    int pos = -1;
    for (int q = 0; q < k; ++q) pos = (g.cols[q] == idx) ? q : pos;
    g.inv[idx] = pos;
  }
}

// X [rows][ncol] row-major in LDS, piv [rows] scratch.  For pivot column n the rotation sequence m = rows-1 .. n+1 is the
// reference's; (c, s) are recomputed by every thread from a read-only copy of the pivot column, which reproduces bit-for-bit
// what the column's owner computes, so threads never wait on each other inside a pass.  Called by the whole workgroup.
__device__ __forceinline__ void nullspace_rotate(double *X, double *piv, int rows, int ncol, int fdim) {
  for (int n = 0; n < fdim; ++n) {
    __syncthreads();
    for (int i = threadIdx.x; i < rows; i += blockDim.x) piv[i] = X[i * ncol + n];
    __syncthreads();
    for (int j = n + threadIdx.x; j < ncol; j += blockDim.x) {
      double carry_p = piv[rows - 1];
      double carry_o = X[(rows - 1) * ncol + j];
      for (int m = rows - 1; m > n; --m) {
        const double p = piv[m - 1];
        const double q = carry_p;
        const double up = X[(m - 1) * ncol + j];
        if (q == 0.0) {  // REF: `if (A(m, n) == 0.0) continue;`
          X[m * ncol + j] = carry_o;
          carry_p = p;
          carry_o = up;
          continue;
        }
        double c, s;
        make_givens_fast(p, q, c, s);
        carry_p = c * p - s * q;
        const double nu = c * up - s * carry_o;
        const double nl = s * up + c * carry_o;
        X[m * ncol + j] = (j == n) ? 0.0 : nl;  // REF: `A(m, n) = 0;`
        carry_o = nu;
      }
      X[n * ncol + j] = carry_o;
    }
  }
  __syncthreads();
}

// The same projection by three Householder reflections instead of fdim * (rows - 1) dependent Givens rotations.  Q^T [Hf] = [R; 0]
// either way, so rows fdim.. of [Hx | res] span the same left null space of Hf: every quantity the update forms from them (the
// norm of the projected residual, chi2 = r'^T S^-1 r', the Gram matrix H'^T H' of the stacked system and with it the compressed
// R up to row signs, dx, P) is invariant to the choice of orthonormal basis and agrees with the Givens route to rounding.  The
// dependent chain drops from ~87 rotations (rcp + rsq + Newton each) to three passes of ~2 * rows fused multiply-adds per
// column: 14 us -> ~1 us in the fused kernel.  (plv_nullspace_batch, whose OUTPUT is the projected block itself, keeps the
// reference's rotation order.)
__device__ __forceinline__ void nullspace_householder(double *X, double *piv, int rows, int ncol, int fdim) {
  for (int n = 0; n < fdim; ++n) {
    __syncthreads();
    for (int i = threadIdx.x; i < rows; i += blockDim.x) piv[i] = X[i * ncol + n];
    __syncthreads();
    // (the row loops below are unrolled by eight with the LDS reads of a batch issued ahead of its dependent fma chain; the order of
    // the additions is unchanged.  Measured inside jacobian_nullspace_kernel at 32 rows: three reflections 9.4 -> 8.6 us — the
    // dependent fp64 chains (norm, sqrt, divide, dot product), not the LDS latency, are what a reflection costs)
    double nrm2 = 0.0;  // every thread forms the reflector from the shared copy of column n: the same bits everywhere
    {
      int i = n;
      for (; i + 8 <= rows; i += 8) {
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = piv[i + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) nrm2 = fma(a[u], a[u], nrm2);
      }
      for (; i < rows; ++i) nrm2 = fma(piv[i], piv[i], nrm2);
    }
    if (nrm2 == 0.0) continue;  // (uniform) nothing to eliminate
    const double x0 = piv[n];
    const double nrm = sqrt(nrm2);
    const double alpha = x0 >= 0.0 ? -nrm : nrm;  // v = x - alpha e1 without cancellation
    const double v0 = x0 - alpha;
    const double tau = 1.0 / (nrm2 - alpha * x0);  // 2 / v^T v
    for (int j = n + threadIdx.x; j < ncol; j += blockDim.x) {
      if (j == n) {
        X[n * ncol + n] = alpha;
        for (int i = n + 1; i < rows; ++i) X[i * ncol + n] = 0.0;
        continue;
      }
      double w = v0 * X[n * ncol + j];
      {
        int i = n + 1;
        for (; i + 8 <= rows; i += 8) {
          double a[8], b[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] = piv[i + u], b[u] = X[(i + u) * ncol + j];
#pragma unroll
          for (int u = 0; u < 8; ++u) w = fma(a[u], b[u], w);
        }
        for (; i < rows; ++i) w = fma(piv[i], X[i * ncol + j], w);
      }
      w *= tau;
      X[n * ncol + j] = fma(-v0, w, X[n * ncol + j]);
      {
        int i = n + 1;
        for (; i + 8 <= rows; i += 8) {
          double a[8], b[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] = piv[i + u], b[u] = X[(i + u) * ncol + j];
#pragma unroll
          for (int u = 0; u < 8; ++u) X[(i + u) * ncol + j] = fma(-a[u], w, b[u]);
        }
        for (; i < rows; ++i) X[i * ncol + j] = fma(-piv[i], w, X[i * ncol + j]);
      }
    }
  }
  __syncthreads();
}

// The Householder projection in compact form (round 4).  nullspace_householder spends a reflection on: every thread forming the
// reflector from the pivot column (a chain over the rows, sqrt, divide), a dot product over the rows, an update pass, two barriers —
// 2.6 us each at 30 rows, three (points) or six (lines) in sequence.  Here wave 0 factors the rows x FD panel Hf alone, one lane
// per row with DPP wave sums (the FD reflectors v_n, tau_n and their Gram matrix G), and every other column is then projected in ONE
// pass: d_n = v_n . x for all n at once (FD independent chains), w_n = tau_n (d_n - sum_{b<n} G_nb w_b) — what the reflections
// applied one after the other would have used — and x -= sum_n v_n w_n.  The same orthogonal transformation up to rounding.
// scratch: nullspace_wy_scratch(ld, FD) doubles.  rows <= 64 (one lane per row), else the reflection-by-reflection form.
__host__ __device__ inline int nullspace_wy_scratch(int ld, int fd) { return ld * fd + fd + fd * fd + 2; }
__device__ __forceinline__ double ns_readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
template <int FD> __device__ __forceinline__ void nullspace_householder_wy(double *X, double *scr, int rows, int ncol) {
  if (rows > 64) {
    nullspace_householder(X, scr, rows, ncol, FD);
    return;
  }
  double *V = scr, *tau = scr + rows * FD, *G = tau + FD;  // V [rows][FD], G [FD][FD] (entries a > b)
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    double a[FD], v[FD], t[FD];
#pragma unroll
    for (int c = 0; c < FD; ++c) a[c] = lane < rows ? X[lane * ncol + c] : 0.0;
#pragma unroll
    for (int n = 0; n < FD; ++n) {
      const double xn = lane >= n ? a[n] : 0.0;
      const double nrm2 = wave_sum_f64(xn * xn);
      v[n] = 0.0, t[n] = 0.0;
      if (nrm2 == 0.0) continue;  // (uniform) nothing to eliminate
      const double x0 = ns_readlane_f64(a[n], n);
      const double nrm = sqrt(nrm2);
      const double alpha = x0 >= 0.0 ? -nrm : nrm;  // v = x - alpha e1 without cancellation
      t[n] = 1.0 / (nrm2 - alpha * x0);             // 2 / v^T v
      v[n] = lane == n ? x0 - alpha : xn;
      if (lane < n) v[n] = 0.0;
      double w[FD];
#pragma unroll
      for (int c = n + 1; c < FD; ++c) w[c] = wave_sum_f64(v[n] * a[c]);
#pragma unroll
      for (int c = n + 1; c < FD; ++c) a[c] = fma(-v[n], t[n] * w[c], a[c]);
      a[n] = lane == n ? alpha : (lane > n ? 0.0 : a[n]);
    }
    double gm[FD * FD];
#pragma unroll
    for (int p = 1; p < FD; ++p)
#pragma unroll
      for (int q = 0; q < p; ++q) gm[p * FD + q] = wave_sum_f64(v[p] * v[q]);
    if (lane < rows) {
#pragma unroll
      for (int c = 0; c < FD; ++c) X[lane * ncol + c] = a[c], V[lane * FD + c] = v[c];
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < FD; ++c) tau[c] = t[c];
#pragma unroll
      for (int p = 1; p < FD; ++p)
#pragma unroll
        for (int q = 0; q < p; ++q) G[p * FD + q] = gm[p * FD + q];
    }
  }
  __syncthreads();
  for (int j = FD + threadIdx.x; j < ncol; j += blockDim.x) {
    double d[FD];
#pragma unroll
    for (int c = 0; c < FD; ++c) d[c] = 0.0;
    int i = 0;
    for (; i + 4 <= rows; i += 4) {
      double x[4], vv[4][FD];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x[u] = X[(i + u) * ncol + j];
#pragma unroll
        for (int c = 0; c < FD; ++c) vv[u][c] = V[(i + u) * FD + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < FD; ++c) d[c] = fma(vv[u][c], x[u], d[c]);
    }
    for (; i < rows; ++i) {
      const double x = X[i * ncol + j];
#pragma unroll
      for (int c = 0; c < FD; ++c) d[c] = fma(V[i * FD + c], x, d[c]);
    }
    double w[FD];
#pragma unroll
    for (int c = 0; c < FD; ++c) {
      double sacc = d[c];
#pragma unroll
      for (int b = 0; b < c; ++b) sacc = fma(-G[c * FD + b], w[b], sacc);
      w[c] = tau[c] * sacc;
    }
    i = 0;
    for (; i + 4 <= rows; i += 4) {
      double x[4], vv[4][FD];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x[u] = X[(i + u) * ncol + j];
#pragma unroll
        for (int c = 0; c < FD; ++c) vv[u][c] = V[(i + u) * FD + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int c = 0; c < FD; ++c) x[u] = fma(-vv[u][c], w[c], x[u]);
        X[(i + u) * ncol + j] = x[u];
      }
    }
    for (; i < rows; ++i) {
      double x = X[i * ncol + j];
#pragma unroll
      for (int c = 0; c < FD; ++c) x = fma(-V[i * FD + c], w[c], x);
      X[i * ncol + j] = x;
    }
  }
  __syncthreads();
}

}  // namespace plv
