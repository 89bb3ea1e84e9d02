// dense_kernels.hip — second-generation kernels of the EKF-update half (fp64): low sequential depth.
//
// Measured on MI355X (profiles/r01): the Householder TSQR of update_kernels.hip is latency-bound —
// 99 reflector steps x 4 row chunks x 4 tree levels ~ 1600 dependent steps of ~1.1 us.  Here the
// compression (REF: StateHelper::measurement_compress_inplace, PL/state/StateHelper.cpp:602-614)
// is re-expressed so that only ONE chain of k dependent steps remains:
//     G = [H r]^T [H r]             gram_kernel      (v_mfma_f64_16x16x4_f64, split over rows)
//     G = [R z]^T [R z]             chol_compress_kernel (one workgroup, LDS-resident)
// R is the same upper-triangular factor the reference's Givens QR produces (unique for full
// column rank, diag >= 0).  Columns whose pivot vanishes after equilibration (the gauge directions
// an MSCKF Jacobian cannot observe) give a zero row, which is what an exact QR gives as well.
// DESIGN.md "Compression numerics" has the error analysis (backward error eps*|G| either way).
//
// The EKF step (REF: StateHelper::EKFUpdate, StateHelper.cpp:94-173) uses the same elimination
// with an identity border to get L^-1 in the same k steps, then three tile-parallel products:
//     W = L^-1 [Mt | res],  dC = W^T W (= K M^T),  dx = W^T y,  commit (diag test, P -= dC).
#include "update_kernels.hpp"
#include "wave_ops.hpp"

namespace plv {

typedef double d4 __attribute__((ext_vector_type(4)));

// 16x16 fp64 MFMA tile with the operand loads of four k-steps issued ahead of the MFMAs.
template <class FA, class FB>
__device__ __forceinline__ d4 mfma_tile_f64_p(FA a, FB b, int K, d4 acc) {
  const int lane = threadIdx.x & 63;
  const int ij = lane & 15, kq = lane >> 4;
  int k0 = 0;
  for (; k0 + 16 <= K; k0 += 16) {
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

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------
// Gp[s] (nc x nc, col-major) = A[rows of split s]^T A[rows of split s], upper tiles only.
__global__ void __launch_bounds__(256) gram_kernel(const double *__restrict__ A, int lda, int m, int nc, int nsplit,
                                                   int rows_per_split, double *__restrict__ Gp) {
  const int nt = (nc + 15) >> 4;
  const int ntri = nt * (nt + 1) / 2;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= ntri * nsplit) return;
  const int s = w / ntri;
  int rem = w - s * ntri, ti = 0;
  while (rem >= nt - ti) {
    rem -= nt - ti;
    ++ti;
  }
  const int tj = ti + rem;
  const int row0 = s * rows_per_split;
  const int nrows = max(0, min(m, row0 + rows_per_split) - row0);
  const int lane = threadIdx.x & 63;
  const double *Ar = A + row0;
  d4 acc = {0, 0, 0, 0};
  auto fa = [&](int i, int kk) { int c = ti * 16 + i; return c < nc ? Ar[(size_t)c * lda + kk] : 0.0; };
  auto fb = [&](int kk, int j) { int c = tj * 16 + j; return c < nc ? Ar[(size_t)c * lda + kk] : 0.0; };
  acc = mfma_tile_f64_p(fa, fb, nrows, acc);
  double *G = Gp + (size_t)s * nc * nc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int i = ti * 16 + (lane >> 4) + 4 * q, j = tj * 16 + (lane & 15);
    if (i < nc && j < nc) G[(size_t)j * nc + i] = acc[q];
  }
}

// ------------------------------------------------------------------------------------------
// Register-resident root-free right-looking elimination (Cholesky / LDL^T) of a symmetric k x k
// matrix with border rows appended.  Thread (tr, tc) of a TR x 16 grid permanently owns the
// elements (i, c) = (tr + u*TR, tc + w*16), u < EL_MAXR, w < EL_MAXC, in registers.  Step j: the
// owners of column j publish it through a double-buffered LDS vector (ONE barrier per step), every
// thread reads its <= 4 row multipliers and <= 8 pivot-column entries and updates its own block —
// no read-modify-write on LDS.  The j loop is unrolled over the 16-column register slot, so every
// register index is static, and work that can never be live is removed at compile time:
//   LAYOUT 0 (compress):  rows [0,k) symmetric part, ONE border row parked at row 4*TR-1;
//   LAYOUT 1 (inverse):   rows [0,k) symmetric part (k <= 2*TR), identity border row b at 2*TR+b.
// Row slot u (rows [u*TR,(u+1)*TR)) of the symmetric part only has columns c <= i, is dead once the
// pivot has passed it; identity-border row b is zero left of column b and wakes up at step b.
// After step j: column j holds l_ij*l_jj, pivs[j] = l_jj^2; pivots <= tau mark the column dead
// (skipped: its row of the factor is zero).  Elements above the diagonal are junk, never read.
// Measured (MI355X): the first LDS read-modify-write version cost ~1.6 us per step; this one is
// bound by one barrier + two LDS round trips per step.
#define EL_MAXR 4
#define EL_MAXC 8
struct ElimLds {
  double *colbuf;  // [2][4*TR]
  double *pivs;    // [k]
  unsigned char *dead;  // [k]
  int rows_pad;
};
template <int TR, int LAYOUT> struct ElimMap {
  // can register slot (u,w) ever hold a live element
  static constexpr __host__ __device__ bool pair_live(int u, int w) {
    if (LAYOUT == 0) {
      if (u == EL_MAXR - 1) return true;  // holds the border row (and symmetric rows)
      return 16 * w <= u * TR + TR - 1;
    } else {
      if (u < 2) return 16 * w <= u * TR + TR - 1;
      return 16 * w + 15 >= (u - 2) * TR;  // border row b >= (u-2)*TR lives in columns >= b
    }
  }
};
template <int TR, int LAYOUT>
__device__ __forceinline__ bool eliminate_regs(double (&val)[EL_MAXR][EL_MAXC], const ElimLds &L, int k) {
  typedef ElimMap<TR, LAYOUT> M;
  const int tr = threadIdx.x >> 4, tc = threadIdx.x & 15;
  bool any_dead = false;
#pragma unroll
  for (int wj = 0; wj < EL_MAXC; ++wj) {
    if (wj * 16 < k) {
      for (int jj = 0; jj < 16; ++jj) {
        const int j = wj * 16 + jj;
        if (j < k) {
          double *cb = L.colbuf + (j & 1) * L.rows_pad;
          if (tc == jj) {
#pragma unroll
            for (int u = 0; u < EL_MAXR; ++u)
              if (M::pair_live(u, wj)) cb[tr + u * TR] = val[u][wj];
          }
          __syncthreads();
          const double piv = cb[j];
          const bool isdead = !(piv > L.pivs[-1]);  // pivs[-1] holds tau
          any_dead |= isdead;
          if (threadIdx.x == 0) {
            L.pivs[j] = piv;
            L.dead[j] = isdead ? 1 : 0;
          }
          if (!isdead) {
            const double ninv = -1.0 / piv;
            double a[EL_MAXC];
#pragma unroll
            for (int w = 0; w < EL_MAXC; ++w)
              if (w >= wj) a[w] = cb[min(tc + w * 16, 4 * TR - 1)];  // column j entry of row c
#pragma unroll
            for (int u = 0; u < EL_MAXR; ++u) {
              // uniform (scalar) liveness of the whole row slot at this step
              bool slot_live;
              if (LAYOUT == 0)
                slot_live = (u == EL_MAXR - 1) || (j + 1 < (u + 1) * TR);
              else
                slot_live = u < 2 ? (j + 1 < (u + 1) * TR) : (j >= (u - 2) * TR);
              if (slot_live) {
                const double f = cb[tr + u * TR] * ninv;
#pragma unroll
                for (int w = 0; w < EL_MAXC; ++w) {
                  if (w >= wj && M::pair_live(u, w)) {  // compile-time
                    const double nv = fma(f, a[w], val[u][w]);
                    val[u][w] = ((w > wj) || (tc > jj)) ? nv : val[u][w];
                  }
                }
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  return any_dead;
}
__host__ __device__ inline size_t eliminate_lds_bytes(int TR, int k, int stage_elems) {
  return (size_t)(8 * TR + 8 + 2 * k) * sizeof(double) + (size_t)((k + 63) & ~63) + (size_t)stage_elems * sizeof(double);
}
// colbuf | tau | pivs[k] | extra[k] | dead[k] | stage...
__device__ __forceinline__ ElimLds eliminate_carve(double *smem, int TR, int k, double tau, double **extra, double **stage) {
  ElimLds L;
  L.rows_pad = 4 * TR;
  L.colbuf = smem;
  L.pivs = smem + 8 * TR + 8;
  if (threadIdx.x == 0) L.pivs[-1] = tau;
  *extra = L.pivs + k;
  L.dead = reinterpret_cast<unsigned char *>(*extra + k);
  *stage = reinterpret_cast<double *>(L.dead + ((k + 63) & ~63));
  return L;
}

// [R z] from the Gram partial sums: Cholesky of the equilibrated G with the residual column as
// the border row (parked at row 4*TR-1).  One workgroup of TR x 16 threads; k <= min(128, 4*TR-1).
template <int TR>
__global__ void __launch_bounds__(TR * 16) chol_compress_kernel(const double *__restrict__ Gp, int nsplit, int nc,
                                                                double *__restrict__ R, int ldr, double *__restrict__ z) {
  extern __shared__ double smem[];
  const int k = nc - 1;
  const int BR = EL_MAXR * TR - 1;  // border row slot
  double *sc, *Gs;
  // pivots of the unit-diagonal matrix lie in [0,1]; below tau a column is numerically dependent
  ElimLds L = eliminate_carve(smem, TR, k, 64.0 * 2.220446049250313e-16 * (double)nc, &sc, &Gs);
  const int t = threadIdx.x, T = TR * 16, tr = t >> 4, tc = t & 15;
  const size_t gsz = (size_t)nc * nc;
  // stage G = sum of the split partial sums in LDS (coalesced, independent loads)
  for (int idx = t; idx < nc * nc; idx += T) {
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    int sp = 0;
    for (; sp + 4 <= nsplit; sp += 4) {
      v0 += Gp[(sp + 0) * gsz + idx];
      v1 += Gp[(sp + 1) * gsz + idx];
      v2 += Gp[(sp + 2) * gsz + idx];
      v3 += Gp[(sp + 3) * gsz + idx];
    }
    for (; sp < nsplit; ++sp) v0 += Gp[sp * gsz + idx];
    Gs[idx] = (v0 + v1) + (v2 + v3);
  }
  __syncthreads();
  for (int j = t; j < k; j += T) {  // column scales D = diag(G)^-1/2
    const double d = Gs[(size_t)j * nc + j];
    sc[j] = d > 0.0 ? 1.0 / sqrt(d) : 0.0;
  }
  __syncthreads();
  // element (row i, col c) of the lower triangle = upper(c, i) = Gs[i*nc + c]; G row k -> row BR
  double val[EL_MAXR][EL_MAXC];
#pragma unroll
  for (int u = 0; u < EL_MAXR; ++u)
#pragma unroll
    for (int w = 0; w < EL_MAXC; ++w) {
      const int i = tr + u * TR, c = tc + w * 16;
      const int gi = (i == BR) ? k : i;
      const bool ok = c < k && ((i < k && c <= i) || i == BR);
      const double g = Gs[ok ? gi * nc + c : 0];
      val[u][w] = ok ? g * sc[c] * (i < k ? sc[i] : 1.0) : 0.0;
    }
  eliminate_regs<TR, 0>(val, L, k);
  // R[c][i] = l_ic / s_i, z_c = border_c / l_cc   (dead column c: zero row)
#pragma unroll
  for (int u = 0; u < EL_MAXR; ++u)
#pragma unroll
    for (int w = 0; w < EL_MAXC; ++w) {
      const int i = tr + u * TR, c = tc + w * 16;
      if (c < k && (i < k || i == BR)) {
        const bool live = !L.dead[c];
        const double lcc = live ? sqrt(L.pivs[c]) : 1.0;
        if (i < k) {
          double out = 0.0;
          if (c <= i && live && sc[i] > 0.0) out = (i == c ? lcc : val[u][w] / lcc) / sc[i];
          if (c <= i) R[(size_t)i * ldr + c] = out;  // (row c, col i) of R
          if (c < i) R[(size_t)c * ldr + i] = 0.0;   // strictly lower part
        } else {
          z[c] = live ? val[u][w] / lcc : 0.0;
        }
      }
    }
}

// L^-1 of S (upper triangle valid, REF: `S.selfadjointView<Upper>().llt()`) through an identity
// border (row b parked at 2*TR + b).  r <= 2*TR.  flag |= 2 when S is not positive definite.
template <int TR>
__global__ void __launch_bounds__(TR * 16) chol_inv_kernel(const double *__restrict__ S, int lds_, int r,
                                                           double *__restrict__ Linv, int ldl, int *__restrict__ flag) {
  extern __shared__ double smem[];
  double *unused, *Ss;
  ElimLds L = eliminate_carve(smem, TR, r, 0.0, &unused, &Ss);
  const int t = threadIdx.x, T = TR * 16, tr = t >> 4, tc = t & 15;
  for (int idx = t; idx < r * r; idx += T) {
    int c = idx / r, rr = idx - c * r;
    Ss[idx] = S[(size_t)c * lds_ + rr];
  }
  __syncthreads();
  double val[EL_MAXR][EL_MAXC];
#pragma unroll
  for (int u = 0; u < EL_MAXR; ++u)
#pragma unroll
    for (int w = 0; w < EL_MAXC; ++w) {
      const int i = tr + u * TR, c = tc + w * 16;
      const bool sym = c < r && i < r && c <= i;
      const double g = Ss[sym ? i * r + c : 0];  // lower(i,c) := upper(c,i) = element (row c, col i)
      double v = sym ? g : 0.0;
      if (c < r && i >= 2 * TR && i - 2 * TR == c) v = 1.0;
      val[u][w] = v;
    }
  const bool bad = eliminate_regs<TR, 1>(val, L, r);
#pragma unroll
  for (int u = 2; u < EL_MAXR; ++u)
#pragma unroll
    for (int w = 0; w < EL_MAXC; ++w) {
      const int b = tr + (u - 2) * TR, c = tc + w * 16;
      if (c < r && b < r)  // Linv(c, b) = border[b][c] / l_cc, zero above the diagonal
        Linv[(size_t)b * ldl + c] = (c >= b) ? val[u][w] / sqrt(L.pivs[c]) : 0.0;
    }
  if (bad && t == 0) atomicOr(flag, 2);
}

// W = Linv * [Mt | res]   (r x (n+1)); one wave per 16x16 tile; Linv is lower triangular so the
// contraction for row tile tq stops at column 16*(tq+1).
__global__ void __launch_bounds__(256) ekf_w_kernel(const double *__restrict__ Linv, int ldl, int r,
                                                    const double *__restrict__ Mt, int ldm, int n,
                                                    const double *__restrict__ res, double *__restrict__ W, int ldw) {
  const int tq_n = (r + 15) >> 4, tc_n = (n + 1 + 15) >> 4;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= tq_n * tc_n) return;
  const int tq = w / tc_n, tc = w - tq * tc_n;
  const int lane = threadIdx.x & 63;
  const int K = min(r, 16 * (tq + 1));
  d4 acc = {0, 0, 0, 0};
  auto fa = [&](int i, int kk) { int q = tq * 16 + i; return q < r ? Linv[(size_t)kk * ldl + q] : 0.0; };
  auto fb = [&](int kk, int j) {
    int c = tc * 16 + j;
    return c < n ? Mt[(size_t)c * ldm + kk] : (c == n ? res[kk] : 0.0);
  };
  acc = mfma_tile_f64_p(fa, fb, K, acc);
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    int q = tq * 16 + (lane >> 4) + 4 * q4, c = tc * 16 + (lane & 15);
    if (q < r && c <= n) W[(size_t)c * ldw + q] = acc[q4];
  }
}

// dC = W[:, :n]^T W[:, :n] (upper tiles, = K M^T of the reference) and dx = W[:, :n]^T y.
__global__ void __launch_bounds__(256) ekf_dc_kernel(const double *__restrict__ W, int ldw, int r, int n,
                                                     double *__restrict__ dC, int ldc, double *__restrict__ dx) {
  const int tn = (n + 15) >> 4;
  const int ntri = tn * (tn + 1) / 2;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wid < ntri) {
    int ti = 0, rem = wid;
    while (rem >= tn - ti) {
      rem -= tn - ti;
      ++ti;
    }
    const int tj = ti + rem;
    d4 acc = {0, 0, 0, 0};
    auto fa = [&](int i, int kk) { int c = ti * 16 + i; return c < n ? W[(size_t)c * ldw + kk] : 0.0; };
    auto fb = [&](int kk, int j) { int c = tj * 16 + j; return c < n ? W[(size_t)c * ldw + kk] : 0.0; };
    acc = mfma_tile_f64_p(fa, fb, r, acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int i = ti * 16 + (lane >> 4) + 4 * q, j = tj * 16 + (lane & 15);
      if (i < n && j < n) dC[(size_t)j * ldc + i] = acc[q];
    }
  } else {
    const int i = (wid - ntri) * 64 + lane;
    if (i < n) {
      const double *y = W + (size_t)n * ldw;
      double s = 0.0;
      for (int q = 0; q < r; ++q) s += W[(size_t)i * ldw + q] * y[q];
      dx[i] = s;
    }
  }
}

// REF: StateHelper.cpp:143-156 — reject when any P_ii - dC_ii < 0 (nothing modified), else
// P.upper -= dC, mirrored.  Every workgroup re-evaluates the (n-entry) test; block 0 publishes it.
__global__ void __launch_bounds__(256) ekf_commit_kernel(double *__restrict__ P, int ldp, int n,
                                                         const double *__restrict__ dC, int ldc, int *__restrict__ flag) {
  __shared__ int neg;
  if (threadIdx.x == 0) neg = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    if (P[(size_t)i * ldp + i] - dC[(size_t)i * ldc + i] < 0.0) neg = 1;
  __syncthreads();
  const int f = *flag;
  if (neg || (f & 2)) {
    if (neg && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(flag, 1);
    return;
  }
  // off-diagonal commit only: the diagonal (which every block reads for the test above) is
  // committed by ekf_commit_diag_kernel, launched after this kernel.
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n * n; idx += gridDim.x * blockDim.x) {
    int j = idx / n, i = idx - j * n;
    if (i < j) {
      double v = P[(size_t)j * ldp + i] - dC[(size_t)j * ldc + i];
      P[(size_t)j * ldp + i] = v;
      P[(size_t)i * ldp + j] = v;
    }
  }
}
__global__ void __launch_bounds__(256) ekf_commit_diag_kernel(double *__restrict__ P, int ldp, int n,
                                                              const double *__restrict__ dC, int ldc,
                                                              const int *__restrict__ flag) {
  if (*flag != 0) return;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) P[(size_t)i * ldp + i] -= dC[(size_t)i * ldc + i];
}

// ========================================================================================== launchers
// Compression of the stacked m x nc matrix [H | r] (col-major, lda) into R (k x k upper, ldr) and z.
int launch_gram_compress(plv_ctx *ctx, const double *d_A, int lda, int m, int nc, double *d_Gp, size_t gp_elems,
                         double *d_R, int ldr, double *d_z) {
  const int k = nc - 1;
  if (k > 128) {
    set_last_error("gram compress: %d columns exceed the register-resident factorisation (128)", k);
    return PLV_E_CAPACITY;
  }
  const int TR = k <= 63 ? 16 : (k <= 127 ? 32 : 64);
  size_t shm = eliminate_lds_bytes(TR, k, nc * nc);
  int nsplit = std::min(8, std::max(1, m / 128));
  while ((size_t)nsplit * nc * nc > gp_elems && nsplit > 1) --nsplit;
  if ((size_t)nsplit * nc * nc > gp_elems || shm > 160 * 1024) return PLV_E_CAPACITY;
  int rps = cdiv(cdiv(m, nsplit), 4) * 4;
  const int nt = cdiv(nc, 16);
  {
    ProfScope ps(ctx->prof, "gram_kernel", ctx->stream);
    int waves = nt * (nt + 1) / 2 * nsplit;
    hipLaunchKernelGGL(gram_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, ctx->stream, d_A, lda, m, nc, nsplit, rps, d_Gp);
  }
  {
    ProfScope ps(ctx->prof, "chol_compress_kernel", ctx->stream);
    if (TR == 16) {
      PLV_HIP_CHECK(hipFuncSetAttribute((const void *)chol_compress_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      hipLaunchKernelGGL(chol_compress_kernel<16>, dim3(1), dim3(256), shm, ctx->stream, d_Gp, nsplit, nc, d_R, ldr, d_z);
    } else if (TR == 32) {
      PLV_HIP_CHECK(hipFuncSetAttribute((const void *)chol_compress_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      hipLaunchKernelGGL(chol_compress_kernel<32>, dim3(1), dim3(512), shm, ctx->stream, d_Gp, nsplit, nc, d_R, ldr, d_z);
    } else {
      PLV_HIP_CHECK(hipFuncSetAttribute((const void *)chol_compress_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      hipLaunchKernelGGL(chol_compress_kernel<64>, dim3(1), dim3(1024), shm, ctx->stream, d_Gp, nsplit, nc, d_R, ldr, d_z);
    }
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

bool ekf_fast_fits(int r) { return r <= 128 && eliminate_lds_bytes(64, r, r * r) <= 160 * 1024; }

// EKF update with the identity-border Cholesky.  Requires ekf_fast_fits(r).
int launch_ekf_fast(plv_ctx *ctx, double *d_P, int n, int ldp, const double *d_H, int r, int k, int ldh, const int *d_cols,
                    const double *d_res, const double *d_Rdiag, double *d_dx, int *d_flag) {
  int rc;
  const int ldm = r, ldw = r;
  if ((rc = ctx->d_Mt.reserve((size_t)r * n * 8)) || (rc = ctx->d_S.reserve((size_t)r * r * 8 * 2)) ||
      (rc = ctx->d_W.reserve((size_t)r * (n + 1) * 8)) || (rc = ctx->d_y.reserve((size_t)n * n * 8)))
    return rc;
  double *Mt = ctx->d_Mt.as<double>(), *S = ctx->d_S.as<double>(), *Linv = S + (size_t)r * r, *W = ctx->d_W.as<double>(),
         *dC = ctx->d_y.as<double>();
  const int TR = r <= 32 ? 16 : (r <= 64 ? 32 : 64);
  size_t shm = eliminate_lds_bytes(TR, r, r * r);
  PLV_HIP_CHECK(hipMemsetAsync(d_flag, 0, sizeof(int), ctx->stream));
  launch_ekf_ms(ctx, d_P, n, ldp, d_H, r, k, ldh, d_cols, d_Rdiag, Mt, ldm, S);
  {
    ProfScope ps(ctx->prof, "chol_inv_kernel", ctx->stream);
    if (TR == 16) {
      PLV_HIP_CHECK(hipFuncSetAttribute((const void *)chol_inv_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      hipLaunchKernelGGL(chol_inv_kernel<16>, dim3(1), dim3(256), shm, ctx->stream, S, r, r, Linv, r, d_flag);
    } else if (TR == 32) {
      PLV_HIP_CHECK(hipFuncSetAttribute((const void *)chol_inv_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      hipLaunchKernelGGL(chol_inv_kernel<32>, dim3(1), dim3(512), shm, ctx->stream, S, r, r, Linv, r, d_flag);
    } else {
      PLV_HIP_CHECK(hipFuncSetAttribute((const void *)chol_inv_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      hipLaunchKernelGGL(chol_inv_kernel<64>, dim3(1), dim3(1024), shm, ctx->stream, S, r, r, Linv, r, d_flag);
    }
  }
  {
    ProfScope ps(ctx->prof, "ekf_w_kernel", ctx->stream);
    int waves = cdiv(r, 16) * cdiv(n + 1, 16);
    hipLaunchKernelGGL(ekf_w_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, ctx->stream, Linv, r, r, Mt, ldm, n, d_res, W, ldw);
  }
  {
    ProfScope ps(ctx->prof, "ekf_dc_kernel", ctx->stream);
    int tn = cdiv(n, 16);
    int waves = tn * (tn + 1) / 2 + cdiv(n, 64);
    hipLaunchKernelGGL(ekf_dc_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, ctx->stream, W, ldw, r, n, dC, n, d_dx);
  }
  {
    ProfScope ps(ctx->prof, "ekf_commit_kernel", ctx->stream);
    hipLaunchKernelGGL(ekf_commit_kernel, dim3(std::min(64, cdiv(n * n, 256))), dim3(256), 0, ctx->stream, d_P, ldp, n, dC, n,
                       d_flag);
    hipLaunchKernelGGL(ekf_commit_diag_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, d_P, ldp, n, dC, n, d_flag);
  }
  PLV_HIP_CHECK(hipGetLastError());
  return PLV_OK;
}

}  // namespace plv
