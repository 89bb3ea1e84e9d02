// dense_kernels.hip — tile-parallel products of the EKF-update half (fp64) and the launchers that chain
// them with the blocked factorisations of blocked_chol.hip.
//
// Measured on MI355X (profiles/r01): a Householder TSQR of the stacked Jacobian is latency-bound
// (99 reflector steps x 4 row chunks x 4 tree levels ~ 1600 dependent steps).  The compression
// (REF: StateHelper::measurement_compress_inplace, PL/state/StateHelper.cpp:602-614) is therefore
// re-expressed so that only ONE chain of k dependent pivots remains:
//     G = [H r]^T [H r]             gram_kernel            (v_mfma_f64_16x16x4_f64, rows split over waves)
//     G = [R z]^T [R z]             bchol_compress_kernel  (blocked Cholesky, blocked_chol.hip)
// R is the same upper-triangular factor the reference's Givens QR produces (unique for full
// column rank, diag >= 0).  Columns whose pivot vanishes after equilibration (the gauge directions
// an MSCKF Jacobian cannot observe) give a zero row, which is what an exact QR gives as well.
// DESIGN.md "Compression numerics" has the error analysis (backward error eps*|G| either way).
//
// The EKF step (REF: StateHelper::EKFUpdate, StateHelper.cpp:94-173):
//     Mt = H P[cols,:], S = Mt[:,cols] H^T + R      ekf_mt_kernel, ekf_s_kernel (update_kernels.hip)
//     W  = L^-1 [Mt | res],  S = L L^T               bchol_ekf_kernel
//     dC = W^T W (= K M^T),  dx = W^T y              ekf_dc_kernel
//     diagonal test, P -= dC                         ekf_commit_kernel
#include "mfma_tile.hpp"
#include "update_kernels.hpp"
#include "wave_ops.hpp"

namespace plv {


static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------
// G (nc x nc, col-major, upper tiles) = A^T A.  One workgroup per upper tile; its GRAM_WAVES waves
// split the rows and the partial tiles are summed through LDS in a fixed order (deterministic).
#define GRAM_WAVES 16
__global__ void __launch_bounds__(64 * GRAM_WAVES) gram_kernel(const double *__restrict__ A, int lda, int m, int nc,
                                                                double *__restrict__ G) {
  __shared__ double part[GRAM_WAVES][4][64];
  const int nt = (nc + 15) >> 4;
  int rem = blockIdx.x, ti = 0;
  while (rem >= nt - ti) {
    rem -= nt - ti;
    ++ti;
  }
  const int tj = ti + rem;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rows_per = ((m + GRAM_WAVES - 1) / GRAM_WAVES + 3) & ~3;
  const int row0 = wave * rows_per;
  const int nrows = max(0, min(m, row0 + rows_per) - row0);
  const double *Ai = A + (size_t)min(ti * 16 + (lane & 15), nc - 1) * lda + row0;  // columns beyond nc only feed
  const double *Aj = A + (size_t)min(tj * 16 + (lane & 15), nc - 1) * lda + row0;  // entries that are not stored
  d4 acc = {0, 0, 0, 0};
  auto fa = [&](int, int kk) { return Ai[kk]; };
  auto fb = [&](int kk, int) { return Aj[kk]; };
  acc = mfma_tile_f64_pipe<8>(fa, fb, nrows, acc);
#pragma unroll
  for (int q = 0; q < 4; ++q) part[wave][q][lane] = acc[q];
  __syncthreads();
  if (wave < 4) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < GRAM_WAVES; ++w) s += part[w][wave][lane];
    const int i = ti * 16 + (lane >> 4) + 4 * wave, j = tj * 16 + (lane & 15);
    if (i < nc && j < nc) G[(size_t)j * nc + i] = s;
  }
}

// ------------------------------------------------------------------------------------------
// [dC | dx] = W[:, :n]^T [W[:, :n] | y]  (upper tiles; dC = K M^T of the reference, y = W[:, n]).
__global__ void __launch_bounds__(256) ekf_dc_kernel(const double *__restrict__ W, int ldw, int r, int n,
                                                     double *__restrict__ dC, int ldc, double *__restrict__ dx) {
  const int tn = (n + 1 + 15) >> 4;
  const int ntri = tn * (tn + 1) / 2;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wid >= ntri) return;
  int ti = 0, rem = wid;
  while (rem >= tn - ti) {
    rem -= tn - ti;
    ++ti;
  }
  const int tj = ti + rem;
  const double *Wi = W + (size_t)min(ti * 16 + (lane & 15), n) * ldw;
  const double *Wj = W + (size_t)min(tj * 16 + (lane & 15), n) * ldw;
  d4 acc = {0, 0, 0, 0};
  auto fa = [&](int, int kk) { return Wi[kk]; };
  auto fb = [&](int kk, int) { return Wj[kk]; };
  acc = mfma_tile_f64_pipe<16>(fa, fb, r, acc);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = ti * 16 + (lane >> 4) + 4 * q, j = tj * 16 + (lane & 15);
    if (i < n && j < n) dC[(size_t)j * ldc + i] = acc[q];
    if (i < n && j == n) dx[i] = acc[q];
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
int launch_gram_compress(plv_ctx *ctx, const double *d_A, int lda, int m, int nc, double *d_G, size_t g_elems,
                         double *d_R, int ldr, double *d_z) {
  const int k = nc - 1;
  if (k > 128 || (size_t)nc * nc > g_elems) {
    set_last_error("gram compress: %d columns exceed the register-resident factorisation (128)", k);
    return PLV_E_CAPACITY;
  }
  const int nt = cdiv(nc, 16);
  {
    ProfScope ps(ctx->prof, "gram_kernel", ctx->stream);
    hipLaunchKernelGGL(gram_kernel, dim3(nt * (nt + 1) / 2), dim3(64 * GRAM_WAVES), 0, ctx->stream, d_A, lda, m, nc, d_G);
  }
  return launch_bchol_compress(ctx, d_G, nc, d_R, ldr, d_z);
}

bool ekf_fast_fits(int r) { return r <= 128; }

// EKF update with the identity-border Cholesky.  Requires ekf_fast_fits(r).
int launch_ekf_fast(plv_ctx *ctx, double *d_P, int n, int ldp, const double *d_H, int r, int k, int ldh, const int *d_cols,
                    const double *d_res, const double *d_Rdiag, double *d_dx, int *d_flag, bool gathered) {
  int rc;
  const int ldm = r, ldw = r;
  if ((rc = ctx->d_Mt.reserve((size_t)r * (n + 1 + k) * 8)) || (rc = ctx->d_S.reserve((size_t)r * r * 8 * 2)) ||
      (rc = ctx->d_W.reserve((size_t)r * (n + 1) * 8)) || (rc = ctx->d_y.reserve((size_t)n * n * 8)))
    return rc;
  double *Mt = ctx->d_Mt.as<double>(), *S = ctx->d_S.as<double>(), *W = ctx->d_W.as<double>(), *dC = ctx->d_y.as<double>();
  launch_ekf_ms(ctx, d_P, n, ldp, d_H, r, k, ldh, d_cols, d_Rdiag, Mt, ldm, S, gathered, d_flag);
  if ((rc = launch_bchol_ekf(ctx, S, r, r, Mt, ldm, n, d_res, W, ldw, d_flag))) return rc;
  {
    ProfScope ps(ctx->prof, "ekf_dc_kernel", ctx->stream);
    int tn = cdiv(n + 1, 16);
    int waves = tn * (tn + 1) / 2;
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
