// slam_api.hip — in-state landmarks (a30): update, delayed initialisation, marginalisation on the
// device-resident covariance.
//   UpdaterCamera::slam_update / slam_init          REF: PL-VIWO/src/update/cam/UpdaterCamera.cpp:296-369
//   StateHelper::initialize / initialize_invertible REF: PL-VIWO/src/state/StateHelper.cpp:357-439, 495-600
//   StateHelper::marginalize                        REF: StateHelper.cpp:235-303
// Inactive at the reference's shipped configuration (max_slam: 0), therefore built for correctness, one
// landmark per call as the reference loops: the Givens split reuses nullspace_kernel (rows kept), the gate
// and the EKF step reuse the MSCKF kernels; new here are the state augmentation and the marginalisation,
// which write a second covariance buffer that is then swapped in.
#include <algorithm>
#include <vector>

#include "update_kernels.hpp"
#include "update_state.hpp"

using namespace plv;

#define TRY(expr)                  \
  do {                             \
    int _rc = (expr);              \
    if (_rc != PLV_OK) return _rc; \
  } while (0)

namespace {

struct Inv3 {
  double a[9];
  bool ok;
};
// 3x3 inverse, Gauss-Jordan with partial pivoting (the oracle's `inverse`; Eigen's dynamic inverse() is a
// partially pivoted LU as well)
__device__ Inv3 inverse3(const double *A) {
  double M[9], I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) M[i] = A[i];  // row-major [row*3 + col]
  Inv3 out;
  out.ok = true;
  for (int col = 0; col < 3; ++col) {
    int piv = col;
    double best = fabs(M[col * 3 + col]);
    for (int i = col + 1; i < 3; ++i)
      if (fabs(M[i * 3 + col]) > best) best = fabs(M[i * 3 + col]), piv = i;
    if (!(best > 0.0)) out.ok = false;
    if (piv != col)
      for (int j = 0; j < 3; ++j) {
        double t = M[piv * 3 + j];
        M[piv * 3 + j] = M[col * 3 + j];
        M[col * 3 + j] = t;
        t = I[piv * 3 + j];
        I[piv * 3 + j] = I[col * 3 + j];
        I[col * 3 + j] = t;
      }
    const double d = 1.0 / M[col * 3 + col];
    for (int j = 0; j < 3; ++j) M[col * 3 + j] *= d, I[col * 3 + j] *= d;
    for (int i = 0; i < 3; ++i) {
      if (i == col) continue;
      const double f = M[i * 3 + col];
      if (f == 0.0) continue;
      for (int j = 0; j < 3; ++j) M[i * 3 + j] -= f * M[col * 3 + j], I[i * 3 + j] -= f * I[col * 3 + j];
    }
  }
  for (int i = 0; i < 9; ++i) out.a[i] = I[i];
  return out;
}

// StateHelper::initialize_invertible for a 3-dof variable appended at index n.
// Hx / Hf / res: the Givens-rotated system (col-major, ld), rows 0..2 are the initialising rows.
// out[0..2] = H_L^-1 res, flag = 1 when the reference rejects (:572-587).  Pn is (n+3) x (n+3), ld = n+3.
__global__ void __launch_bounds__(256) cov_init_invertible_kernel(const double *__restrict__ P, int n, const int *__restrict__ cols,
                                                                  int k, const double *__restrict__ Hx, const double *__restrict__ Hf,
                                                                  const double *__restrict__ res, int ld, double *__restrict__ Pn,
                                                                  double *__restrict__ out, int *__restrict__ flag) {
  extern __shared__ double sm[];
  double *Ma = sm;           // [n][3]
  double *sc = sm + 3 * n;   // M[9] | HLinv[9] | PLL[9] | v[3] | ok
  const int t = threadIdx.x;
  for (int i = t; i < n; i += blockDim.x) {  // M_a = P[:, cols] Hinit^T   (:546-556)
    double a0 = 0, a1 = 0, a2 = 0;
    for (int j = 0; j < k; ++j) {
      const double p = P[(size_t)cols[j] * n + i];
      a0 += p * Hx[(size_t)j * ld + 0];
      a1 += p * Hx[(size_t)j * ld + 1];
      a2 += p * Hx[(size_t)j * ld + 2];
    }
    Ma[3 * i] = a0;
    Ma[3 * i + 1] = a1;
    Ma[3 * i + 2] = a2;
  }
  __syncthreads();
  if (t < 9) {  // M = Hinit P_s Hinit^T + R, R = I   (:560-564)
    const int q = t / 3, q2 = t - 3 * q;
    double s = q == q2 ? 1.0 : 0.0;
    for (int j = 0; j < k; ++j) s += Hx[(size_t)j * ld + q] * Ma[3 * cols[j] + q2];
    sc[t] = s;
  }
  __syncthreads();
  if (t == 0) {
    double *M = sc, *HLinv = sc + 9, *PLL = sc + 18, *v = sc + 27;
    M[3] = M[1];  // selfadjointView<Upper>
    M[6] = M[2];
    M[7] = M[5];
    double HL[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) HL[i * 3 + j] = Hf[(size_t)j * ld + i];
    const Inv3 hi = inverse3(HL);
    bool ok = hi.ok;
    for (int i = 0; i < 9; ++i) HLinv[i] = hi.a[i];
    double T[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) T[i * 3 + j] = HLinv[i * 3] * M[j] + HLinv[i * 3 + 1] * M[3 + j] + HLinv[i * 3 + 2] * M[6 + j];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) PLL[i * 3 + j] = T[i * 3] * HLinv[j * 3] + T[i * 3 + 1] * HLinv[j * 3 + 1] + T[i * 3 + 2] * HLinv[j * 3 + 2];
    for (int i = 0; i < 3; ++i) v[i] = HLinv[i * 3] * res[0] + HLinv[i * 3 + 1] * res[1] + HLinv[i * 3 + 2] * res[2];
    const Inv3 pi = inverse3(PLL);
    ok = ok && pi.ok;
    double chi = 0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) chi += v[i] * pi.a[i * 3 + j] * v[j];
    const double dn = sqrt(PLL[0] * PLL[0] + PLL[4] * PLL[4] + PLL[8] * PLL[8]);
    if (!(chi >= 1e-7) || dn > 1000) ok = false;       // :572-578 (a NaN chi is rejected too)
    if (PLL[0] < 0.0 || PLL[4] < 0.0 || PLL[8] < 0.0) ok = false;  // :581-587
    sc[30] = ok ? 1.0 : 0.0;
    *flag = ok ? 0 : 1;
    out[0] = v[0];
    out[1] = v[1];
    out[2] = v[2];
  }
  __syncthreads();
  if (sc[30] == 0.0) return;
  const int n2 = n + 3;
  const double *HLinv = sc + 9, *PLL = sc + 18;
  for (int idx = t; idx < n * n; idx += blockDim.x) {  // :590-594
    const int j = idx / n, i = idx - j * n;
    Pn[(size_t)j * n2 + i] = P[idx];
  }
  for (int idx = t; idx < 3 * n; idx += blockDim.x) {
    const int q = idx / n, i = idx - q * n;
    const double c = -(Ma[3 * i] * HLinv[q * 3] + Ma[3 * i + 1] * HLinv[q * 3 + 1] + Ma[3 * i + 2] * HLinv[q * 3 + 2]);  // -M_a H_L^-T
    Pn[(size_t)(n + q) * n2 + i] = c;
    Pn[(size_t)i * n2 + n + q] = c;
  }
  if (t < 9) Pn[(size_t)(n + t % 3) * n2 + n + t / 3] = PLL[(t / 3) * 3 + t % 3];
}

__global__ void __launch_bounds__(256) cov_marginalize_kernel(const double *__restrict__ P, int n, int id, int size,
                                                              double *__restrict__ Pn) {
  const int m = n - size;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < m * m; idx += gridDim.x * blockDim.x) {
    const int j = idx / m, i = idx - j * m;
    const int si = i < id ? i : i + size, sj = j < id ? j : j + size;
    Pn[idx] = P[(size_t)sj * n + si];
  }
}

}  // namespace

extern "C" {

int plv_cov_marginalize(plv_ctx *ctx, int id, int size) {
  if (!ctx || ctx->cov_n < 1 || id < 0 || size < 1 || id + size > ctx->cov_n) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  const int n = ctx->cov_n, m = n - size;
  if (m < 1) return PLV_E_BADARG;
  TRY(ctx->d_P2.reserve((size_t)m * m * 8));
  hipLaunchKernelGGL(cov_marginalize_kernel, dim3(std::min(64, (m * m + 255) / 256)), dim3(256), 0, ctx->stream,
                     ctx->d_P.as<double>(), n, id, size, ctx->d_P2.as<double>());
  PLV_HIP_CHECK(hipGetLastError());
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  std::swap(ctx->d_P, ctx->d_P2);
  ctx->cov_n = m;
  ++ctx->gather_stamp;
  return PLV_OK;
}

int plv_slam_update(plv_ctx *ctx, int rows, int k, int ld, const double *H, const double *res, const int *col_to_state,
                    double chi2_mult, uint8_t *accepted, double *dx) {
  if (!ctx || !H || !res || !col_to_state || !accepted || !dx || rows < 1 || k < 1 || ld < rows || ctx->cov_n < 1)
    return PLV_E_BADARG;
  const int n = ctx->cov_n;
  *accepted = 0;
  std::fill(dx, dx + n, 0.0);
  if (rows < 2) return PLV_OK;  // REF :316-319
  double chi = 0.0;
  TRY(plv_chi2_batch(ctx, nullptr, n, n, 1, k, ld, &rows, H, res, col_to_state, 1.0, &chi));  // R = I (whitened)
  if (!(chi < chi2_mult * plv_chi2_quantile95(rows))) return PLV_OK;                         // REF :331
  const int rc = plv_ekf_update(ctx, nullptr, n, n, H, rows, k, ld, col_to_state, res, nullptr, dx);
  if (rc == PLV_OK) *accepted = 1;
  return rc;
}

int plv_slam_initialize(plv_ctx *ctx, int rows, int k, int ld, const double *Hf, const double *Hx, const double *res,
                        const int *col_to_state, double chi2_mult, uint8_t *ok, double *dx_init, double *dx) {
  if (!ctx || !Hf || !Hx || !res || !col_to_state || !ok || !dx_init || !dx || rows < 4 || k < 1 || ld < rows || ctx->cov_n < 1)
    return PLV_E_BADARG;  // REF :354-357 needs more than one measurement
  (void)hipSetDevice(ctx->device);
  const int n = ctx->cov_n, f = 3;
  *ok = 0;
  std::fill(dx, dx + n + f, 0.0);
  std::fill(dx_init, dx_init + f, 0.0);
  for (int j = 0; j < k; ++j)
    if (col_to_state[j] < 0 || col_to_state[j] >= n) return PLV_E_BADARG;
  // ---- Givens split on the device, all rows kept (StateHelper.cpp:391)
  const size_t nHf = (size_t)f * ld, nHx = (size_t)k * ld;
  TRY(ctx->d_fHf.reserve((nHf + nHx + ld) * 8));
  TRY(ctx->d_frows.reserve(4));
  TRY(ctx->d_cols.reserve((size_t)k * 4));
  double *dHf = ctx->d_fHf.as<double>(), *dHx = dHf + nHf, *dres = dHx + nHx;
  PLV_HIP_CHECK(plv::memcpy_async(dHf, Hf, nHf * 8, hipMemcpyHostToDevice, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(dHx, Hx, nHx * 8, hipMemcpyHostToDevice, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(dres, res, (size_t)ld * 8, hipMemcpyHostToDevice, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(ctx->d_frows.p, &rows, 4, hipMemcpyHostToDevice, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(ctx->d_cols.p, col_to_state, (size_t)k * 4, hipMemcpyHostToDevice, ctx->stream));
  TRY(launch_nullspace(ctx, 1, f, k, ld, ctx->d_frows.as<int>(), dHf, dHx, dres, nullptr, 0, 0, nullptr, /*shift*/ 0));
  std::vector<double> hHx(nHx), hres(ld);
  PLV_HIP_CHECK(plv::memcpy_async(hHx.data(), dHx, nHx * 8, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(hres.data(), dres, (size_t)ld * 8, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  // ---- Mahalanobis gate on the updating rows; the threshold uses ALL rows (:409-424)
  const int mup = rows - f;
  if (mup > 0) {
    double chi = 0.0;
    TRY(plv_chi2_batch(ctx, nullptr, n, n, 1, k, ld, &mup, hHx.data() + f, hres.data() + f, col_to_state, 1.0, &chi));
    if (!(chi <= chi2_mult * plv_chi2_quantile95(rows))) return PLV_OK;  // rejected: nothing changed
  }
  // ---- initialize_invertible: augment into the second buffer
  const int n2 = n + f;
  TRY(ctx->d_P2.reserve((size_t)n2 * n2 * 8));
  TRY(ctx->d_dx.reserve(64));
  TRY(ctx->d_flag.reserve(16));
  // (plv_chi2_batch re-staged Hx from row f on: the rotated full system is still in dHf / dHx / dres)
  PLV_HIP_CHECK(plv::memcpy_async(ctx->d_cols.p, col_to_state, (size_t)k * 4, hipMemcpyHostToDevice, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(dHx, hHx.data(), nHx * 8, hipMemcpyHostToDevice, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(dres, hres.data(), (size_t)ld * 8, hipMemcpyHostToDevice, ctx->stream));
  const size_t shm = (size_t)(3 * n + 32) * sizeof(double);
  hipLaunchKernelGGL(cov_init_invertible_kernel, dim3(1), dim3(256), shm, ctx->stream, ctx->d_P.as<double>(), n, ctx->d_cols.as<int>(),
                     k, dHx, dHf, dres, ld, ctx->d_P2.as<double>(), ctx->d_dx.as<double>(), ctx->d_flag.as<int>());
  PLV_HIP_CHECK(hipGetLastError());
  int flag = 1;
  double v[3];
  PLV_HIP_CHECK(plv::memcpy_async(&flag, ctx->d_flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(v, ctx->d_dx.p, 24, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  if (flag != 0) return PLV_OK;
  std::swap(ctx->d_P, ctx->d_P2);  // the old covariance stays intact in d_P2 until the update below succeeded
  ctx->cov_n = n2;
  ++ctx->gather_stamp;
  // ---- EKFUpdate with the updating rows; failure reverts the initialisation (:430-435)
  if (mup > 0) {
    const int rc = plv_ekf_update(ctx, nullptr, n2, n2, hHx.data() + f, mup, k, ld, col_to_state, hres.data() + f, nullptr, dx);
    if (rc != PLV_OK) {
      std::swap(ctx->d_P, ctx->d_P2);
      ctx->cov_n = n;
      std::fill(dx, dx + n2, 0.0);
      return rc == PLV_E_NOT_PSD ? PLV_OK : rc;
    }
  }
  std::copy(v, v + 3, dx_init);
  *ok = 1;
  return PLV_OK;
}

}  // extern "C"
