// ORACLE — TEST INFRASTRUCTURE ONLY (see dense.h).  PARITY UNPINNED.
//
// CPU fp64 restatement of the EKF-update half of the hot path:
//   Givens_Rotation / nullspace_project_inplace / measurement_compress_inplace
//       REF: PL-VIWO/src/state/StateHelper.cpp:602-672
//   get_marginal_covariance                REF: StateHelper.cpp:466-493
//   UpdaterStatistics::get_chi2            REF: PL-VIWO/src/update/UpdaterStatistics.cpp:94-117
//   StateHelper::EKFUpdate                 REF: StateHelper.cpp:94-173
//   UpdaterCamera::msckf_update (from the nullspace projection on)
//                                          REF: PL-VIWO/src/update/cam/UpdaterCamera.cpp:230-293
// Eigen's JacobiRotation::makeGivens / applyOnTheLeft (Eigen 3, Jacobi/Jacobi.h) are restated
// from their published real-scalar algorithm; Eigen itself is not in /root/reference.
#include "dense.h"
#include <cstdint>
#include <cstdio>

using namespace orc;

namespace {

// Eigen::JacobiRotation<double>::makeGivens(p, q) — real case.  G = [c s; -s c],
// G^T [p; q] = [r; 0] with r >= 0.
inline void make_givens(double p, double q, double &c, double &s) {
  if (q == 0.0) {
    c = p < 0.0 ? -1.0 : 1.0;
    s = 0.0;
  } else if (p == 0.0) {
    c = 0.0;
    s = q < 0.0 ? 1.0 : -1.0;
  } else if (std::fabs(p) > std::fabs(q)) {
    double t = q / p;
    double u = std::sqrt(1.0 + t * t);
    if (p < 0.0) u = -u;
    c = 1.0 / u;
    s = -t * c;
  } else {
    double t = p / q;
    double u = std::sqrt(1.0 + t * t);
    if (q < 0.0) u = -u;
    s = -1.0 / u;
    c = -t * s;
  }
}

// block.applyOnTheLeft(0, 1, G.adjoint()) on rows (m-1, m):  x' = c x - s y ; y' = s x + c y
inline void rot_rows(Mat &A, int m, int col0, double c, double s) {
  for (int j = col0; j < A.c; ++j) {
    double x = A(m - 1, j), y = A(m, j);
    A(m - 1, j) = c * x - s * y;
    A(m, j) = s * x + c * y;
  }
}

// StateHelper::Givens_Rotation(A, B, C)  REF: StateHelper.cpp:631-651
void givens3(Mat &A, Mat &B, Mat &C) {
  for (int n = 0; n < A.c; ++n) {
    for (int m = A.r - 1; m > n; m--) {
      if (A(m, n) == 0.0) continue;
      double c, s;
      make_givens(A(m - 1, n), A(m, n), c, s);
      rot_rows(A, m, n, c, s);
      A(m, n) = 0.0;
      rot_rows(B, m, 0, c, s);
      rot_rows(C, m, 0, c, s);
    }
  }
}

// StateHelper::Givens_Rotation(A, B)  REF: StateHelper.cpp:653-672
void givens2(Mat &A, Mat &B) {
  for (int n = 0; n < A.c; ++n) {
    for (int m = A.r - 1; m > n; m--) {
      if (A(m, n) == 0.0) continue;
      double c, s;
      make_givens(A(m - 1, n), A(m, n), c, s);
      rot_rows(A, m, n, c, s);
      A(m, n) = 0.0;
      rot_rows(B, m, 0, c, s);
    }
  }
}

Mat top_rows(const Mat &A, int row0, int nrows) {
  Mat R(nrows, A.c);
  for (int j = 0; j < A.c; ++j)
    for (int i = 0; i < nrows; ++i) R(i, j) = A(row0 + i, j);
  return R;
}

// REF: StateHelper.cpp:616-629
void nullspace_project(Mat &Hf, Mat &Hx, Mat &res) {
  givens3(Hf, Hx, res);
  int f = Hf.c;
  Hx = top_rows(Hx, f, Hx.r - f);
  res = top_rows(res, f, res.r - f);
}

// REF: StateHelper.cpp:602-614
void compress(Mat &Hx, Mat &res) {
  if (Hx.r <= Hx.c) return;
  givens2(Hx, res);
  int k = Hx.c;
  Hx = top_rows(Hx, 0, k);
  res = top_rows(res, 0, k);
}

// REF: StateHelper.cpp:466-493 with the flat column table instead of Type ids
Mat marginal_cov(const Mat &P, const int *cols, int k) {
  Mat S(k, k);
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < k; ++i) S(i, j) = P(cols[i], cols[j]);
  return S;
}

// REF: UpdaterStatistics.cpp:94-117.  S = (H P H^T + R).selfadjointView<Upper>();
// chi = res^T S^-1 res  (the reference inverts S explicitly and reads the upper triangle)
bool chi2(const Mat &Ps, const Mat &H, const Mat &res, double sigma2, double &chi) {
  Mat T = matmul(H, Ps);
  Mat S = matmul(T, transpose(H));
  int m = H.r;
  for (int i = 0; i < m; ++i) S(i, i) += sigma2;
  for (int j = 0; j < m; ++j)
    for (int i = j + 1; i < m; ++i) S(i, j) = S(j, i);
  Mat Sinv;
  if (!inverse(S, Sinv)) {
    chi = NAN;
    return false;
  }
  for (int j = 0; j < m; ++j)
    for (int i = j + 1; i < m; ++i) Sinv(i, j) = Sinv(j, i);
  double acc = 0.0;
  for (int j = 0; j < m; ++j) {
    double t = 0.0;
    for (int i = 0; i < m; ++i) t += res(i, 0) * Sinv(i, j);
    acc += t * res(j, 0);
  }
  chi = acc;
  return !std::isnan(chi);
}

// REF: StateHelper.cpp:94-173.  Returns false (nothing modified) on a negative diagonal.
bool ekf_update(Mat &P, const Mat &H, const int *cols, const Mat &res, const double *Rdiag, Mat &dx) {
  int n = P.r, r = H.r, k = H.c;
  // M_a = P[:, cols] * H^T
  Mat M(n, r);
  for (int j = 0; j < k; ++j) {
    int cj = cols[j];
    for (int q = 0; q < r; ++q) {
      double h = H(q, j);
      if (h == 0.0) continue;
      for (int i = 0; i < n; ++i) M(i, q) += P(i, cj) * h;
    }
  }
  Mat Ps = marginal_cov(P, cols, k);
  Mat S = matmul(matmul(H, Ps), transpose(H));
  for (int i = 0; i < r; ++i) S(i, i) += Rdiag ? Rdiag[i] : 1.0;
  Mat L;
  if (!cholesky_lower_from_upper(S, L)) return false;
  Mat Sinv = Mat::identity(r);
  cholesky_solve(L, Sinv);
  for (int j = 0; j < r; ++j)  // selfadjointView<Upper>
    for (int i = j + 1; i < r; ++i) Sinv(i, j) = Sinv(j, i);
  Mat K = matmul(M, Sinv);
  dx = matmul(K, res);
  Mat dC = matmul(K, transpose(M));
  for (int i = 0; i < n; ++i)
    if (P(i, i) - dC(i, i) < 0.0) return false;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= j; ++i) {
      P(i, j) -= dC(i, j);
      P(j, i) = P(i, j);
    }
  return true;
}

}  // namespace

extern "C" {

void orc_make_givens(double p, double q, double *c, double *s) { make_givens(p, q, *c, *s); }

// One feature: Hf (rows x fdim, ld), Hx (rows x k, ld), res (rows).  Output shifted up like the
// ABI (first rows-fdim rows valid).
void orc_nullspace_project(double *Hf, double *Hx, double *res, int rows, int fdim, int k, int ld) {
  Mat A = Mat::from(Hf, rows, fdim, ld), B = Mat::from(Hx, rows, k, ld), C = Mat::from(res, rows, 1, ld);
  givens3(A, B, C);
  A.to(Hf, ld);
  Mat B2 = top_rows(B, fdim, rows - fdim), C2 = top_rows(C, fdim, rows - fdim);
  B2.to(Hx, ld);
  C2.to(res, ld);
}

void orc_nullspace_batch(int F, int fdim, int k, int ld, const int *rows, double *Hf, double *Hx, double *res) {
  for (int f = 0; f < F; ++f)
    orc_nullspace_project(Hf + (size_t)f * fdim * ld, Hx + (size_t)f * k * ld, res + (size_t)f * ld, rows[f], fdim, k, ld);
}

// H (m x k, ld) / res (m) -> top min(m,k) rows.  Returns the new row count.
int orc_compress(double *H, int m, int k, int ld, double *res) {
  Mat A = Mat::from(H, m, k, ld), B = Mat::from(res, m, 1, m);
  compress(A, B);
  A.to(H, ld);
  B.to(res, A.r);
  return A.r;
}

int orc_chi2(const double *P, int n, int ldp, const double *H, int m, int k, int ldh, const int *cols,
             const double *res, double sigma2, double *chi) {
  Mat Pm = Mat::from(P, n, n, ldp);
  Mat Ps = marginal_cov(Pm, cols, k);
  Mat Hm = Mat::from(H, m, k, ldh), r = Mat::from(res, m, 1, m);
  return chi2(Ps, Hm, r, sigma2, *chi) ? 0 : -7;
}

int orc_chi2_batch(const double *P, int n, int ldp, int F, int k, int ld, const int *rows, const double *Hx,
                   const double *res, const int *cols, double sigma2, double *chi) {
  Mat Pm = Mat::from(P, n, n, ldp);
  Mat Ps = marginal_cov(Pm, cols, k);
  for (int f = 0; f < F; ++f) {
    Mat Hm = Mat::from(Hx + (size_t)f * k * ld, rows[f], k, ld), r = Mat::from(res + (size_t)f * ld, rows[f], 1, ld);
    chi2(Ps, Hm, r, sigma2, chi[f]);
  }
  return 0;
}

// returns 0, or -3 (PLV_E_NOT_PSD) with P/dx untouched
int orc_ekf_update(double *P, int n, int ldp, const double *H, int r, int k, int ldh, const int *cols,
                   const double *res, const double *Rdiag, double *dx) {
  Mat Pm = Mat::from(P, n, n, ldp), Hm = Mat::from(H, r, k, ldh), rm = Mat::from(res, r, 1, r), d;
  if (!ekf_update(Pm, Hm, cols, rm, Rdiag, d)) return -3;
  Pm.to(P, ldp);
  for (int i = 0; i < n; ++i) dx[i] = d(i, 0);
  return 0;
}

// REF: UpdaterCamera.cpp:230-293 (points, fdim=3, res_norm_gate=3) and :400-463 (lines, fdim=6,
// no norm gate).  Same packed layout as plv_msckf_update.  q95[dof] is the chi-square table.
// Stacks in the shared column space (union column order).
// (test aid, tests/decision_trace.py) when set, orc_msckf_update leaves [F][3] = chi2, the threshold, the norm of the projected residual
static double *g_gate_debug = nullptr;
void orc_set_gate_debug(double *three_per_entry) { g_gate_debug = three_per_entry; }

int orc_msckf_update(double *P, int n, int ldp, int F, int fdim, int k, int ld, const int *rows, const double *Hf_in,
                     const double *Hx_in, const double *res_in, const int *cols, double sigma2, double chi2_mult,
                     double res_norm_gate, const double *q95, uint8_t *accepted, int *n_rows_out, double *dx) {
  Mat Pm = Mat::from(P, n, n, ldp);
  Mat Ps = marginal_cov(Pm, cols, k);
  std::vector<Mat> Hs, Rs;
  int total = 0;
  for (int f = 0; f < F; ++f) {
    if (accepted) accepted[f] = 0;
    // REF: UpdaterCamera.cpp:228 `L.res.size() < 4` (points) / :406 `< 5` (lines)
    if (rows[f] < (fdim == 3 ? 4 : 5)) continue;
    Mat A = Mat::from(Hf_in + (size_t)f * fdim * ld, rows[f], fdim, ld);
    Mat B = Mat::from(Hx_in + (size_t)f * k * ld, rows[f], k, ld);
    Mat C = Mat::from(res_in + (size_t)f * ld, rows[f], 1, ld);
    nullspace_project(A, B, C);
    if (B.r < 1) continue;
    double nrm = 0.0;
    for (int i = 0; i < C.r; ++i) nrm += C(i, 0) * C(i, 0);
    nrm = std::sqrt(nrm);
    double chi;
    bool ok = chi2(Ps, B, C, sigma2, chi);
    if (g_gate_debug) g_gate_debug[3 * f] = ok ? chi : std::nan(""), g_gate_debug[3 * f + 1] = chi2_mult * q95[C.r], g_gate_debug[3 * f + 2] = nrm;
    bool pass = ok && (res_norm_gate <= 0.0 || nrm < res_norm_gate) && chi < chi2_mult * q95[C.r];
    if (!pass) continue;
    if (accepted) accepted[f] = 1;
    total += B.r;
    Hs.push_back(B);
    Rs.push_back(C);
  }
  if (n_rows_out) *n_rows_out = total;
  for (int i = 0; i < n; ++i) dx[i] = 0.0;
  if (total < 1) return 0;
  Mat H(total, k), r(total, 1);
  int at = 0;
  for (size_t i = 0; i < Hs.size(); ++i) {
    for (int j = 0; j < k; ++j)
      for (int q = 0; q < Hs[i].r; ++q) H(at + q, j) = Hs[i](q, j);
    for (int q = 0; q < Hs[i].r; ++q) r(at + q, 0) = Rs[i](q, 0);
    at += Hs[i].r;
  }
  compress(H, r);
  Mat d;
  if (!ekf_update(Pm, H, cols, r, nullptr, d)) return -3;
  Pm.to(P, ldp);
  for (int i = 0; i < n; ++i) dx[i] = d(i, 0);
  return 0;
}


// ---------------------------------------------------------------- a30: SLAM landmarks
// UpdaterCamera::slam_update for one landmark   REF: UpdaterCamera.cpp:296-338
// H = [Hx | Hf] of get_feature_jacobian_full with the landmark's columns appended, R = I (whitened).
// Returns 0 and accepted = 1/0; -3 when EKFUpdate rejects (state untouched).
int orc_slam_update(double *P, int n, int ldp, const double *H, int rows, int k, int ldh, const int *cols, const double *res,
                    double chi2_mult, const double *q95, unsigned char *accepted, double *dx) {
  Mat Pm = Mat::from(P, n, n, ldp), Hm = Mat::from(H, rows, k, ldh), rm = Mat::from(res, rows, 1, rows), d;
  *accepted = 0;
  for (int i = 0; i < n; ++i) dx[i] = 0.0;
  double chi;
  Mat Ps = marginal_cov(Pm, cols, k);
  if (!chi2(Ps, Hm, rm, 1.0, chi) || !(chi < chi2_mult * q95[rows])) return 0;  // Chi2Check (UpdaterStatistics.cpp:47-84)
  if (!ekf_update(Pm, Hm, cols, rm, nullptr, d)) return -3;
  *accepted = 1;
  Pm.to(P, ldp);
  for (int i = 0; i < n; ++i) dx[i] = d(i, 0);
  return 0;
}

// StateHelper::initialize + initialize_invertible (+ marginalize on a failed update)
// REF: StateHelper.cpp:357-439, 495-600, 235-303.  The new variable (size 3) is appended at index n.
// P_out is (n+3) x (n+3), ld = n+3.  Returns 1 = initialised, 0 = rejected (P_out undefined).
// dx_init (3) = H_L^-1 res_init, dx (n+3) = the EKF correction of the updating part (zeros if it has no rows).
int orc_slam_initialize(const double *P, int n, int ldp, int rows, int k, int ld, const double *Hf_in, const double *Hx_in,
                        const double *res_in, const int *cols, double chi2_mult, const double *q95, double *P_out, double *dx_init,
                        double *dx) {
  const int f = 3;
  Mat Pm = Mat::from(P, n, n, ldp);
  Mat HL = Mat::from(Hf_in, rows, f, ld), HR = Mat::from(Hx_in, rows, k, ld), r = Mat::from(res_in, rows, 1, ld);
  givens3(HL, HR, r);  // :391
  Mat Hxinit = top_rows(HR, 0, f), Hfinit = top_rows(HL, 0, f), rinit = top_rows(r, 0, f);
  Mat Hup = top_rows(HR, f, rows - f), rup = top_rows(r, f, rows - f);
  Mat Ps = marginal_cov(Pm, cols, k);
  // :409-424 health check: R isotropic = I after whitening; threshold uses res.rows() (all rows)
  if (rows - f > 0) {
    double chi;
    if (!chi2(Ps, Hup, rup, 1.0, chi)) return 0;
    if (chi > chi2_mult * q95[rows]) return 0;
  }
  // ---- initialize_invertible :535-598
  Mat Ma(n, f);
  for (int j = 0; j < k; ++j)
    for (int q = 0; q < f; ++q) {
      const double h = Hxinit(q, j);
      for (int i = 0; i < n; ++i) Ma(i, q) += Pm(i, cols[j]) * h;
    }
  Mat M = matmul(matmul(Hxinit, Ps), transpose(Hxinit));
  for (int i = 0; i < f; ++i) M(i, i) += 1.0;
  for (int j = 0; j < f; ++j)
    for (int i = j + 1; i < f; ++i) M(i, j) = M(j, i);  // selfadjointView<Upper>
  Mat HLinv;
  if (!inverse(Hfinit, HLinv)) return 0;
  Mat PLL = matmul(matmul(HLinv, M), transpose(HLinv));
  Mat v = matmul(HLinv, rinit);
  {
    Mat PLLinv;
    if (!inverse(PLL, PLLinv)) return 0;
    double chi = 0;
    for (int i = 0; i < f; ++i)
      for (int j = 0; j < f; ++j) chi += v(i, 0) * PLLinv(i, j) * v(j, 0);
    const double dn = std::sqrt(PLL(0, 0) * PLL(0, 0) + PLL(1, 1) * PLL(1, 1) + PLL(2, 2) * PLL(2, 2));
    if (chi < 1e-7 || dn > 1000) return 0;
  }
  for (int i = 0; i < f; ++i)
    if (PLL(i, i) < 0.0) return 0;
  const int n2 = n + f;
  Mat P2(n2, n2);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) P2(i, j) = Pm(i, j);
  Mat cross = matmul(Ma, transpose(HLinv));
  for (int q = 0; q < f; ++q)
    for (int i = 0; i < n; ++i) {
      P2(i, n + q) = -cross(i, q);
      P2(n + q, i) = -cross(i, q);
    }
  for (int i = 0; i < f; ++i)
    for (int j = 0; j < f; ++j) P2(n + i, n + j) = PLL(i, j);
  for (int i = 0; i < f; ++i) dx_init[i] = v(i, 0);
  for (int i = 0; i < n2; ++i) dx[i] = 0.0;
  // ---- :430-435 update with the remaining rows; a rejected update reverts the initialisation
  if (rows - f > 0) {
    Mat d;
    if (!ekf_update(P2, Hup, cols, rup, nullptr, d)) return 0;
    for (int i = 0; i < n2; ++i) dx[i] = d(i, 0);
  }
  P2.to(P_out, n2);
  return 1;
}

// StateHelper::marginalize: drop rows / columns [id, id + size)   REF: StateHelper.cpp:235-303
void orc_cov_marginalize(const double *P, int n, int id, int size, double *P_out) {
  const int m = n - size;
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < m; ++i) {
      const int si = i < id ? i : i + size, sj = j < id ? j : j + size;
      P_out[(size_t)j * m + i] = P[(size_t)sj * n + si];
    }
}

}  // extern "C"
