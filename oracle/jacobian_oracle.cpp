// ORACLE — TEST INFRASTRUCTURE ONLY (see dense.h).  PARITY UNPINNED.
//
// CPU fp64 restatement of the per-feature linearisation of the MSCKF update:
//   State::bounding_times / bounding_poses_n          REF: PL-VIWO/src/state/State.cpp:1023-1136
//   State::build_polynomial_data / add_polynomial     REF: State.cpp:631-798
//   State::get_interpolated_jacobian                  REF: State.cpp:833-973
//   State::get_interpolated_pose_poly                 REF: State.cpp:979-1021
//   CamRadtan::distort_f / compute_distort_jacobian   REF: open_vins/ov_core/src/cam/CamRadtan.h:127-198
//   CamBase::distort_d (float round trip)             REF: open_vins/ov_core/src/cam/CamBase.h:150-155
//   CamHelper::get_feature_jacobian_representation    REF: PL-VIWO/src/update/cam/CamHelper.cpp:21-56
//   CamHelper::get_feature_jacobian_full              REF: CamHelper.cpp:58-267
//   FeatureInitializer::single_triangulation / single_gaussnewton / compute_error
//                                                     REF: open_vins/ov_core/src/feat/FeatureInitializer.cpp:30-112,197-423
//   SO(3) helpers                                     REF: open_vins/ov_core/src/utils/quat_ops.h:135-535
// The 9x9 constraint matrix V_t of the reference is (Vandermonde 3x3) (x) I3, so its inverse is
// taken on the 3x3 factor; Eigen's colPivHouseholderQr 3x3 solves are replaced by a pivoted
// Gaussian solve and JacobiSVD's condition number by the symmetric eigenvalues (A is SPD).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

#include "../include/plviwo.h"

namespace {

struct V3 {
  double v[3];
  double &operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
};
struct M3 {
  double m[9];  // row-major
  double &operator()(int r, int c) { return m[3 * r + c]; }
  double operator()(int r, int c) const { return m[3 * r + c]; }
};
inline M3 eye() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
inline M3 mul(const M3 &a, const M3 &b) {
  M3 c;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
  return c;
}
inline M3 tr(const M3 &a) {
  M3 c;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c(i, j) = a(j, i);
  return c;
}
inline V3 mul(const M3 &a, const V3 &x) {
  return V3{{a(0, 0) * x[0] + a(0, 1) * x[1] + a(0, 2) * x[2], a(1, 0) * x[0] + a(1, 1) * x[1] + a(1, 2) * x[2],
             a(2, 0) * x[0] + a(2, 1) * x[1] + a(2, 2) * x[2]}};
}
inline M3 scale(const M3 &a, double s) {
  M3 c = a;
  for (double &x : c.m) x *= s;
  return c;
}
inline M3 add(const M3 &a, const M3 &b) {
  M3 c;
  for (int i = 0; i < 9; ++i) c.m[i] = a.m[i] + b.m[i];
  return c;
}
inline V3 sub(const V3 &a, const V3 &b) { return V3{{a[0] - b[0], a[1] - b[1], a[2] - b[2]}}; }
inline V3 addv(const V3 &a, const V3 &b) { return V3{{a[0] + b[0], a[1] + b[1], a[2] + b[2]}}; }
inline V3 sc(const V3 &a, double s) { return V3{{a[0] * s, a[1] * s, a[2] * s}}; }
inline double norm(const V3 &a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
inline M3 skew(const V3 &w) { return M3{{0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}}; }  // quat_ops.h:135
inline M3 inv3(const M3 &a) {
  double c00 = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1), c01 = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2),
         c02 = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
  double det = a(0, 0) * c00 + a(0, 1) * c01 + a(0, 2) * c02;
  double id = 1.0 / det;
  M3 r;
  r(0, 0) = c00 * id;
  r(0, 1) = (a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2)) * id;
  r(0, 2) = (a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1)) * id;
  r(1, 0) = c01 * id;
  r(1, 1) = (a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0)) * id;
  r(1, 2) = (a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2)) * id;
  r(2, 0) = c02 * id;
  r(2, 1) = (a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1)) * id;
  r(2, 2) = (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0)) * id;
  return r;
}

// quat_ops.h:231-251
M3 exp_so3(const V3 &w) {
  M3 wx = skew(w);
  double theta = norm(w);
  double A, B;
  if (theta < 1e-7) {
    A = 1;
    B = 0.5;
  } else {
    A = std::sin(theta) / theta;
    B = (1 - std::cos(theta)) / (theta * theta);
  }
  if (theta == 0) return eye();
  return add(add(eye(), scale(wx, A)), scale(mul(wx, wx), B));
}
// quat_ops.h:273-313
V3 log_so3(const M3 &R) {
  double R11 = R(0, 0), R12 = R(0, 1), R13 = R(0, 2), R21 = R(1, 0), R22 = R(1, 1), R23 = R(1, 2), R31 = R(2, 0),
         R32 = R(2, 1), R33 = R(2, 2);
  const double trc = R11 + R22 + R33;
  V3 omega;
  if (trc + 1.0 < 1e-10) {
    if (std::fabs(R33 + 1.0) > 1e-5)
      omega = sc(V3{{R13, R23, 1.0 + R33}}, M_PI / std::sqrt(2.0 + 2.0 * R33));
    else if (std::fabs(R22 + 1.0) > 1e-5)
      omega = sc(V3{{R12, 1.0 + R22, R32}}, M_PI / std::sqrt(2.0 + 2.0 * R22));
    else
      omega = sc(V3{{1.0 + R11, R21, R31}}, M_PI / std::sqrt(2.0 + 2.0 * R11));
  } else {
    double magnitude;
    const double tr_3 = trc - 3.0;
    if (tr_3 < -1e-7) {
      double theta = std::acos((trc - 1.0) / 2.0);
      magnitude = theta / (2.0 * std::sin(theta));
    } else {
      magnitude = 0.5 - tr_3 / 12.0;
    }
    omega = sc(V3{{R32 - R23, R13 - R31, R21 - R12}}, magnitude);
  }
  return omega;
}
// quat_ops.h:515-526
M3 Jl_so3(const V3 &w) {
  double theta = norm(w);
  if (theta < 1e-6) return eye();
  V3 a = sc(w, 1.0 / theta);
  M3 aat;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) aat(i, j) = a[i] * a[j];
  return add(add(scale(eye(), std::sin(theta) / theta), scale(aat, 1 - std::sin(theta) / theta)),
             scale(skew(a), (1 - std::cos(theta)) / theta));
}

inline M3 getM(const double *p) {
  M3 m;
  memcpy(m.m, p, sizeof(m.m));
  return m;
}
inline V3 getV(const double *p) { return V3{{p[0], p[1], p[2]}}; }

// State::bounding_times + bounding_poses_n for n_order = 3: index of the first of the 4 clones, or -1
int bounding_start(const plv_state_view &st, double t) {
  const int N = st.n_clones, n_order = st.intr_order;
  if (N < n_order + 1 || N < 2) return -1;
  const double *ct = st.clone_time;
  if (t < ct[0] - st.dt_exp || t > ct[N - 1] + st.dt_exp) return -1;
  int n_b = -1;
  for (int i = 0; i < N - 1; ++i)
    if (ct[i] - st.dt_exp <= t && t <= ct[i + 1] + st.dt_exp) {
      n_b = i;
      break;
    }
  if (n_b < 0) return -1;
  const int n_e = n_b + 1;
  const int n_side = (int)((double)(n_order + 1) / 2.0);
  int start = n_b - n_side + 1;
  if (n_b - n_side + 1 < 0)
    start = 0;
  else if (n_e + n_side - 1 >= N)
    start = N - 1 - n_order;
  if (start < 0 || start + 1 + n_order > N) return -1;
  return start;
}

struct Interp {
  int start;          // first of the 4 clones
  M3 R;               // interpolated R_GtoI
  V3 p;               // interpolated p_IinG
  double H[4][2][9];  // per pose: [0] = 3x3 orientation block, [1] = 3x3 position block (the 6x6 is block diagonal)
  double dt_jac[6];   // dpose/dt_offset
};

// polynomial through clones start..start+3 evaluated at t; fej selects the first-estimate poses.
// REF: State.cpp:631-723 (coefficients) + :881-958 (evaluation and Jacobians)
bool interpolate(const plv_state_view &st, double t, bool fej, bool want_jac, Interp &o) {
  const int s0 = bounding_start(st, t);
  if (s0 < 0) return false;
  if (t > st.clone_time[st.n_clones - 1]) return false;  // State.cpp:852-855 (newer than the state)
  o.start = s0;
  const double *Rs = fej ? st.clone_R_fej : st.clone_R, *ps = fej ? st.clone_p_fej : st.clone_p;
  const M3 R0 = getM(Rs + 9 * s0);
  const V3 p0 = getV(ps + 3 * s0);
  V3 th[3], dp[3];
  M3 Rw[3];
  double dts[3];
  for (int w = 0; w < 3; ++w) {
    const M3 Ri = getM(Rs + 9 * (s0 + 1 + w));
    Rw[w] = mul(Ri, tr(R0));
    th[w] = log_so3(Rw[w]);
    dp[w] = sub(getV(ps + 3 * (s0 + 1 + w)), p0);
    dts[w] = st.clone_time[s0 + 1 + w] - st.clone_time[s0];
  }
  // V(p, i) = dt_p^(i+1);  coefficients = V^-1 * differences
  M3 V;
  for (int p = 0; p < 3; ++p)
    for (int i = 0; i < 3; ++i) V(p, i) = std::pow(dts[p], i + 1);
  const M3 Vi = inv3(V);
  const double dtm = t - st.clone_time[s0];
  double lam[3], lamd[3];
  for (int w = 0; w < 3; ++w) {
    lam[w] = lamd[w] = 0;
    for (int i = 0; i < 3; ++i) {
      lam[w] += std::pow(dtm, i + 1) * Vi(i, w);
      lamd[w] += (double)(i + 1) * std::pow(dtm, i) * Vi(i, w);
    }
  }
  V3 A_ori{{0, 0, 0}}, A_pos{{0, 0, 0}};
  for (int w = 0; w < 3; ++w) {
    A_ori = addv(A_ori, sc(th[w], lam[w]));
    A_pos = addv(A_pos, sc(dp[w], lam[w]));
  }
  const M3 Rio = exp_so3(A_ori);
  o.R = mul(Rio, R0);
  o.p = addv(p0, A_pos);
  if (!want_jac) return true;
  const M3 Jl = Jl_so3(A_ori);
  // H_0 = [ sum_w (-lam_w Jl) (Jl(th_w)^-1 R_w) + Rio , (1 - sum lam) I ]
  M3 H0o = Rio;
  double lsum = 0;
  for (int w = 0; w < 3; ++w) {
    const M3 JinvW = inv3(Jl_so3(th[w]));
    H0o = add(H0o, scale(mul(Jl, mul(JinvW, Rw[w])), -lam[w]));
    const M3 Hw = scale(mul(Jl, JinvW), lam[w]);  // -dth_db_w * JlinOtoiInv_w
    memcpy(o.H[w + 1][0], Hw.m, sizeof(Hw.m));
    const M3 Pw = scale(eye(), lam[w]);
    memcpy(o.H[w + 1][1], Pw.m, sizeof(Pw.m));
    lsum += lam[w];
  }
  memcpy(o.H[0][0], H0o.m, sizeof(H0o.m));
  const M3 P0 = scale(eye(), 1.0 - lsum);
  memcpy(o.H[0][1], P0.m, sizeof(P0.m));
  V3 dori{{0, 0, 0}}, dpos{{0, 0, 0}};
  for (int w = 0; w < 3; ++w) {
    dori = addv(dori, sc(th[w], lamd[w]));
    dpos = addv(dpos, sc(dp[w], lamd[w]));
  }
  const V3 top = sc(mul(Jl, dori), -1.0);
  for (int i = 0; i < 3; ++i) {
    o.dt_jac[i] = top[i];
    o.dt_jac[3 + i] = dpos[i];
  }
  return true;
}

// CamRadtan::distort_f via CamBase::distort_d: input and output rounded through float
void distort_d(const double *K, const double uvn[2], double out[2]) {
  const float xf = (float)uvn[0], yf = (float)uvn[1];
  const double x = xf, y = yf;
  double r = std::sqrt(x * x + y * y);
  double r_2 = r * r, r_4 = r_2 * r_2;
  double x1 = x * (1 + K[4] * r_2 + K[5] * r_4) + 2 * K[6] * x * y + K[7] * (r_2 + 2 * x * x);
  double y1 = y * (1 + K[4] * r_2 + K[5] * r_4) + K[6] * (r_2 + 2 * y * y) + 2 * K[7] * x * y;
  out[0] = (double)(float)(K[0] * x1 + K[2]);
  out[1] = (double)(float)(K[1] * y1 + K[3]);
}
// CamRadtan::compute_distort_jacobian
void distort_jacobian(const double *K, const double uvn[2], double dzn[4], double dzeta[16]) {
  const double x = uvn[0], y = uvn[1];
  double r = std::sqrt(x * x + y * y);
  double r_2 = r * r, r_4 = r_2 * r_2;
  double x_2 = x * x, y_2 = y * y, x_y = x * y;
  dzn[0] = K[0] * ((1 + K[4] * r_2 + K[5] * r_4) + (2 * K[4] * x_2 + 4 * K[5] * x_2 * r_2) + 2 * K[6] * y + (2 * K[7] * x + 4 * K[7] * x));
  dzn[1] = K[0] * (2 * K[4] * x_y + 4 * K[5] * x_y * r_2 + 2 * K[6] * x + 2 * K[7] * y);
  dzn[2] = K[1] * (2 * K[4] * x_y + 4 * K[5] * x_y * r_2 + 2 * K[6] * x + 2 * K[7] * y);
  dzn[3] = K[1] * ((1 + K[4] * r_2 + K[5] * r_4) + (2 * K[4] * y_2 + 4 * K[5] * y_2 * r_2) + 2 * K[7] * x + (2 * K[6] * y + 4 * K[6] * y));
  double x1 = x * (1 + K[4] * r_2 + K[5] * r_4) + 2 * K[6] * x * y + K[7] * (r_2 + 2 * x * x);
  double y1 = y * (1 + K[4] * r_2 + K[5] * r_4) + K[6] * (r_2 + 2 * y * y) + 2 * K[7] * x * y;
  memset(dzeta, 0, 16 * sizeof(double));
  dzeta[0] = x1;
  dzeta[2] = 1;
  dzeta[4] = K[0] * x * r_2;
  dzeta[5] = K[0] * x * r_4;
  dzeta[6] = 2 * K[0] * x * y;
  dzeta[7] = K[0] * (r_2 + 2 * x * x);
  dzeta[8 + 1] = y1;
  dzeta[8 + 3] = 1;
  dzeta[8 + 4] = K[1] * y * r_2;
  dzeta[8 + 5] = K[1] * y * r_4;
  dzeta[8 + 6] = K[1] * (r_2 + 2 * y * y);
  dzeta[8 + 7] = 2 * K[1] * x * y;
}

// CamHelper::get_feature_jacobian_representation
M3 representation_jacobian(int rep, const V3 &pf) {
  if (rep == PLV_FEAT_GLOBAL_3D) return eye();
  double g_rho = 1 / norm(pf);
  double g_phi = std::acos(g_rho * pf[2]);
  double g_theta = std::atan2(pf[1], pf[0]);
  double sin_th = std::sin(g_theta), cos_th = std::cos(g_theta), sin_phi = std::sin(g_phi), cos_phi = std::cos(g_phi), rho = g_rho;
  M3 H;
  H(0, 0) = -(1.0 / rho) * sin_th * sin_phi;
  H(0, 1) = (1.0 / rho) * cos_th * cos_phi;
  H(0, 2) = -(1.0 / (rho * rho)) * cos_th * sin_phi;
  H(1, 0) = (1.0 / rho) * cos_th * sin_phi;
  H(1, 1) = (1.0 / rho) * sin_th * cos_phi;
  H(1, 2) = -(1.0 / (rho * rho)) * sin_th * sin_phi;
  H(2, 0) = 0.0;
  H(2, 1) = -(1.0 / rho) * sin_phi;
  H(2, 2) = -(1.0 / (rho * rho)) * cos_phi;
  return H;
}

inline int find_col(const int *col_to_state, int k, int state_id) {
  for (int j = 0; j < k; ++j)
    if (col_to_state[j] == state_id) return j;
  return -1;
}

}  // namespace

// REF: CamHelper.cpp:217-224 (and LineHelper's twin): R += H_ Q H_^T * mlt with H_ = HI * blockdiag(I, R_clone_fej^T)
static void add_imu_cov(const plv_state_view *st, const double *Q, int clone, const double *HI, double *Rn) {
  const double *Rc = st->clone_R_fej + 9 * (size_t)clone;
  double Hc[12], HQ[12];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j) {
      Hc[6 * i + j] = HI[6 * i + j];
      Hc[6 * i + 3 + j] = HI[6 * i + 3] * Rc[3 * j] + HI[6 * i + 4] * Rc[3 * j + 1] + HI[6 * i + 5] * Rc[3 * j + 2];
    }
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 6; ++j) {
      double s = 0;
      for (int q = 0; q < 6; ++q) s += Hc[6 * i + q] * Q[6 * q + j];
      HQ[6 * i + j] = s;
    }
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) {
      double s = 0;
      for (int q = 0; q < 6; ++q) s += HQ[6 * i + q] * Hc[6 * j + q];
      Rn[2 * i + j] += s * st->intr_err_mlt;
    }
}

extern "C" {

// column order: see plv_jacobian_columns
int orc_jacobian_columns(const plv_state_view *st, const plv_tracks *tr, int *col_to_state, int cap, int *k_out) {
  int k = 0;
  auto push = [&](int id, int size) {
    if (id < 0) return true;
    for (int j = 0; j < k; ++j)
      if (col_to_state[j] == id) return true;
    if (k + size > cap) return false;
    for (int d = 0; d < size; ++d) col_to_state[k++] = id + d;
    return true;
  };
  if (!push(st->extrinsic_state_id, 6) || !push(st->intrinsic_state_id, 8) || !push(st->dt_state_id, 1)) return -5;
  for (int f = 0; f < tr->n_feat; ++f)
    for (int o = tr->obs_ptr[f]; o < tr->obs_ptr[f + 1]; ++o) {
      int s0 = bounding_start(*st, tr->obs_time[o] + st->cam_dt);
      if (s0 < 0 || tr->obs_time[o] + st->cam_dt > st->clone_time[st->n_clones - 1]) continue;
      for (int w = 0; w < 4; ++w)
        if (!push(st->clone_state_id[s0 + w], 6)) return -5;
    }
  *k_out = k;
  return 0;
}

// interpolation at one time (exposed for the unit tests): R (9), p (3), H (4 x 2 x 9), dt_jac (6), start
int orc_interpolate(const plv_state_view *st, double t, int fej, double *R, double *p, double *H, double *dtj, int *start) {
  Interp it;
  if (!interpolate(*st, t, fej != 0, true, it)) return -1;
  memcpy(R, it.R.m, 72);
  memcpy(p, it.p.v, 24);
  memcpy(H, it.H, sizeof(it.H));
  memcpy(dtj, it.dt_jac, 48);
  *start = it.start;
  return 0;
}

// REF: CamHelper.cpp:58-267, all features.  Layout identical to plv_build_jacobians.
int orc_build_jacobians(const plv_state_view *st, const plv_tracks *tr, int k, const int *col_to_state, int ld, int *rows,
                        double *Hf, double *Hx, double *res) {
  const int F = tr->n_feat;
  memset(Hf, 0, sizeof(double) * (size_t)F * 3 * ld);
  memset(Hx, 0, sizeof(double) * (size_t)F * k * ld);
  memset(res, 0, sizeof(double) * (size_t)F * ld);
  const M3 R_ItoC = getM(st->R_ItoC);
  const V3 p_IinC = getV(st->p_IinC);
  const double *K = st->intrinsics;
  const int col_ext = st->extrinsic_state_id >= 0 ? find_col(col_to_state, k, st->extrinsic_state_id) : -1;
  const int col_int = st->intrinsic_state_id >= 0 ? find_col(col_to_state, k, st->intrinsic_state_id) : -1;
  const int col_dt = st->dt_state_id >= 0 ? find_col(col_to_state, k, st->dt_state_id) : -1;
  for (int f = 0; f < F; ++f) {
    double *hf = Hf + (size_t)f * 3 * ld, *hx = Hx + (size_t)f * k * ld, *rs = res + (size_t)f * ld;
    const V3 pf = getV(tr->p_FinG + 3 * f), pf_fej = getV(tr->p_FinG_fej + 3 * f);
    const M3 dpdl = representation_jacobian(st->feat_rep, pf_fej);
    int c = 0;
    for (int o = tr->obs_ptr[f]; o < tr->obs_ptr[f + 1]; ++o) {
      const double tm = tr->obs_time[o] + st->cam_dt;
      Interp jac;
      if (!interpolate(*st, tm, true, true, jac)) continue;  // dropped measurement
      if (2 * c + 2 > ld) return -5;
      // ---- residual with the estimate pose (provided, or the estimate polynomial)
      M3 R_GtoI;
      V3 p_IinG;
      if (tr->res_R) {
        R_GtoI = getM(tr->res_R + 9 * o);
        p_IinG = getV(tr->res_p + 3 * o);
      } else {
        Interp est;
        if (!interpolate(*st, tm, false, false, est)) continue;
        R_GtoI = est.R;
        p_IinG = est.p;
      }
      V3 p_FinI = mul(R_GtoI, sub(pf, p_IinG));
      V3 p_FinC = addv(mul(R_ItoC, p_FinI), p_IinC);
      double uvn[2] = {p_FinC[0] / p_FinC[2], p_FinC[1] / p_FinC[2]};
      double uvd[2];
      distort_d(K, uvn, uvd);
      double r2[2] = {(double)tr->obs_uv[2 * o] - uvd[0], (double)tr->obs_uv[2 * o + 1] - uvd[1]};
      // ---- Jacobians at the first estimates
      R_GtoI = jac.R;
      p_IinG = jac.p;
      double dzn[4], dzeta[16];
      distort_jacobian(K, uvn, dzn, dzeta);
      p_FinI = mul(R_GtoI, sub(pf_fej, p_IinG));
      p_FinC = addv(mul(R_ItoC, p_FinI), p_IinC);
      const double iz = 1 / p_FinC[2];
      const double dznp[6] = {iz, 0, -p_FinC[0] / (p_FinC[2] * p_FinC[2]), 0, iz, -p_FinC[1] / (p_FinC[2] * p_FinC[2])};
      const M3 dpC_dpG = mul(R_ItoC, R_GtoI);
      double dpC_dI[18];  // 3x6: [R_ItoC skew(p_FinI) | -dpC_dpG]
      const M3 left = mul(R_ItoC, skew(p_FinI));
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          dpC_dI[6 * i + j] = left(i, j);
          dpC_dI[6 * i + 3 + j] = -dpC_dpG(i, j);
        }
      double dz_dpC[6];  // 2x3 = dzn (2x2) * dznp (2x3)
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) dz_dpC[3 * i + j] = dzn[2 * i] * dznp[j] + dzn[2 * i + 1] * dznp[3 + j];
      double HI[12];  // 2x6
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 6; ++j) HI[6 * i + j] = dz_dpC[3 * i] * dpC_dI[j] + dz_dpC[3 * i + 1] * dpC_dI[6 + j] + dz_dpC[3 * i + 2] * dpC_dI[12 + j];
      // ---- noise and whitening (REF :207-239; note the reference solves with the SYMMETRIC matrix
      // built from the lower triangle of chol(R), `R_llt.llt().solve(I)`, reproduced here)
      double Rn[4] = {st->sigma_pix * st->sigma_pix, 0, 0, st->sigma_pix * st->sigma_pix};
      bool at_clone = false;
      for (int i = 0; i < st->n_clones; ++i) at_clone = at_clone || st->clone_time[i] == tm;
      if (!at_clone && st->use_pol_cov) {
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 2; ++j) {
            double s = 0;
            for (int q = 0; q < 6; ++q) s += HI[6 * i + q] * (q < 3 ? st->intr_ori_cov : st->intr_pos_cov) * HI[6 * j + q];
            Rn[2 * i + j] += s;
          }
      } else if (!at_clone && st->use_imu_cov && tr->res_Q) {
        add_imu_cov(st, tr->res_Q + 36 * (size_t)o, tr->res_clone[o], HI, Rn);
      }
      const double l00 = std::sqrt(Rn[0]), l10 = Rn[2] / l00, l11 = std::sqrt(Rn[3] - l10 * l10);
      // B = [[l00, l10],[l10, l11]] (selfadjoint lower); X = B^-1 via its own Cholesky
      const double m00 = std::sqrt(l00), m10 = l10 / m00, m11 = std::sqrt(l11 - m10 * m10);
      double Wm[4];
      for (int col = 0; col < 2; ++col) {
        double b0 = col == 0 ? 1.0 : 0.0, b1 = col == 1 ? 1.0 : 0.0;
        double y0 = b0 / m00, y1 = (b1 - m10 * y0) / m11;
        double x1 = y1 / m11, x0 = (y0 - m10 * x1) / m00;
        Wm[col] = x0;
        Wm[2 + col] = x1;
      }
      const double rw[2] = {Wm[0] * r2[0] + Wm[1] * r2[1], Wm[2] * r2[0] + Wm[3] * r2[1]};
      double wz[6], wzeta[16];
      for (int j = 0; j < 3; ++j) {
        wz[j] = Wm[0] * dz_dpC[j] + Wm[1] * dz_dpC[3 + j];
        wz[3 + j] = Wm[2] * dz_dpC[j] + Wm[3] * dz_dpC[3 + j];
      }
      for (int j = 0; j < 8; ++j) {
        wzeta[j] = Wm[0] * dzeta[j] + Wm[1] * dzeta[8 + j];
        wzeta[8 + j] = Wm[2] * dzeta[j] + Wm[3] * dzeta[8 + j];
      }
      rs[2 * c] = rw[0];
      rs[2 * c + 1] = rw[1];
      // Hf rows: wz * dpC_dpG * dpdl
      const M3 G = mul(dpC_dpG, dpdl);
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) hf[(size_t)j * ld + 2 * c + i] += wz[3 * i] * G(0, j) + wz[3 * i + 1] * G(1, j) + wz[3 * i + 2] * G(2, j);
      // Hx: wz * dpC_dI (2x6) * dTdx_w (6x6 block diagonal) for the 4 poses
      double WI[12];
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 6; ++j) WI[6 * i + j] = wz[3 * i] * dpC_dI[j] + wz[3 * i + 1] * dpC_dI[6 + j] + wz[3 * i + 2] * dpC_dI[12 + j];
      for (int w = 0; w < 4; ++w) {
        const int col = find_col(col_to_state, k, st->clone_state_id[jac.start + w]);
        if (col < 0) return -1;
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 3; ++j) {
            double so = 0, sp = 0;
            for (int q = 0; q < 3; ++q) {
              so += WI[6 * i + q] * jac.H[w][0][3 * q + j];
              sp += WI[6 * i + 3 + q] * jac.H[w][1][3 * q + j];
            }
            hx[(size_t)(col + j) * ld + 2 * c + i] += so;
            hx[(size_t)(col + 3 + j) * ld + 2 * c + i] += sp;
          }
      }
      if (col_dt >= 0)
        for (int i = 0; i < 2; ++i) {
          double s = 0;
          for (int q = 0; q < 6; ++q) s += WI[6 * i + q] * jac.dt_jac[q];
          hx[(size_t)col_dt * ld + 2 * c + i] += s;
        }
      if (col_ext >= 0) {  // dp_FinC_dT_ItoC = [skew(p_FinC - p_IinC) | I]
        const M3 sk = skew(sub(p_FinC, p_IinC));
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 3; ++j) {
            hx[(size_t)(col_ext + j) * ld + 2 * c + i] += wz[3 * i] * sk(0, j) + wz[3 * i + 1] * sk(1, j) + wz[3 * i + 2] * sk(2, j);
            hx[(size_t)(col_ext + 3 + j) * ld + 2 * c + i] += wz[3 * i + j];
          }
      }
      if (col_int >= 0)
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 8; ++j) hx[(size_t)(col_int + j) * ld + 2 * c + i] += wzeta[8 * i + j];
      ++c;
    }
    rows[f] = 2 * c;
  }
  return 0;
}

// ---------------------------------------------------------------- triangulation (a18)
// cam poses per observation: R_GtoC (9 row-major), p_CinG (3); uvn = normalised float coords.
// anchor = LAST observation (mono: REF FeatureInitializer.cpp:44-45).  Returns 1 on success.
static double tri_error(int M, const double *Rc, const double *pc, const float *uvn, const M3 &R_GtoA, const V3 &p_AinG,
                        double alpha, double beta, double rho) {
  double err = 0;
  for (int m = 0; m < M; ++m) {
    const M3 R_GtoCi = getM(Rc + 9 * m);
    const M3 R_AtoCi = mul(R_GtoCi, tr(R_GtoA));
    const V3 p_CiinA = mul(R_GtoA, sub(getV(pc + 3 * m), p_AinG));
    const V3 p_AinCi = sc(mul(R_AtoCi, p_CiinA), -1.0);
    double hi1 = R_AtoCi(0, 0) * alpha + R_AtoCi(0, 1) * beta + R_AtoCi(0, 2) + rho * p_AinCi[0];
    double hi2 = R_AtoCi(1, 0) * alpha + R_AtoCi(1, 1) * beta + R_AtoCi(1, 2) + rho * p_AinCi[1];
    double hi3 = R_AtoCi(2, 0) * alpha + R_AtoCi(2, 1) * beta + R_AtoCi(2, 2) + rho * p_AinCi[2];
    float z0 = (float)(hi1 / hi3), z1 = (float)(hi2 / hi3);
    float r0 = uvn[2 * m] - z0, r1 = uvn[2 * m + 1] - z1;
    float nrm = std::sqrt(r0 * r0 + r1 * r1);
    err += std::pow((double)nrm, 2);
  }
  return err;
}
static bool solve3(const M3 &A, const V3 &b, V3 &x) {
  double a[3][4];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) a[i][j] = A(i, j);
    a[i][3] = b[i];
  }
  for (int c = 0; c < 3; ++c) {
    int piv = c;
    for (int r = c + 1; r < 3; ++r)
      if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
    if (a[piv][c] == 0) return false;
    for (int j = 0; j < 4; ++j) std::swap(a[piv][j], a[c][j]);
    for (int r = c + 1; r < 3; ++r) {
      double f = a[r][c] / a[c][c];
      for (int j = c; j < 4; ++j) a[r][j] -= f * a[c][j];
    }
  }
  for (int i = 2; i >= 0; --i) {
    double s = a[i][3];
    for (int j = i + 1; j < 3; ++j) s -= a[i][j] * x[j];
    x[i] = s / a[i][i];
  }
  return true;
}
// eigenvalues of a symmetric 3x3 (for cond(A) = sigma_max / sigma_min of the SPD normal matrix)
static void sym_eig3(const M3 &A, double ev[3]) {
  double p1 = A(0, 1) * A(0, 1) + A(0, 2) * A(0, 2) + A(1, 2) * A(1, 2);
  double q = (A(0, 0) + A(1, 1) + A(2, 2)) / 3;
  double p2 = (A(0, 0) - q) * (A(0, 0) - q) + (A(1, 1) - q) * (A(1, 1) - q) + (A(2, 2) - q) * (A(2, 2) - q) + 2 * p1;
  double p = std::sqrt(p2 / 6);
  if (p == 0) {
    ev[0] = ev[1] = ev[2] = q;
    return;
  }
  M3 B = scale(add(A, scale(eye(), -q)), 1 / p);
  double detB = B(0, 0) * (B(1, 1) * B(2, 2) - B(1, 2) * B(2, 1)) - B(0, 1) * (B(1, 0) * B(2, 2) - B(1, 2) * B(2, 0)) +
                B(0, 2) * (B(1, 0) * B(2, 1) - B(1, 1) * B(2, 0));
  double r = detB / 2;
  double phi = r <= -1 ? M_PI / 3 : (r >= 1 ? 0 : std::acos(r) / 3);
  ev[0] = q + 2 * p * std::cos(phi);
  ev[2] = q + 2 * p * std::cos(phi + (2 * M_PI / 3));
  ev[1] = 3 * q - ev[0] - ev[2];
}

// (test aid, tests/decision_trace.py) when set, orc_triangulate leaves the values its tests looked at here: condition number and
// depth of the linear solution, depth and baseline ratio of the refined one
static double *g_tri_debug = nullptr;
void orc_set_tri_debug(double *four) { g_tri_debug = four; }

int orc_triangulate(int M, const double *Rc, const double *pc, const float *uvn, double min_dist, double max_dist,
                    double max_cond, double max_baseline, int refine, double *p_FinG_out) {
  if (g_tri_debug) g_tri_debug[0] = g_tri_debug[1] = g_tri_debug[2] = g_tri_debug[3] = std::nan("");
  if (M < 2) return 0;
  const M3 R_GtoA = getM(Rc + 9 * (M - 1));
  const V3 p_AinG = getV(pc + 3 * (M - 1));
  M3 A{{0, 0, 0, 0, 0, 0, 0, 0, 0}};
  V3 b{{0, 0, 0}};
  for (int m = 0; m < M; ++m) {
    const M3 R_AtoCi = mul(getM(Rc + 9 * m), tr(R_GtoA));
    const V3 p_CiinA = mul(R_GtoA, sub(getV(pc + 3 * m), p_AinG));
    V3 bi = mul(tr(R_AtoCi), V3{{(double)uvn[2 * m], (double)uvn[2 * m + 1], 1.0}});
    bi = sc(bi, 1.0 / norm(bi));
    const M3 Bp = skew(bi);
    const M3 Ai = mul(tr(Bp), Bp);
    A = add(A, Ai);
    b = addv(b, mul(Ai, p_CiinA));
  }
  V3 pf;
  if (!solve3(A, b, pf)) return 0;
  double ev[3];
  sym_eig3(A, ev);
  double condA = ev[0] / ev[2];
  if (g_tri_debug) g_tri_debug[0] = std::fabs(condA), g_tri_debug[1] = pf[2];
  if (std::fabs(condA) > max_cond || pf[2] < min_dist || pf[2] > max_dist || std::isnan(norm(pf))) return 0;
  if (refine) {
    // FeatureInitializerOptions defaults: max_runs 5, init_lamda 1e-3, max_lamda 1e10, min_dx 1e-6,
    // min_dcost 1e-6, lam_mult 10  (REF: FeatureInitializerOptions.h:36-69)
    double rho = 1 / pf[2], alpha = pf[0] / pf[2], beta = pf[1] / pf[2];
    double lam = 1e-3, eps = 10000;
    int runs = 0;
    bool recompute = true;
    M3 Hess{{0}};
    V3 grad{{0, 0, 0}};
    double cost_old = tri_error(M, Rc, pc, uvn, R_GtoA, p_AinG, alpha, beta, rho);
    while (runs < 5 && lam < 1e10 && eps > 1e-6) {
      if (recompute) {
        memset(Hess.m, 0, sizeof(Hess.m));
        grad = V3{{0, 0, 0}};
        for (int m = 0; m < M; ++m) {
          const M3 R_AtoCi = mul(getM(Rc + 9 * m), tr(R_GtoA));
          const V3 p_CiinA = mul(R_GtoA, sub(getV(pc + 3 * m), p_AinG));
          const V3 p_AinCi = sc(mul(R_AtoCi, p_CiinA), -1.0);
          double hi1 = R_AtoCi(0, 0) * alpha + R_AtoCi(0, 1) * beta + R_AtoCi(0, 2) + rho * p_AinCi[0];
          double hi2 = R_AtoCi(1, 0) * alpha + R_AtoCi(1, 1) * beta + R_AtoCi(1, 2) + rho * p_AinCi[1];
          double hi3 = R_AtoCi(2, 0) * alpha + R_AtoCi(2, 1) * beta + R_AtoCi(2, 2) + rho * p_AinCi[2];
          double h3s = std::pow(hi3, 2);
          double H[6] = {(R_AtoCi(0, 0) * hi3 - hi1 * R_AtoCi(2, 0)) / h3s, (R_AtoCi(0, 1) * hi3 - hi1 * R_AtoCi(2, 1)) / h3s,
                         (p_AinCi[0] * hi3 - hi1 * p_AinCi[2]) / h3s,        (R_AtoCi(1, 0) * hi3 - hi2 * R_AtoCi(2, 0)) / h3s,
                         (R_AtoCi(1, 1) * hi3 - hi2 * R_AtoCi(2, 1)) / h3s, (p_AinCi[1] * hi3 - hi2 * p_AinCi[2]) / h3s};
          float z0 = (float)(hi1 / hi3), z1 = (float)(hi2 / hi3);
          double r0 = (double)(uvn[2 * m] - z0), r1 = (double)(uvn[2 * m + 1] - z1);
          for (int i = 0; i < 3; ++i) {
            grad[i] += H[i] * r0 + H[3 + i] * r1;
            for (int j = 0; j < 3; ++j) Hess(i, j) += H[i] * H[j] + H[3 + i] * H[3 + j];
          }
        }
      }
      M3 Hl = Hess;
      for (int r = 0; r < 3; ++r) Hl(r, r) *= (1.0 + lam);
      V3 dx;
      if (!solve3(Hl, grad, dx)) break;
      double cost = tri_error(M, Rc, pc, uvn, R_GtoA, p_AinG, alpha + dx[0], beta + dx[1], rho + dx[2]);
      if (cost <= cost_old && (cost_old - cost) / cost_old < 1e-6) {
        alpha += dx[0];
        beta += dx[1];
        rho += dx[2];
        eps = 0;
        break;
      }
      if (cost <= cost_old) {
        recompute = true;
        cost_old = cost;
        alpha += dx[0];
        beta += dx[1];
        rho += dx[2];
        runs++;
        lam = lam / 10;
        eps = norm(dx);
      } else {
        recompute = false;
        lam = lam * 10;
      }
    }
    pf = V3{{alpha / rho, beta / rho, 1 / rho}};
    // baseline test: components of the camera offsets orthogonal to p_FinA (the reference spans that
    // plane with the last two columns of a Householder Q of p_FinA)
    const V3 dir = sc(pf, 1.0 / norm(pf));
    double base_max = 0;
    for (int m = 0; m < M; ++m) {
      const V3 p_CiinA = mul(R_GtoA, sub(getV(pc + 3 * m), p_AinG));
      double along = p_CiinA[0] * dir[0] + p_CiinA[1] * dir[1] + p_CiinA[2] * dir[2];
      V3 perp = sub(p_CiinA, sc(dir, along));
      base_max = std::max(base_max, norm(perp));
    }
    if (g_tri_debug) g_tri_debug[2] = pf[2], g_tri_debug[3] = norm(pf) / base_max;
    if (pf[2] < min_dist || pf[2] > max_dist || (norm(pf) / base_max) > max_baseline || std::isnan(norm(pf))) return 0;
  }
  const V3 pg = addv(mul(tr(R_GtoA), pf), p_AinG);
  p_FinG_out[0] = pg[0];
  p_FinG_out[1] = pg[1];
  p_FinG_out[2] = pg[2];
  return 1;
}


// CamHelper::get_imu_poses / get_cam_poses / feature_triangulation + the mean reprojection error of
// moving_consistency (REF: PL-VIWO/src/update/cam/CamHelper.cpp:327-483), all features.
int orc_triangulate_batch(const plv_state_view *st, const plv_tracks *trk, const plv_tri_options *opt, double *p_FinG,
                          uint8_t *ok, double *reproj_err) {
  const M3 R_ItoC = getM(st->R_ItoC);
  const V3 p_IinC = getV(st->p_IinC);
  for (int f = 0; f < trk->n_feat; ++f) {
    std::vector<double> Rc, pc;
    std::vector<float> uvn, uv;
    for (int o = trk->obs_ptr[f]; o < trk->obs_ptr[f + 1]; ++o) {
      M3 R_GtoI;
      V3 p_IinG;
      if (trk->res_R) {
        const double tq = trk->obs_time[o] + st->cam_dt;
        if (bounding_start(*st, tq) < 0 || tq > st->clone_time[st->n_clones - 1]) continue;
        R_GtoI = getM(trk->res_R + 9 * o);
        p_IinG = getV(trk->res_p + 3 * o);
      } else {
        Interp est;
        if (!interpolate(*st, trk->obs_time[o] + st->cam_dt, false, false, est)) continue;
        R_GtoI = est.R;
        p_IinG = est.p;
      }
      const M3 R_GtoC = mul(R_ItoC, R_GtoI);                      // CamHelper.cpp:388
      const V3 p_CinG = sub(p_IinG, mul(tr(R_GtoC), p_IinC));    // :389
      for (int i = 0; i < 9; ++i) Rc.push_back(R_GtoC.m[i]);
      for (int i = 0; i < 3; ++i) pc.push_back(p_CinG[i]);
      uvn.push_back(trk->obs_uvn[2 * o]);
      uvn.push_back(trk->obs_uvn[2 * o + 1]);
      uv.push_back(trk->obs_uv[2 * o]);
      uv.push_back(trk->obs_uv[2 * o + 1]);
    }
    const int M = (int)uvn.size() / 2;
    double *pf = p_FinG + 3 * f;
    pf[0] = pf[1] = pf[2] = 0;
    double *const dbg_all = g_tri_debug;  // (batch form: four values per feature)
    if (dbg_all) g_tri_debug = dbg_all + 4 * (size_t)f;
    ok[f] = (uint8_t)orc_triangulate(M, Rc.data(), pc.data(), uvn.data(), opt->min_dist, opt->max_dist, opt->max_cond_number,
                                     opt->max_baseline, opt->refine_features, pf);
    g_tri_debug = dbg_all;
    double e = 0;
    if (ok[f]) {
      for (int m = 0; m < M; ++m) {
        const V3 pC = mul(getM(&Rc[9 * m]), sub(getV(pf), getV(&pc[3 * m])));
        double un[2] = {pC[0] / pC[2], pC[1] / pC[2]}, ud[2];
        distort_d(st->intrinsics, un, ud);
        const double r0 = (double)uv[2 * m] - ud[0], r1 = (double)uv[2 * m + 1] - ud[1];
        e += std::sqrt(r0 * r0 + r1 * r1);
      }
      e /= M;
    }
    if (reproj_err) reproj_err[f] = e;
  }
  return 0;
}

}  // extern "C"

// ================================================================ lines (a27-a29)
// REF: PL-VIWO/src/update/cam/linefeat/LineHelper.cpp:733-1024 (Jacobians), :132-229, :231-293,
// :372-495, :615-650 (triangulation).  Point-line coupling is off (UpdaterCamera.cpp:373).
namespace {
inline V3 cross(const V3 &a, const V3 &b) {
  return V3{{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}};
}
inline double dot(const V3 &a, const V3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// pose used for residuals / triangulation at observation o: provided pose or the estimate polynomial
bool line_est_pose(const plv_state_view &st, const plv_line_tracks &lt, int o, double tm, M3 &R, V3 &p) {
  if (bounding_start(st, tm) < 0) return false;
  if (lt.res_R) {
    R = getM(lt.res_R + 9 * o);
    p = getV(lt.res_p + 3 * o);
    return true;
  }
  Interp est;
  if (!interpolate(st, tm, false, false, est)) return false;
  R = est.R;
  p = est.p;
  return true;
}
}  // namespace

extern "C" {

int orc_line_jacobian_columns(const plv_state_view *st, const plv_line_tracks *lt, int *col_to_state, int cap, int *k_out) {
  int k = 0;
  auto push = [&](int id, int size) {
    if (id < 0) return true;
    for (int j = 0; j < k; ++j)
      if (col_to_state[j] == id) return true;
    if (k + size > cap) return false;
    for (int d = 0; d < size; ++d) col_to_state[k++] = id + d;
    return true;
  };
  for (int l = 0; l < lt->n_lines; ++l)
    for (int o = lt->obs_ptr[l]; o < lt->obs_ptr[l + 1]; ++o) {
      const int s0 = bounding_start(*st, lt->obs_time[o] + st->cam_dt);
      if (s0 < 0) continue;
      for (int w = 0; w < 4; ++w)
        if (!push(st->clone_state_id[s0 + w], 6)) return -5;
      if (!push(st->dt_state_id, 1)) return -5;
    }
  *k_out = k;
  return 0;
}

int orc_build_line_jacobians(const plv_state_view *st, const plv_line_tracks *lt, int k, const int *col_to_state, int ld,
                             int *rows, double *Hf, double *Hx, double *res) {
  const int L = lt->n_lines;
  memset(Hf, 0, sizeof(double) * (size_t)L * 6 * ld);
  memset(Hx, 0, sizeof(double) * (size_t)L * k * ld);
  memset(res, 0, sizeof(double) * (size_t)L * ld);
  const M3 R_ItoC = getM(st->R_ItoC);
  const V3 p_IinC = getV(st->p_IinC);
  const double *Kc = st->intrinsics;  // fx fy cx cy
  const double Kl[9] = {Kc[1], 0, 0, 0, Kc[0], 0, -Kc[1] * Kc[2], -Kc[0] * Kc[3], Kc[0] * Kc[1]};  // :861-864
  const int col_dt = st->dt_state_id >= 0 ? find_col(col_to_state, k, st->dt_state_id) : -1;
  for (int l = 0; l < L; ++l) {
    double *hf = Hf + (size_t)l * 6 * ld, *hx = Hx + (size_t)l * k * ld, *rs = res + (size_t)l * ld;
    const V3 nG = getV(lt->line_FinG + 6 * l), vG = getV(lt->line_FinG + 6 * l + 3);
    int c = 0;
    for (int o = lt->obs_ptr[l]; o < lt->obs_ptr[l + 1]; ++o) {
      const double tm = lt->obs_time[o] + st->cam_dt;
      Interp jac;
      if (!interpolate(*st, tm, true, true, jac)) continue;
      if (2 * c + 2 > ld) return -5;
      M3 Re;
      V3 pe;
      if (!line_est_pose(*st, *lt, o, tm, Re, pe)) continue;
      // G_to_I (6x6) = [R, -R skew(p); 0, R]   :846-850
      double GtoI[36] = {0};
      const M3 Rsk = scale(mul(Re, skew(pe)), -1.0);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          GtoI[6 * i + j] = Re(i, j);
          GtoI[6 * i + 3 + j] = Rsk(i, j);
          GtoI[6 * (3 + i) + 3 + j] = Re(i, j);
        }
      const V3 nI = addv(mul(Re, nG), mul(Rsk, vG)), vI = mul(Re, vG);
      // I_to_C top rows = [R_ItoC, skew(p_IinC) R_ItoC]   :853-857
      const M3 SR = mul(skew(p_IinC), R_ItoC);
      const V3 nC = addv(mul(R_ItoC, nI), mul(SR, vI));
      const double ld3[3] = {Kl[0] * nC[0] + Kl[1] * nC[1] + Kl[2] * nC[2], Kl[3] * nC[0] + Kl[4] * nC[1] + Kl[5] * nC[2],
                             Kl[6] * nC[0] + Kl[7] * nC[1] + Kl[8] * nC[2]};
      const double us[3] = {(double)lt->seg_uv[4 * o], (double)lt->seg_uv[4 * o + 1], 1.0};
      const double ue[3] = {(double)lt->seg_uv[4 * o + 2], (double)lt->seg_uv[4 * o + 3], 1.0};
      const double lnorm = std::sqrt(ld3[0] * ld3[0] + ld3[1] * ld3[1]);
      const double ds = us[0] * ld3[0] + us[1] * ld3[1] + us[2] * ld3[2], de = ue[0] * ld3[0] + ue[1] * ld3[1] + ue[2] * ld3[2];
      const double r2[2] = {ds / lnorm, de / lnorm};
      // dz/dl: Identity(2,3) with four entries overwritten, ln_2 = l0^2 + l1 + l1   :921-928
      const double ln_2 = ld3[0] * ld3[0] + ld3[1] + ld3[1];
      double dzl[6] = {1, 0, 0, 0, 1, 0};
      dzl[0] = us[0] - (ld3[0] * ds) / ln_2;
      dzl[1] = us[1] - (ld3[1] * ds) / ln_2;
      dzl[3] = ue[0] - (ld3[0] * de) / ln_2;
      dzl[4] = ue[1] - (ld3[1] * de) / ln_2;
      const double isq = 1 / std::sqrt(ln_2);
      for (int i = 0; i < 6; ++i) dzl[i] *= isq;
      // dz_li (2x6) = dz_l * Kl * [R_ItoC | skew(p_IinC) R_ItoC]
      double dzK[6];
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) dzK[3 * i + j] = dzl[3 * i] * Kl[j] + dzl[3 * i + 1] * Kl[3 + j] + dzl[3 * i + 2] * Kl[6 + j];
      double dzli[12];
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {
          dzli[6 * i + j] = dzK[3 * i] * R_ItoC(0, j) + dzK[3 * i + 1] * R_ItoC(1, j) + dzK[3 * i + 2] * R_ItoC(2, j);
          dzli[6 * i + 3 + j] = dzK[3 * i] * SR(0, j) + dzK[3 * i + 1] * SR(1, j) + dzK[3 * i + 2] * SR(2, j);
        }
      // dli_dI (6x6) with the pose get_interpolated_jacobian wrote back (first estimates)   :898,940-945
      const M3 Rf = jac.R;
      const V3 pf = jac.p;
      const M3 A00 = skew(mul(Rf, sub(nG, mul(skew(pf), vG)))), A30 = skew(mul(Rf, vG)), A03 = mul(Rf, skew(vG));
      double dliI[36] = {0};
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          dliI[6 * i + j] = A00(i, j);
          dliI[6 * (3 + i) + j] = A30(i, j);
          dliI[6 * i + 3 + j] = A03(i, j);
        }
      double HI[12];
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 6; ++j) {
          double s = 0;
          for (int q = 0; q < 6; ++q) s += dzli[6 * i + q] * dliI[6 * q + j];
          HI[6 * i + j] = s;
        }
      // noise + whitening (same form as the point path)   :957-998
      double Rn[4] = {st->sigma_pix * st->sigma_pix, 0, 0, st->sigma_pix * st->sigma_pix};
      bool at_clone = false;
      for (int i = 0; i < st->n_clones; ++i) at_clone = at_clone || st->clone_time[i] == tm;
      if (!at_clone && st->use_pol_cov) {
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 2; ++j) {
            double s = 0;
            for (int q = 0; q < 6; ++q) s += HI[6 * i + q] * (q < 3 ? st->intr_ori_cov : st->intr_pos_cov) * HI[6 * j + q];
            Rn[2 * i + j] += s;
          }
      } else if (!at_clone && st->use_imu_cov && lt->res_Q) {
        add_imu_cov(st, lt->res_Q + 36 * (size_t)o, lt->res_clone[o], HI, Rn);
      }
      const double l00 = std::sqrt(Rn[0]), l10 = Rn[2] / l00, l11 = std::sqrt(Rn[3] - l10 * l10);
      const double m00 = std::sqrt(l00), m10 = l10 / m00, m11 = std::sqrt(l11 - m10 * m10);
      double Wm[4];
      for (int col = 0; col < 2; ++col) {
        double b0 = col == 0 ? 1.0 : 0.0, b1 = col == 1 ? 1.0 : 0.0;
        double y0 = b0 / m00, y1 = (b1 - m10 * y0) / m11;
        double x1 = y1 / m11, x0 = (y0 - m10 * x1) / m00;
        Wm[col] = x0;
        Wm[2 + col] = x1;
      }
      rs[2 * c] = Wm[0] * r2[0] + Wm[1] * r2[1];
      rs[2 * c + 1] = Wm[2] * r2[0] + Wm[3] * r2[1];
      double wli[12];
      for (int j = 0; j < 6; ++j) {
        wli[j] = Wm[0] * dzli[j] + Wm[1] * dzli[6 + j];
        wli[6 + j] = Wm[2] * dzli[j] + Wm[3] * dzli[6 + j];
      }
      // Hf = wli * G_to_I ; Hx = wli * dli_dI * dTdx
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 6; ++j) {
          double s = 0;
          for (int q = 0; q < 6; ++q) s += wli[6 * i + q] * GtoI[6 * q + j];
          hf[(size_t)j * ld + 2 * c + i] += s;
        }
      double WI[12];
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 6; ++j) {
          double s = 0;
          for (int q = 0; q < 6; ++q) s += wli[6 * i + q] * dliI[6 * q + j];
          WI[6 * i + j] = s;
        }
      for (int w = 0; w < 4; ++w) {
        const int col = find_col(col_to_state, k, st->clone_state_id[jac.start + w]);
        if (col < 0) return -1;
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 3; ++j) {
            double so = 0, sp = 0;
            for (int q = 0; q < 3; ++q) {
              so += WI[6 * i + q] * jac.H[w][0][3 * q + j];
              sp += WI[6 * i + 3 + q] * jac.H[w][1][3 * q + j];
            }
            hx[(size_t)(col + j) * ld + 2 * c + i] += so;
            hx[(size_t)(col + 3 + j) * ld + 2 * c + i] += sp;
          }
      }
      if (col_dt >= 0)
        for (int i = 0; i < 2; ++i) {
          double s = 0;
          for (int q = 0; q < 6; ++q) s += WI[6 * i + q] * jac.dt_jac[q];
          hx[(size_t)col_dt * ld + 2 * c + i] += s;
        }
      ++c;
    }
    rows[l] = 2 * c;
  }
  return 0;
}

int orc_triangulate_lines(const plv_state_view *st, const plv_line_tracks *lt, double *line_FinG, unsigned char *ok) {
  const M3 R_ItoC = getM(st->R_ItoC);
  const V3 p_IinC = getV(st->p_IinC);
  for (int l = 0; l < lt->n_lines; ++l) {
    ok[l] = 0;
    for (int i = 0; i < 6; ++i) line_FinG[6 * l + i] = 0;
    // usable views (LineHelper::get_imu_poses drops the ones without bounding clones)
    std::vector<int> obs;
    std::vector<M3> RI;
    std::vector<V3> pI;
    for (int o = lt->obs_ptr[l]; o < lt->obs_ptr[l + 1]; ++o) {
      M3 R;
      V3 p;
      if (!line_est_pose(*st, *lt, o, lt->obs_time[o] + st->cam_dt, R, p)) continue;
      obs.push_back(o);
      RI.push_back(R);
      pI.push_back(p);
    }
    if (obs.size() < 2) continue;  // :204-206
    const int D = lt->D ? lt->D[l] : 0;
    if (D > 0 && lt->has_pt && lt->has_pt[l]) {  // :231-293
      const V3 e{{D == 1 ? 1.0 : 0.0, D == 2 ? 1.0 : 0.0, D == 3 ? 1.0 : 0.0}};
      const V3 dir = mul(tr(RI[0]), e);
      const V3 mom = cross(getV(lt->anchor_pt + 3 * l), dir);
      for (int i = 0; i < 3; ++i) {
        line_FinG[6 * l + i] = mom[i];
        line_FinG[6 * l + 3 + i] = dir[i];
      }
      ok[l] = 1;
      continue;
    }
    // :372-495
    std::vector<M3> RC(obs.size());
    std::vector<V3> pC(obs.size());
    for (size_t m = 0; m < obs.size(); ++m) {
      RC[m] = mul(R_ItoC, RI[m]);
      pC[m] = sub(pI[m], mul(tr(RC[m]), p_IinC));
    }
    const float *u0 = lt->seg_uvn + 4 * obs[0];
    const V3 p11{{(double)u0[0], (double)u0[1], 1.0}}, p12{{(double)u0[2], (double)u0[3], 1.0}}, cam0{{0, 0, 0}};
    auto plane = [&](const V3 &a, const V3 &b, const V3 &c3, double pl[4]) {  // :615-623
      const V3 n = cross(sub(a, c3), sub(b, c3));
      pl[0] = n[0];
      pl[1] = n[1];
      pl[2] = n[2];
      pl[3] = -dot(c3, cross(a, b));
    };
    double pl0[4];
    plane(p11, p12, cam0, pl0);
    V3 dsum{{0, 0, 0}}, nsum{{0, 0, 0}};
    double dnorm = 0;
    int cnt = 0;
    for (size_t m = 1; m < obs.size(); ++m) {
      const M3 R0i = mul(RC[m], tr(RC[0]));
      const V3 pi0 = mul(RC[0], sub(pC[m], pC[0]));
      const float *um = lt->seg_uvn + 4 * obs[m];
      V3 p31{{(double)um[0], (double)um[1], 1.0}}, p32{{(double)um[2], (double)um[3], 1.0}};
      p31 = addv(mul(tr(R0i), p31), pi0);
      p32 = addv(mul(tr(R0i), p32), pi0);
      double pl1[4];
      plane(p31, p32, pi0, pl1);
      // :625-650
      V3 n1{{pl0[0], pl0[1], pl0[2]}}, n2{{pl1[0], pl1[1], pl1[2]}};
      n1 = sc(n1, 1 / norm(n1));
      n2 = sc(n2, 1 / norm(n2));
      const double cth = dot(n1, n2) / (norm(n1) * norm(n2));
      if (std::fabs(cth) >= 0.99) continue;
      auto dp = [&](int i, int j) { return pl0[i] * pl1[j] - pl1[i] * pl0[j]; };
      const V3 head{{dp(0, 3), dp(1, 3), dp(2, 3)}}, tail{{-dp(1, 2), dp(0, 2), -dp(0, 1)}};
      dsum = addv(dsum, tail);  // "direction_vector = lines[i].tail(3)"
      nsum = addv(nsum, head);
      dnorm += norm(tail);
      ++cnt;
    }
    if (cnt == 0) continue;
    const V3 rhead = sc(dsum, 1 / dnorm), rtail = sc(nsum, 1.0 / cnt);  // line_result = [dir; normal]
    const M3 R0t = tr(RC[0]);
    const V3 vW = mul(R0t, rhead);
    const V3 nW = addv(mul(R0t, rtail), mul(skew(pC[0]), mul(R0t, rhead)));
    for (int i = 0; i < 3; ++i) {
      line_FinG[6 * l + i] = nW[i];
      line_FinG[6 * l + 3 + i] = vW[i];
    }
    ok[l] = 1;
  }
  return 0;
}

// a19, use_imu_res: State::get_interpolated_pose_imu over have_cpi's lookup and create_new_cpi_linear
// (REF: PL-VIWO/src/state/State.cpp:273-355,1138-1155), query by query in the order given, inserting every
// interpolated record back into the map exactly as the reference does (`cpis[t_given] = cpi_new`).
// create_new_cpi_integrate is not restated (needs the IMU buffer): ok = 0 there.
int orc_cpi_poses(const plv_state_view *st, const plv_cpi_table *tab, int n_q, const double *t_q, double *R_out, double *p_out,
                  uint8_t *ok) {
  struct Rec {
    double clone_t, dt;
    M3 R;
    V3 alpha, v;
  };
  std::map<double, Rec> cpis;
  for (int i = 0; i < tab->n; ++i) {
    Rec r;
    r.clone_t = tab->clone_t[i], r.dt = tab->dt[i];
    std::memcpy(r.R.m, tab->R_I0toIk + 9 * i, 72);
    std::memcpy(r.alpha.v, tab->alpha + 3 * i, 24);
    std::memcpy(r.v.v, tab->v + 3 * i, 24);
    cpis[tab->t[i]] = r;
  }
  auto clone_at = [&](double t) {
    for (int i = 0; i < st->n_clones; ++i)
      if (st->clone_time[i] == t) return i;
    return -1;
  };
  for (int q = 0; q < n_q; ++q) {
    ok[q] = 0;
    std::memset(R_out + 9 * q, 0, 72);
    std::memset(p_out + 3 * q, 0, 24);
    const double t = t_q[q];
    auto have = [&]() { return cpis.find(t) != cpis.end() && clone_at(cpis.at(t).clone_t) >= 0; };
    if (!have()) {  // create_new_cpi_linear
      if (cpis.empty()) continue;
      if (t < cpis.begin()->first || t > cpis.rbegin()->first) continue;
      double t0, t1;
      if (t == cpis.begin()->first)
        t0 = t;
      else
        t0 = (--cpis.lower_bound(t))->first;
      if (t == cpis.rbegin()->first)
        t1 = t;
      else
        t1 = cpis.upper_bound(t)->first;
      const Rec c0 = cpis.at(t0), c1 = cpis.at(t1);
      if (c0.clone_t != c1.clone_t) continue;
      if (c0.clone_t < st->clone_time[0]) continue;
      Rec n;
      n.dt = t - c0.clone_t;
      n.clone_t = c0.clone_t;
      const double lambda = (t - t0) / (t1 - t0);
      n.R = mul(exp_so3(sc(log_so3(mul(c1.R, tr(c0.R))), lambda)), c0.R);
      n.alpha = addv(sc(c0.alpha, 1 - lambda), sc(c1.alpha, lambda));
      n.v = addv(sc(c0.v, 1 - lambda), sc(c1.v, lambda));
      cpis[t] = n;
    }
    const Rec c = cpis.at(t);
    const int ci = clone_at(c.clone_t);
    if (ci < 0 || cpis.find(c.clone_t) == cpis.end()) continue;  // the reference's .at() would throw
    M3 R0;
    V3 p0;
    std::memcpy(R0.m, st->clone_R + 9 * ci, 72);
    std::memcpy(p0.v, st->clone_p + 3 * ci, 24);
    const V3 v0 = cpis.at(c.clone_t).v;
    const V3 g{{tab->gravity[0], tab->gravity[1], tab->gravity[2]}};
    const M3 R = mul(c.R, R0);
    V3 p = addv(p0, sc(v0, c.dt));
    p = sub(p, sc(sc(sc(g, 0.5), c.dt), c.dt));
    p = addv(p, mul(tr(R0), c.alpha));
    std::memcpy(R_out + 9 * q, R.m, 72);
    std::memcpy(p_out + 3 * q, p.v, 24);
    ok[q] = 1;
  }
  return 0;
}

}  // extern "C"
