// ORACLE — TEST INFRASTRUCTURE ONLY (see dense.h).  PARITY UNPINNED: the reference holds no
// known-answer test for the propagator; this restatement is checked against closed-form motion, finite differences
// of its own mean propagation and a numpy restatement of EKFPropagation (tests/test_oracle_propagate.py).
//
// CPU fp64 restatement of SURVEY §8(f) rank 2, IMU propagation + window maintenance:
//   Propagator::select_imu_readings / interpolate_data   REF: PL-VIWO/src/state/Propagator.cpp:93-152,320-331
//   Propagator::propagate                                REF: Propagator.cpp:30-91
//   Propagator::predict_and_compute                      REF: Propagator.cpp:154-238
//   Propagator::predict_mean_rk4                         REF: Propagator.cpp:240-318
//   Propagator::reset_cpi                                REF: Propagator.cpp:333-357
//   CpiV1::feed_IMU (means + measurement covariance)     REF: open_vins/ov_core/src/cpi/CpiV1.cpp:32-315
//   StateHelper::EKFPropagation                          REF: PL-VIWO/src/state/StateHelper.cpp:20-92
//   StateHelper::clone / augment_clone                   REF: StateHelper.cpp:175-201,305-355
//   quaternion helpers                                   REF: open_vins/ov_core/src/utils/quat_ops.h:88-195,482-537
//   UpdaterWheel (3D types): select_wheel_data, preintegration_3D, preintegration_intrinsics_3D, compute_linear_system_3D
//                                                        REF: PL-VIWO/src/update/wheel/UpdaterWheel.cpp:72-215,327-424,472-500,648-794
// CpiV1's bias Jacobians (J_q, J_a, J_b, H_a, H_b) are not restated: nothing in PL-VIWO reads them.
#include <cmath>
#include <cstring>
#include <vector>

#include "../include/plviwo.h"

namespace {

template <int R, int C> struct Mat {
  double a[R * C];
  double &operator()(int r, int c) { return a[r * C + c]; }
  double operator()(int r, int c) const { return a[r * C + c]; }
  static Mat zero() {
    Mat m;
    std::memset(m.a, 0, sizeof(m.a));
    return m;
  }
  static Mat eye() {
    Mat m = zero();
    for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1;
    return m;
  }
};
template <int R, int K, int C> Mat<R, C> operator*(const Mat<R, K> &x, const Mat<K, C> &y) {
  Mat<R, C> o = Mat<R, C>::zero();
  for (int r = 0; r < R; ++r)
    for (int k = 0; k < K; ++k) {
      const double v = x(r, k);
      if (v == 0) continue;
      for (int c = 0; c < C; ++c) o(r, c) += v * y(k, c);
    }
  return o;
}
template <int R, int C> Mat<R, C> operator+(const Mat<R, C> &x, const Mat<R, C> &y) {
  Mat<R, C> o;
  for (int i = 0; i < R * C; ++i) o.a[i] = x.a[i] + y.a[i];
  return o;
}
template <int R, int C> Mat<R, C> operator-(const Mat<R, C> &x, const Mat<R, C> &y) {
  Mat<R, C> o;
  for (int i = 0; i < R * C; ++i) o.a[i] = x.a[i] - y.a[i];
  return o;
}
template <int R, int C> Mat<R, C> operator*(double s, const Mat<R, C> &x) {
  Mat<R, C> o;
  for (int i = 0; i < R * C; ++i) o.a[i] = s * x.a[i];
  return o;
}
template <int R, int C> Mat<C, R> T(const Mat<R, C> &x) {
  Mat<C, R> o;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) o(c, r) = x(r, c);
  return o;
}
using M3 = Mat<3, 3>;
using V3 = Mat<3, 1>;
using V4 = Mat<4, 1>;
using M15 = Mat<15, 15>;

V3 v3(const double *p) { return V3{{p[0], p[1], p[2]}}; }
V4 v4(const double *p) { return V4{{p[0], p[1], p[2], p[3]}}; }
double norm(const V3 &v) { return std::sqrt(v.a[0] * v.a[0] + v.a[1] * v.a[1] + v.a[2] * v.a[2]); }
M3 skew(const V3 &w) { return M3{{0, -w.a[2], w.a[1], w.a[2], 0, -w.a[0], -w.a[1], w.a[0], 0}}; }

M3 quat_2_Rot(const V4 &q) {  // quat_ops.h:152-157
  const V3 qv{{q.a[0], q.a[1], q.a[2]}};
  return ((2 * q.a[3] * q.a[3] - 1) * M3::eye() - (2 * q.a[3]) * skew(qv)) + 2.0 * (qv * T(qv));
}
V4 quat_multiply(const V4 &q, const V4 &p) {  // quat_ops.h:180-195
  Mat<4, 4> Qm = Mat<4, 4>::zero();
  const V3 qv{{q.a[0], q.a[1], q.a[2]}};
  const M3 B = q.a[3] * M3::eye() - skew(qv);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) Qm(r, c) = B(r, c);
    Qm(r, 3) = q.a[r];
    Qm(3, r) = -q.a[r];
  }
  Qm(3, 3) = q.a[3];
  V4 o = Qm * p;
  if (o.a[3] < 0) o = -1.0 * o;
  const double n = std::sqrt(o.a[0] * o.a[0] + o.a[1] * o.a[1] + o.a[2] * o.a[2] + o.a[3] * o.a[3]);
  for (double &x : o.a) x /= n;
  return o;
}
V4 quatnorm(V4 q) {  // quat_ops.h:496-501
  if (q.a[3] < 0) q = -1.0 * q;
  const double n = std::sqrt(q.a[0] * q.a[0] + q.a[1] * q.a[1] + q.a[2] * q.a[2] + q.a[3] * q.a[3]);
  V4 o;
  for (int i = 0; i < 4; ++i) o.a[i] = q.a[i] / n;
  return o;
}
Mat<4, 4> Omega(const V3 &w) {  // quat_ops.h:482-489
  Mat<4, 4> m = Mat<4, 4>::zero();
  const M3 s = skew(w);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) m(r, c) = -s(r, c);
    m(3, r) = -w.a[r];
    m(r, 3) = w.a[r];
  }
  return m;
}
M3 Jl_so3(const V3 &w) {  // quat_ops.h:515-525
  const double th = norm(w);
  if (th < 1e-6) return M3::eye();
  const V3 a = (1.0 / th) * w;
  return ((std::sin(th) / th) * M3::eye() + (1 - std::sin(th) / th) * (a * T(a))) + ((1 - std::cos(th)) / th) * skew(a);
}
M3 Jr_so3(const V3 &w) { return Jl_so3(-1.0 * w); }

M3 rot_of(const double *q) { return quat_2_Rot(v4(q)); }

// Propagator::predict_mean_rk4
void predict_mean_rk4(const plv_imu_state &s, const V3 &g, double dt, const V3 &w_hat1, const V3 &a_hat1, const V3 &w_hat2,
                      const V3 &a_hat2, V4 &new_q, V3 &new_v, V3 &new_p) {
  V3 w_hat = w_hat1, a_hat = a_hat1;
  const V3 w_alpha = (1.0 / dt) * (w_hat2 - w_hat1), a_jerk = (1.0 / dt) * (a_hat2 - a_hat1);
  const V4 q_0 = v4(s.q);
  const V3 p_0 = v3(s.p), v_0 = v3(s.v);
  const V4 dq_0{{0, 0, 0, 1}};
  const V4 q0_dot = 0.5 * (Omega(w_hat) * dq_0);
  const V3 p0_dot = v_0;
  const M3 R_Gto0 = quat_2_Rot(quat_multiply(dq_0, q_0));
  const V3 v0_dot = T(R_Gto0) * a_hat - g;
  const V4 k1_q = dt * q0_dot;
  const V3 k1_p = dt * p0_dot, k1_v = dt * v0_dot;

  w_hat = w_hat + (0.5 * dt) * w_alpha;  // 0.5 * w_alpha * dt
  a_hat = a_hat + (0.5 * dt) * a_jerk;
  const V4 dq_1 = quatnorm(dq_0 + 0.5 * k1_q);
  const V3 v_1 = v_0 + 0.5 * k1_v;
  const V4 q1_dot = 0.5 * (Omega(w_hat) * dq_1);
  const V3 p1_dot = v_1;
  const M3 R_Gto1 = quat_2_Rot(quat_multiply(dq_1, q_0));
  const V3 v1_dot = T(R_Gto1) * a_hat - g;
  const V4 k2_q = dt * q1_dot;
  const V3 k2_p = dt * p1_dot, k2_v = dt * v1_dot;

  const V4 dq_2 = quatnorm(dq_0 + 0.5 * k2_q);
  const V3 v_2 = v_0 + 0.5 * k2_v;
  const V4 q2_dot = 0.5 * (Omega(w_hat) * dq_2);
  const V3 p2_dot = v_2;
  const M3 R_Gto2 = quat_2_Rot(quat_multiply(dq_2, q_0));
  const V3 v2_dot = T(R_Gto2) * a_hat - g;
  const V4 k3_q = dt * q2_dot;
  const V3 k3_p = dt * p2_dot, k3_v = dt * v2_dot;

  w_hat = w_hat + (0.5 * dt) * w_alpha;
  a_hat = a_hat + (0.5 * dt) * a_jerk;
  const V4 dq_3 = quatnorm(dq_0 + k3_q);
  const V3 v_3 = v_0 + k3_v;
  const V4 q3_dot = 0.5 * (Omega(w_hat) * dq_3);
  const V3 p3_dot = v_3;
  const M3 R_Gto3 = quat_2_Rot(quat_multiply(dq_3, q_0));
  const V3 v3_dot = T(R_Gto3) * a_hat - g;
  const V4 k4_q = dt * q3_dot;
  const V3 k4_p = dt * p3_dot, k4_v = dt * v3_dot;

  const V4 dq = quatnorm((((dq_0 + (1.0 / 6.0) * k1_q) + (1.0 / 3.0) * k2_q) + (1.0 / 3.0) * k3_q) + (1.0 / 6.0) * k4_q);
  new_q = quat_multiply(dq, q_0);
  new_p = (((p_0 + (1.0 / 6.0) * k1_p) + (1.0 / 3.0) * k2_p) + (1.0 / 3.0) * k3_p) + (1.0 / 6.0) * k4_p;
  new_v = (((v_0 + (1.0 / 6.0) * k1_v) + (1.0 / 3.0) * k2_v) + (1.0 / 3.0) * k3_v) + (1.0 / 6.0) * k4_v;
}

template <int R, int C, int RR, int CC> void put(Mat<RR, CC> &dst, int r0, int c0, const Mat<R, C> &src) {
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) dst(r0 + r, c0 + c) = src(r, c);
}

// Propagator::predict_and_compute (IMU local ids: theta 0, p 3, v 6, bg 9, ba 12)
void predict_and_compute(plv_imu_state &s, const plv_imu_noise &nz, double t0, const double *wm0, const double *am0, double t1,
                         const double *wm1, const double *am1, M15 &F, M15 &Qd) {
  const int th = 0, p = 3, v = 6, bg = 9, ba = 12;
  F = M15::zero();
  const double dt = t1 - t0;
  const V3 g = v3(nz.gravity);
  const V3 w_hat = v3(wm0) - v3(s.bg), a_hat = v3(am0) - v3(s.ba), w_hat2 = v3(wm1) - v3(s.bg), a_hat2 = v3(am1) - v3(s.ba);
  V4 new_q;
  V3 new_v, new_p;
  predict_mean_rk4(s, g, dt, w_hat, a_hat, w_hat2, a_hat2, new_q, new_v, new_p);
  Mat<15, 12> G = Mat<15, 12>::zero();
  const M3 Rfej = rot_of(s.q_fej);
  const M3 dR = quat_2_Rot(new_q) * T(Rfej);
  const V3 v_fej = v3(s.v_fej), p_fej = v3(s.p_fej);
  const M3 I3 = M3::eye(), RfT = T(Rfej);
  const M3 thbg = dt * ((-1.0 * dR) * Jr_so3((-dt) * w_hat));  // -dR * Jr_so3(-w_hat * dt) * dt
  put(F, th, th, dR);
  put(F, th, bg, thbg);
  put(F, bg, bg, I3);
  put(F, v, th, (-1.0 * skew((new_v - v_fej) + dt * g)) * RfT);
  put(F, v, v, I3);
  put(F, v, ba, (-dt) * RfT);
  put(F, ba, ba, I3);
  put(F, p, th, (-1.0 * skew(((new_p - p_fej) - dt * v_fej) + (0.5 * dt * dt) * g)) * RfT);
  put(F, p, v, dt * I3);
  put(F, p, ba, (-0.5 * dt * dt) * RfT);
  put(F, p, p, I3);
  put(G, th, 0, thbg);
  put(G, v, 3, (-dt) * RfT);
  put(G, p, 3, (-0.5 * dt * dt) * RfT);
  put(G, bg, 6, I3);
  put(G, ba, 9, I3);
  Mat<12, 12> Qc = Mat<12, 12>::zero();
  for (int i = 0; i < 3; ++i) {
    Qc(i, i) = nz.sigma_w * nz.sigma_w / dt;
    Qc(3 + i, 3 + i) = nz.sigma_a * nz.sigma_a / dt;
    Qc(6 + i, 6 + i) = nz.sigma_wb * nz.sigma_wb * dt;
    Qc(9 + i, 9 + i) = nz.sigma_ab * nz.sigma_ab * dt;
  }
  Qd = (G * Qc) * T(G);
  Qd = 0.5 * (Qd + T(Qd));
  for (int i = 0; i < 4; ++i) s.q[i] = s.q_fej[i] = new_q.a[i];
  for (int i = 0; i < 3; ++i) {
    s.p[i] = s.p_fej[i] = new_p.a[i];
    s.v[i] = s.v_fej[i] = new_v.a[i];
  }
}

// CpiV1::feed_IMU with imu_avg = true: means and the RK4 measurement covariance
void cpi_feed(plv_cpi_accum &c, const plv_imu_noise &nz, double t0, double t1, const double *w0, const double *a0, const double *w1,
              const double *a1) {
  const double delta_t = t1 - t0;
  c.DT += delta_t;
  if (delta_t == 0) return;
  V3 w_hat = v3(w0) - v3(c.b_w_lin), a_hat = v3(a0) - v3(c.b_a_lin);
  w_hat = w_hat + (v3(w1) - v3(c.b_w_lin));
  w_hat = 0.5 * w_hat;
  a_hat = a_hat + (v3(a1) - v3(c.b_a_lin));
  a_hat = 0.5 * a_hat;
  const double mag_w = norm(w_hat), w_dt = mag_w * delta_t;
  const bool small_w = mag_w < 0.008726646;
  const double dt_2 = delta_t * delta_t, cos_wt = std::cos(w_dt), sin_wt = std::sin(w_dt);
  const M3 w_x = skew(w_hat), a_x = skew(a_hat), w_x_2 = w_x * w_x, eye3 = M3::eye();
  M3 R_k2tau;
  std::memcpy(R_k2tau.a, c.R_k2tau, 72);
  const M3 R_tau2tau1 = small_w ? (eye3 - delta_t * w_x) + (dt_2 / 2) * w_x_2
                                : (eye3 - (sin_wt / mag_w) * w_x) + ((1.0 - cos_wt) / (mag_w * mag_w)) * w_x_2;
  const M3 R_k2tau1 = R_tau2tau1 * R_k2tau, R_tau12k = T(R_k2tau1);
  double f_1, f_2, f_3, f_4;
  if (small_w) {
    f_1 = -(std::pow(delta_t, 3) / 3);
    f_2 = std::pow(delta_t, 4) / 8;
    f_3 = -(dt_2 / 2);
    f_4 = std::pow(delta_t, 3) / 6;
  } else {
    f_1 = (w_dt * cos_wt - sin_wt) / std::pow(mag_w, 3);
    f_2 = (w_dt * w_dt - 2 * cos_wt - 2 * w_dt * sin_wt + 2) / (2 * std::pow(mag_w, 4));
    f_3 = -(1 - cos_wt) / (mag_w * mag_w);
    f_4 = (w_dt - sin_wt) / std::pow(mag_w, 3);
  }
  const M3 alpha_arg = ((dt_2 / 2.0) * eye3 + f_1 * w_x) + f_2 * w_x_2;
  const M3 Beta_arg = (delta_t * eye3 + f_3 * w_x) + f_4 * w_x_2;
  const M3 H_al = R_tau12k * alpha_arg, H_be = R_tau12k * Beta_arg;
  V3 alpha = v3(c.alpha_tau), beta = v3(c.beta_tau);
  alpha = alpha + (delta_t * beta + H_al * a_hat);
  beta = beta + H_be * a_hat;
  // measurement covariance, RK4
  const double hw = mag_w * .5 * delta_t;
  M3 R_mid = small_w ? (eye3 - (.5 * delta_t) * w_x) + (std::pow(.5 * delta_t, 2) / 2) * w_x_2
                     : (eye3 - (std::sin(hw) / mag_w) * w_x) + ((1.0 - std::cos(hw)) / (mag_w * mag_w)) * w_x_2;
  R_mid = R_mid * R_k2tau;
  Mat<12, 12> Q_c = Mat<12, 12>::zero();  // CpiBase ctor: sigma_w^2, sigma_wb^2, sigma_a^2, sigma_ab^2
  for (int i = 0; i < 3; ++i) {
    Q_c(i, i) = nz.sigma_w * nz.sigma_w;
    Q_c(3 + i, 3 + i) = nz.sigma_wb * nz.sigma_wb;
    Q_c(6 + i, 6 + i) = nz.sigma_a * nz.sigma_a;
    Q_c(9 + i, 9 + i) = nz.sigma_ab * nz.sigma_ab;
  }
  auto FG = [&](const M3 &R, M15 &F, Mat<15, 12> &G) {
    F = M15::zero();
    G = Mat<15, 12>::zero();
    put(F, 0, 0, -1.0 * w_x);
    put(F, 0, 3, -1.0 * eye3);
    put(F, 6, 0, (-1.0 * T(R)) * a_x);
    put(F, 6, 9, -1.0 * T(R));
    put(F, 12, 6, eye3);
    put(G, 0, 0, -1.0 * eye3);
    put(G, 3, 3, eye3);
    put(G, 6, 6, -1.0 * T(R));
    put(G, 9, 9, eye3);
  };
  M15 P;
  std::memcpy(P.a, c.P_meas, sizeof(P.a));
  auto Pdot = [&](const M15 &F, const Mat<15, 12> &G, const M15 &Pk) { return (F * Pk + Pk * T(F)) + (G * Q_c) * T(G); };
  M15 F1, F2, F4;
  Mat<15, 12> G1, G2, G4;
  FG(R_k2tau, F1, G1);
  FG(R_mid, F2, G2);
  FG(R_k2tau1, F4, G4);
  const M15 Pd1 = Pdot(F1, G1, P);
  const M15 Pd2 = Pdot(F2, G2, P + (delta_t / 2.0) * Pd1);
  const M15 Pd3 = Pdot(F2, G2, P + (delta_t / 2.0) * Pd2);
  const M15 Pd4 = Pdot(F4, G4, P + delta_t * Pd3);
  P = P + (delta_t / 6.0) * (((Pd1 + 2.0 * Pd2) + 2.0 * Pd3) + Pd4);
  P = 0.5 * (P + T(P));
  std::memcpy(c.P_meas, P.a, sizeof(P.a));
  std::memcpy(c.R_k2tau, R_k2tau1.a, 72);
  std::memcpy(c.alpha_tau, alpha.a, 24);
  std::memcpy(c.beta_tau, beta.a, 24);
}

}  // namespace

extern "C" {

// Propagator::select_imu_readings.  Returns 1 and fills the outputs on success, 0 when the reference returns false.
int orc_select_imu_readings(int n, const double *t, const double *wm, const double *am, double time0, double time1, int cap,
                            double *ot, double *owm, double *oam, int *n_out) {
  *n_out = 0;
  if (n < 2) return 0;
  if (time1 <= time0) return 0;
  if (t[0] > time0) return 0;
  if (t[n - 1] < time1) return 0;
  int m = 0;
  auto push = [&](double tt, const double *w, const double *a) {
    if (m < cap) {
      ot[m] = tt;
      std::memcpy(owm + 3 * m, w, 24);
      std::memcpy(oam + 3 * m, a, 24);
    }
    ++m;
  };
  auto interp = [&](int i, double ts) {
    const double lambda = (ts - t[i]) / (t[i + 1] - t[i]);
    double w[3], a[3];
    for (int c = 0; c < 3; ++c) {
      a[c] = (1 - lambda) * am[3 * i + c] + lambda * am[3 * (i + 1) + c];
      w[c] = (1 - lambda) * wm[3 * i + c] + lambda * wm[3 * (i + 1) + c];
    }
    push(ts, w, a);
  };
  size_t i = 0;
  const size_t N = (size_t)n;
  for (; i < N - 1; i++)
    if (t[i] <= time0 && time0 <= t[i + 1]) {
      interp((int)i, time0);
      break;
    }
  for (i == 0 ? i = 0 : i--; i < N - 1; i++) {
    if (time0 < t[i] && t[i + 1] < time1) push(t[i], wm + 3 * i, am + 3 * i);
    if (t[i + 1] > time1) break;
  }
  for (i == 0 ? i = 0 : i--; i < N - 1; i++)
    if (t[i] <= time1 && time1 <= t[i + 1]) {
      interp((int)i, time1);
      break;
    }
  *n_out = m;
  return m <= cap ? 1 : 0;
}

// Propagator::reset_cpi for the accumulator (the caller keeps State::cpis)
void orc_reset_cpi(plv_cpi_accum *c, const plv_imu_state *imu, double clone_t) {
  std::memset(c, 0, sizeof(*c));
  c->clone_t = clone_t;
  c->R_k2tau[0] = c->R_k2tau[4] = c->R_k2tau[8] = 1;
  std::memcpy(c->b_w_lin, imu->bg, 24);
  std::memcpy(c->b_a_lin, imu->ba, 24);
  std::memcpy(c->v_clone, imu->v, 24);
}

// Propagator::propagate over the selected samples + StateHelper::EKFPropagation on P (n x n col-major, ld; IMU block at
// imu_id).  records: n_data - 1 entries, nullable with cpi.
int orc_propagate(plv_imu_state *imu, const plv_imu_noise *nz, int n_data, const double *t, const double *wm, const double *am,
                  plv_cpi_accum *cpi, plv_cpi_record *records, double *P, int n, int ld, int imu_id, double *Phi_out,
                  double *Qd_out) {
  if (n_data < 2) return -1;
  M15 Phi = M15::eye(), Qd = M15::zero();
  M3 R_GtoIk = rot_of(imu->q);
  const V3 g = v3(nz->gravity);
  for (int i = 0; i < n_data - 1; ++i) {
    M15 F, Qdi;
    predict_and_compute(*imu, *nz, t[i], wm + 3 * i, am + 3 * i, t[i + 1], wm + 3 * (i + 1), am + 3 * (i + 1), F, Qdi);
    Phi = F * Phi;
    Qd = (F * Qd) * T(F) + Qdi;
    Qd = 0.5 * (Qd + T(Qd));
    if (cpi) {
      cpi_feed(*cpi, *nz, t[i], t[i + 1], wm + 3 * i, am + 3 * i, wm + 3 * (i + 1), am + 3 * (i + 1));
      M3 Rk;
      std::memcpy(Rk.a, cpi->R_k2tau, 72);
      if (records) {
        plv_cpi_record &r = records[i];
        r.t = t[i + 1];
        r.dt = cpi->DT;
        r.clone_t = cpi->clone_t;
        std::memcpy(r.R_I0toIk, cpi->R_k2tau, 72);
        std::memcpy(r.alpha, cpi->alpha_tau, 24);
        const V3 w = v3(wm + 3 * (i + 1)) - v3(imu->bg);
        std::memcpy(r.w, w.a, 24);
        const V3 vv = (v3(cpi->v_clone) - cpi->DT * g) + T(R_GtoIk) * v3(cpi->beta_tau);
        std::memcpy(r.v, vv.a, 24);
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) {
            r.Q[6 * a + b] = cpi->P_meas[15 * a + b];
            r.Q[6 * a + 3 + b] = cpi->P_meas[15 * a + 12 + b];
            r.Q[6 * (3 + a) + b] = cpi->P_meas[15 * (12 + a) + b];
            r.Q[6 * (3 + a) + 3 + b] = cpi->P_meas[15 * (12 + a) + 12 + b];
          }
      }
      R_GtoIk = Rk * R_GtoIk;
    }
  }
  if (Phi_out) std::memcpy(Phi_out, Phi.a, sizeof(Phi.a));
  if (Qd_out) std::memcpy(Qd_out, Qd.a, sizeof(Qd.a));
  if (P) {  // EKFPropagation with order_NEW = order_OLD = {imu}
    std::vector<double> CovPhiT((size_t)n * 15, 0.0);  // n x 15 row-major
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < 15; ++c) {
        double s = 0;
        for (int k = 0; k < 15; ++k) s += P[(size_t)(imu_id + k) * ld + r] * Phi(c, k);
        CovPhiT[(size_t)r * 15 + c] = s;
      }
    M15 PCP;
    for (int r = 0; r < 15; ++r)
      for (int c = 0; c < 15; ++c) {
        double s = r <= c ? Qd(r, c) : Qd(c, r);  // selfadjointView<Upper>
        for (int k = 0; k < 15; ++k) s += Phi(r, k) * CovPhiT[(size_t)(imu_id + k) * 15 + c];
        PCP(r, c) = s;
      }
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < 15; ++c) {
        P[(size_t)r * ld + imu_id + c] = CovPhiT[(size_t)r * 15 + c];      // row block = Cov_PhiT^T  (col-major: P(imu+c, r))
        P[(size_t)(imu_id + c) * ld + r] = CovPhiT[(size_t)r * 15 + c];    // column block = Cov_PhiT
      }
    for (int r = 0; r < 15; ++r)
      for (int c = 0; c < 15; ++c) P[(size_t)(imu_id + c) * ld + imu_id + r] = PCP(r, c);
  }
  return 0;
}

// State::create_new_cpi_integrate (REF: PL-VIWO/src/state/State.cpp:357-415) for one query; returns 1 on success.
int orc_cpi_integrate(const plv_imu_noise *nz, double t_given, double clone_t, const double *R_GtoI_clone, const double *v_clone,
                      const double *bg, const double *ba, int n_imu, const double *t, const double *wm, const double *am,
                      plv_cpi_record *out) {
  std::vector<double> st((size_t)n_imu + 2), sw(3 * ((size_t)n_imu + 2)), sa(3 * ((size_t)n_imu + 2));
  int m = 0;
  const bool fwd = clone_t <= t_given;
  if (!orc_select_imu_readings(n_imu, t, wm, am, fwd ? clone_t : t_given, fwd ? t_given : clone_t, n_imu + 2, st.data(), sw.data(),
                               sa.data(), &m))
    return 0;
  if (!fwd)
    for (int i = 0; i < m / 2; ++i) {
      std::swap(st[i], st[m - 1 - i]);
      for (int c = 0; c < 3; ++c) {
        std::swap(sw[3 * i + c], sw[3 * (m - 1 - i) + c]);
        std::swap(sa[3 * i + c], sa[3 * (m - 1 - i) + c]);
      }
    }
  plv_cpi_accum c;
  std::memset(&c, 0, sizeof(c));
  c.clone_t = clone_t;
  c.R_k2tau[0] = c.R_k2tau[4] = c.R_k2tau[8] = 1;
  std::memcpy(c.b_w_lin, bg, 24);
  std::memcpy(c.b_a_lin, ba, 24);
  std::memcpy(c.v_clone, v_clone, 24);
  M3 R_GtoIk;
  std::memcpy(R_GtoIk.a, R_GtoI_clone, 72);
  for (int i = 0; i < m - 1; ++i) {
    M3 Rk;
    std::memcpy(Rk.a, c.R_k2tau, 72);
    R_GtoIk = Rk * R_GtoIk;
    cpi_feed(c, *nz, st[i], st[i + 1], &sw[3 * i], &sa[3 * i], &sw[3 * (i + 1)], &sa[3 * (i + 1)]);
  }
  out->t = t_given;
  out->dt = t_given - clone_t;
  out->clone_t = clone_t;
  std::memcpy(out->R_I0toIk, c.R_k2tau, 72);
  std::memcpy(out->alpha, c.alpha_tau, 24);
  const V3 w = v3(&sw[3 * (m - 1)]) - v3(bg);
  std::memcpy(out->w, w.a, 24);
  const V3 vv = (v3(v_clone) - c.DT * v3(nz->gravity)) + T(R_GtoIk) * v3(c.beta_tau);
  std::memcpy(out->v, vv.a, 24);
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      out->Q[6 * a + b] = c.P_meas[15 * a + b];
      out->Q[6 * a + 3 + b] = c.P_meas[15 * a + 12 + b];
      out->Q[6 * (3 + a) + b] = c.P_meas[15 * (12 + a) + b];
      out->Q[6 * (3 + a) + 3 + b] = c.P_meas[15 * (12 + a) + 12 + b];
    }
  return 1;
}

// ---------------------------------------------------------------------------------------------------- wheel (3D)
// UpdaterWheel::select_wheel_data
int orc_select_wheel_data(int n, const double *t, const double *m1, const double *m2, double time0, double time1, int cap, double *ot,
                          double *o1, double *o2, int *n_out) {
  *n_out = 0;
  if (n < 1) return 0;
  if (t[n - 1] <= time1 || t[0] > time0) return 0;
  std::vector<double> vt, v1, v2;
  auto push = [&](double a, double b, double c) {
    vt.push_back(a);
    v1.push_back(b);
    v2.push_back(c);
  };
  auto interp = [&](int a, int b, double ts) {
    const double lambda = (ts - t[a]) / (t[b] - t[a]);
    push(ts, (1 - lambda) * m1[a] + lambda * m1[b], (1 - lambda) * m2[a] + lambda * m2[b]);
  };
  for (int i = 0; i < n - 1; i++) {
    if (t[i + 1] > time0 && t[i] < time0) {
      interp(i, i + 1, time0);
      continue;
    }
    if (t[i] >= time0 && t[i + 1] <= time1) {
      push(t[i], m1[i], m2[i]);
      continue;
    }
    if (t[i + 1] > time1) {
      if (t[i] > time1)
        interp(i - 1, i, time1);
      else
        push(t[i], m1[i], m2[i]);
      if (vt.back() != time1) interp(i, i + 1, time1);
      break;
    }
  }
  if (vt.size() < 2) return 0;
  for (size_t i = 0; i + 1 < vt.size(); i++)
    if (std::fabs(vt[i + 1] - vt[i]) < 1e-12) {
      vt.erase(vt.begin() + i);
      v1.erase(v1.begin() + i);
      v2.erase(v2.begin() + i);
      i--;
    }
  *n_out = (int)vt.size();
  if ((int)vt.size() > cap) return 0;
  std::copy(vt.begin(), vt.end(), ot);
  std::copy(v1.begin(), v1.end(), o1);
  std::copy(v2.begin(), v2.end(), o2);
  return 1;
}

namespace {
V4 rot_2_quat(const M3 &rot) {  // quat_ops.h:88-120
  V4 q;
  const double Tr = rot(0, 0) + rot(1, 1) + rot(2, 2);
  if ((rot(0, 0) >= Tr) && (rot(0, 0) >= rot(1, 1)) && (rot(0, 0) >= rot(2, 2))) {
    q.a[0] = std::sqrt((1 + (2 * rot(0, 0)) - Tr) / 4);
    q.a[1] = (1 / (4 * q.a[0])) * (rot(0, 1) + rot(1, 0));
    q.a[2] = (1 / (4 * q.a[0])) * (rot(0, 2) + rot(2, 0));
    q.a[3] = (1 / (4 * q.a[0])) * (rot(1, 2) - rot(2, 1));
  } else if ((rot(1, 1) >= Tr) && (rot(1, 1) >= rot(0, 0)) && (rot(1, 1) >= rot(2, 2))) {
    q.a[1] = std::sqrt((1 + (2 * rot(1, 1)) - Tr) / 4);
    q.a[0] = (1 / (4 * q.a[1])) * (rot(0, 1) + rot(1, 0));
    q.a[2] = (1 / (4 * q.a[1])) * (rot(1, 2) + rot(2, 1));
    q.a[3] = (1 / (4 * q.a[1])) * (rot(2, 0) - rot(0, 2));
  } else if ((rot(2, 2) >= Tr) && (rot(2, 2) >= rot(0, 0)) && (rot(2, 2) >= rot(1, 1))) {
    q.a[2] = std::sqrt((1 + (2 * rot(2, 2)) - Tr) / 4);
    q.a[0] = (1 / (4 * q.a[2])) * (rot(0, 2) + rot(2, 0));
    q.a[1] = (1 / (4 * q.a[2])) * (rot(1, 2) + rot(2, 1));
    q.a[3] = (1 / (4 * q.a[2])) * (rot(0, 1) - rot(1, 0));
  } else {
    q.a[3] = std::sqrt((1 + Tr) / 4);
    q.a[0] = (1 / (4 * q.a[3])) * (rot(1, 2) - rot(2, 1));
    q.a[1] = (1 / (4 * q.a[3])) * (rot(2, 0) - rot(0, 2));
    q.a[2] = (1 / (4 * q.a[3])) * (rot(0, 1) - rot(1, 0));
  }
  if (q.a[3] < 0) q = -1.0 * q;
  const double n = std::sqrt(q.a[0] * q.a[0] + q.a[1] * q.a[1] + q.a[2] * q.a[2] + q.a[3] * q.a[3]);
  for (double &x : q.a) x /= n;
  return q;
}
M3 exp_so3(const V3 &w) {  // quat_ops.h:231-251
  const M3 wx = skew(w);
  const double theta = norm(w);
  double A, B;
  if (theta < 1e-7) {
    A = 1;
    B = 0.5;
  } else {
    A = std::sin(theta) / theta;
    B = (1 - std::cos(theta)) / (theta * theta);
  }
  if (theta == 0) return M3::eye();
  return (M3::eye() + A * wx) + B * (wx * wx);
}
V3 log_so3(const M3 &R) {  // quat_ops.h:273-313
  const double R11 = R(0, 0), R12 = R(0, 1), R13 = R(0, 2), R21 = R(1, 0), R22 = R(1, 1), R23 = R(1, 2), R31 = R(2, 0), R32 = R(2, 1),
               R33 = R(2, 2);
  const double trc = R11 + R22 + R33;
  if (trc + 1.0 < 1e-10) {
    if (std::fabs(R33 + 1.0) > 1e-5) return (M_PI / std::sqrt(2.0 + 2.0 * R33)) * V3{{R13, R23, 1.0 + R33}};
    if (std::fabs(R22 + 1.0) > 1e-5) return (M_PI / std::sqrt(2.0 + 2.0 * R22)) * V3{{R12, 1.0 + R22, R32}};
    return (M_PI / std::sqrt(2.0 + 2.0 * R11)) * V3{{1.0 + R11, R21, R31}};
  }
  double magnitude;
  const double tr_3 = trc - 3.0;
  if (tr_3 < -1e-7) {
    const double theta = std::acos((trc - 1.0) / 2.0);
    magnitude = theta / (2.0 * std::sin(theta));
  } else {
    magnitude = 0.5 - tr_3 / 12.0;
  }
  return magnitude * V3{{R32 - R23, R13 - R31, R21 - R12}};
}
M3 m3(const double *p) {
  M3 m;
  std::memcpy(m.a, p, 72);
  return m;
}
}  // namespace

// UpdaterWheel::update up to the linear system (3D types).  H is 6 x k col-major; returns k.
int orc_wheel_linear_system_2d(const plv_wheel_options *op, const plv_wheel_state *st, int n_data, const double *t, const double *m1,
                               const double *m2, double *H, double *res, double *Cov, int *col_to_state, double *meas);

int orc_wheel_linear_system(const plv_wheel_options *op, const plv_wheel_state *st, int n_data, const double *t, const double *m1,
                            const double *m2, double *H, double *res, double *Cov, int *col_to_state, double *R_out, double *p_out) {
  if (op->type >= PLV_WHEEL2D_ANG) {
    if (R_out) {
      const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      std::memcpy(R_out, I, 72);
    }
    double meas[3];
    const int k2 = orc_wheel_linear_system_2d(op, st, n_data, t, m1, m2, H, res, Cov, col_to_state, meas);
    if (p_out) std::memcpy(p_out, meas, 24);
    return k2;
  }
  using M6 = Mat<6, 6>;
  M3 R_3D = M3::eye(), dR_di = M3::zero(), dp_di = M3::zero();
  V3 p_3D = V3::zero();
  M6 Cov_3D = M6::zero();
  const double rl = st->intr[0], rr = st->intr[1], b = st->intr[2];
  auto vel = [&](double a1, double a2, V3 &w, V3 &v) {
    if (op->type == PLV_WHEEL3D_ANG) {
      w = V3{{0, 0, (a2 * rr - a1 * rl) / b}};
      v = V3{{(a2 * rr + a1 * rl) / 2, 0, 0}};
    } else if (op->type == PLV_WHEEL3D_LIN) {
      w = V3{{0, 0, (a2 - a1) / b}};
      v = V3{{(a2 + a1) / 2, 0, 0}};
    } else {
      w = V3{{0, 0, a1}};
      v = V3{{a2, 0, 0}};
    }
  };
  for (int i = 0; i < n_data - 1; ++i) {
    const double dt = t[i + 1] - t[i];
    if (op->do_calib_int) {  // preintegration_intrinsics_3D :472-500
      const double w_l = m1[i], w_r = m2[i];
      const V3 w{{0, 0, (w_r * rr - w_l * rl) / b}}, v{{(w_r * rr + w_l * rl) / 2, 0, 0}};
      M3 Hwx = M3::zero(), Hvx = M3::zero();
      Hwx(2, 0) = -w_l / b;
      Hwx(2, 1) = w_r / b;
      Hwx(2, 2) = -(w_r * rr - w_l * rl) / (b * b);
      Hvx(0, 0) = w_l / 2;
      Hvx(0, 1) = w_r / 2;
      const M3 R = exp_so3((-dt) * w);
      const M3 Hth = dt * Jl_so3((-dt) * w);
      dp_di = (dp_di - (T(R_3D) * skew(dt * v)) * dR_di) + dt * (T(R_3D) * Hvx);
      dR_di = R * dR_di + Hth * Hwx;
    }
    // preintegration_3D :648-782
    V3 w_hat1, v_hat1, w_hat2, v_hat2;
    vel(m1[i], m2[i], w_hat1, v_hat1);
    vel(m1[i + 1], m2[i + 1], w_hat2, v_hat2);
    V3 w_hat = w_hat1, v_hat = v_hat1;
    const V3 w_alpha = (1.0 / dt) * (w_hat2 - w_hat1), v_jerk = (1.0 / dt) * (v_hat2 - v_hat1);
    const V4 q_local = rot_2_quat(R_3D);
    const V4 dq_0{{0, 0, 0, 1}};
    auto qdot = [&](const V4 &dq) { return 0.5 * (Omega(w_hat) * dq); };
    auto pdot = [&](const V4 &dq) { return T(quat_2_Rot(quat_multiply(dq, q_local))) * v_hat; };
    const V4 k1_q = dt * qdot(dq_0);
    const V3 k1_p = dt * pdot(dq_0);
    w_hat = w_hat + (0.5 * dt) * w_alpha;
    v_hat = v_hat + (0.5 * dt) * v_jerk;
    const V4 dq_1 = quatnorm(dq_0 + 0.5 * k1_q);
    const V4 k2_q = dt * qdot(dq_1);
    const V3 k2_p = dt * pdot(dq_1);
    const V4 dq_2 = quatnorm(dq_0 + 0.5 * k2_q);
    const V4 k3_q = dt * qdot(dq_2);
    const V3 k3_p = dt * pdot(dq_2);
    w_hat = w_hat + (0.5 * dt) * w_alpha;
    v_hat = v_hat + (0.5 * dt) * v_jerk;
    const V4 dq_3 = quatnorm(dq_0 + k3_q);
    const V4 k4_q = dt * qdot(dq_3);
    const V3 k4_p = dt * pdot(dq_3);
    const V4 dq = quatnorm((((dq_0 + (1.0 / 6.0) * k1_q) + (1.0 / 3.0) * k2_q) + (1.0 / 3.0) * k3_q) + (1.0 / 6.0) * k4_q);
    const M3 R_new = quat_2_Rot(quat_multiply(dq, q_local));
    const V3 new_p = (((p_3D + (1.0 / 6.0) * k1_p) + (1.0 / 3.0) * k2_p) + (1.0 / 3.0) * k3_p) + (1.0 / 6.0) * k4_p;
    M6 Q = M6::zero();
    const double nw = op->noise_w * op->noise_w, nv = op->noise_v * op->noise_v, np = op->noise_p * op->noise_p;
    if (op->type == PLV_WHEEL3D_ANG) {
      Q(0, 0) = nw / dt, Q(3, 3) = nw / dt;
    } else if (op->type == PLV_WHEEL3D_LIN) {
      Q(0, 0) = nv / b / b / dt, Q(3, 3) = nv / 2 / 2 / dt;
    } else {
      Q(0, 0) = nw / dt, Q(3, 3) = nv / dt;
    }
    Q(1, 1) = Q(2, 2) = Q(4, 4) = Q(5, 5) = np / dt;
    M6 Phi_tr = M6::zero(), Phi_ns = M6::zero();
    put(Phi_tr, 0, 0, R_new * T(R_3D));
    put(Phi_tr, 3, 0, (-1.0 * T(R_3D)) * skew(T(R_3D) * (new_p - p_3D)));
    put(Phi_tr, 3, 3, M3::eye());
    put(Phi_ns, 0, 0, dt * M3::eye());
    put(Phi_ns, 3, 3, dt * T(R_3D));
    Cov_3D = (Phi_tr * Cov_3D) * T(Phi_tr) + (Phi_ns * Q) * T(Phi_ns);
    Cov_3D = 0.5 * (Cov_3D + T(Cov_3D));
    R_3D = R_new;
    p_3D = new_p;
  }
  // compute_linear_system_3D :327-424
  V3 pI0 = v3(st->p0), pI1 = v3(st->p1);
  M3 RG0 = m3(st->R0), RG1 = m3(st->R1);
  const V3 pIinO = v3(st->p_IinO);
  const M3 RItoO = m3(st->R_ItoO);
  const V3 pOinI = (-1.0 * T(RItoO)) * pIinO;
  M3 RO0toO1 = ((RItoO * RG1) * T(RG0)) * T(RItoO);
  const V3 r_ori = -1.0 * log_so3(R_3D * T(RO0toO1));
  const V3 p_est = (RItoO * RG0) * (((pI1 + T(RG1) * pOinI) - pI0) - T(RG0) * pOinI);
  const V3 r_pos = p_3D - p_est;
  for (int i = 0; i < 3; ++i) {
    res[i] = r_ori.a[i];
    res[3 + i] = r_pos.a[i];
  }
  const int k = 12 + (op->do_calib_ext ? 6 : 0) + (op->do_calib_dt ? 1 : 0) + (op->do_calib_int ? 3 : 0);
  std::vector<double> Hr((size_t)6 * k, 0.0);  // row-major scratch
  auto putH = [&](int r0, int c0, const M3 &B) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Hr[(size_t)(r0 + r) * k + c0 + c] = B(r, c);
  };
  pI0 = v3(st->p0_fej), pI1 = v3(st->p1_fej), RG0 = m3(st->R0_fej), RG1 = m3(st->R1_fej);
  RO0toO1 = ((RItoO * RG1) * T(RG0)) * T(RItoO);
  const M3 RO1toO0 = T(RO0toO1);
  const M3 dzr_dth0 = ((-1.0 * RItoO) * RG1) * T(RG0), dzr_dth1 = RItoO;
  const M3 dzp_dth0 = RItoO * skew((RG0 * pI1 + (RG0 * T(RG1)) * pOinI) - RG0 * pI0);
  const M3 dzp_dp0 = (-1.0 * RItoO) * RG0;
  const M3 dzp_dth1 = (((-1.0 * RItoO) * RG0) * T(RG1)) * skew(pOinI);
  const M3 dzp_dp1 = RItoO * RG0;
  putH(0, 0, dzr_dth0), putH(0, 6, dzr_dth1), putH(3, 0, dzp_dth0), putH(3, 3, dzp_dp0), putH(3, 6, dzp_dth1), putH(3, 9, dzp_dp1);
  int hc = 12, nc = 0;
  for (int i = 0; i < 6; ++i) col_to_state[nc++] = st->pose0_id + i;
  for (int i = 0; i < 6; ++i) col_to_state[nc++] = st->pose1_id + i;
  if (op->do_calib_ext) {
    putH(0, hc, M3::eye() - RO0toO1);
    putH(3, hc, skew((RItoO * RG0) * (pI1 - pI0) - RO1toO0 * pIinO) + RO1toO0 * skew(pIinO));
    putH(3, hc + 3, (-1.0 * RO1toO0) + M3::eye());
    for (int i = 0; i < 6; ++i) col_to_state[nc++] = st->ext_id + i;
    hc += 6;
  }
  if (op->do_calib_dt) {
    const V3 w0 = v3(st->w0), v0 = v3(st->v0), w1 = v3(st->w1), v1 = v3(st->v1);
    const V3 a = dzr_dth0 * w0 + dzr_dth1 * w1;
    const V3 c = ((dzp_dth0 * w0 + dzp_dp0 * v0) + dzp_dth1 * w1) + dzp_dp1 * v1;
    for (int r = 0; r < 3; ++r) {
      Hr[(size_t)r * k + hc] = a.a[r];
      Hr[(size_t)(3 + r) * k + hc] = c.a[r];
    }
    col_to_state[nc++] = st->dt_id;
    hc += 1;
  }
  if (op->do_calib_int) {
    putH(0, hc, -1.0 * dR_di);
    putH(3, hc, -1.0 * dp_di);
    for (int i = 0; i < 3; ++i) col_to_state[nc++] = st->intr_id + i;
  }
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < k; ++c) H[(size_t)c * 6 + r] = Hr[(size_t)r * k + c];
  std::memcpy(Cov, Cov_3D.a, sizeof(Cov_3D.a));
  if (R_out) std::memcpy(R_out, R_3D.a, 72);
  if (p_out) std::memcpy(p_out, p_3D.a, 24);
  return k;
}

// The 2D types: preintegration_2D (:502-646), preintegration_intrinsics_2D (:426-470), compute_linear_system_2D (:217-325).
// H is 3 x k col-major.
int orc_wheel_linear_system_2d(const plv_wheel_options *op, const plv_wheel_state *st, int n_data, const double *t, const double *m1,
                               const double *m2, double *H, double *res, double *Cov, int *col_to_state, double *meas) {
  double th_2D = 0, x_2D = 0, y_2D = 0;
  M3 C2 = M3::zero();
  Mat<1, 3> dth_di = Mat<1, 3>::zero(), dx_di = Mat<1, 3>::zero(), dy_di = Mat<1, 3>::zero();
  const double rl = st->intr[0], rr = st->intr[1], b = st->intr[2];
  for (int i = 0; i < n_data - 1; ++i) {
    const double dt = t[i + 1] - t[i];
    if (op->do_calib_int) {
      const double w_l = m1[i], w_r = m2[i];
      const double w = (w_r * rr - w_l * rl) / b, v = (w_r * rr + w_l * rl) / 2;
      const Mat<1, 3> Hwx{{-w_l / b, w_r / b, -(w_r * rr - w_l * rl) / (b * b)}}, Hvx{{w_l / 2, w_r / 2, 0}};
      double h_thw = dt;
      double h_xth = (v * (std::cos(th_2D - w * dt) - std::cos(th_2D))) / w;
      double h_yth = -(v * (std::sin(th_2D - w * dt) - std::sin(th_2D))) / w;
      double h_xw = (v * (std::sin(th_2D - w * dt) - std::sin(th_2D))) / w / w + (v * std::cos(th_2D - w * dt) * dt) / w;
      double h_yw = (v * (std::cos(th_2D - w * dt) - std::cos(th_2D))) / w / w - (v * std::sin(th_2D - w * dt) * dt) / w;
      double h_xv = -(std::sin(th_2D - w * dt) - std::sin(th_2D)) / w;
      double h_yv = -(std::cos(th_2D - w * dt) - std::cos(th_2D)) / w;
      if (std::fabs(w) < 0.0001) {
        h_xth = v * std::sin(th_2D) * dt;
        h_yth = v * std::cos(th_2D) * dt;
        h_xw = v * std::sin(th_2D) * dt * dt / 2;
        h_yw = v * std::cos(th_2D) * dt * dt / 2;
        h_xv = std::cos(th_2D) * dt;
        h_yv = -std::sin(th_2D) * dt;
      }
      dx_di = ((dx_di + h_xth * dth_di) + h_xw * Hwx) + h_xv * Hvx;
      dy_di = ((dy_di + h_yth * dth_di) + h_yw * Hwx) + h_yv * Hvx;
      dth_di = dth_di + h_thw * Hwx;
    }
    double w1, w2, v1, v2;
    if (op->type == PLV_WHEEL2D_ANG) {
      w1 = (m2[i] * rr - m1[i] * rl) / b, v1 = (m2[i] * rr + m1[i] * rl) / 2;
      w2 = (m2[i + 1] * rr - m1[i + 1] * rl) / b, v2 = (m2[i + 1] * rr + m1[i + 1] * rl) / 2;
    } else if (op->type == PLV_WHEEL2D_LIN) {
      w1 = (m2[i] - m1[i]) / b, v1 = (m2[i] + m1[i]) / 2;
      w2 = (m2[i + 1] - m1[i + 1]) / b, v2 = (m2[i + 1] + m1[i + 1]) / 2;
    } else {
      w1 = m1[i], v1 = m2[i], w2 = m1[i + 1], v2 = m2[i + 1];
    }
    const double w_alpha = (w2 - w1) / dt, v_jerk = (v2 - v1) / dt;
    double w = w1, v = v1;
    const double k1_th = -w * dt, k1_x = v * 1 * dt;
    const double th2 = 0.5 * k1_th;
    w += 0.5 * w_alpha * dt;
    v += 0.5 * v_jerk * dt;
    const double k2_th = -w * dt, k2_x = v * std::cos(th2) * dt;
    const double th3 = 0.5 * k2_th;
    const double k3_th = -w * dt, k3_x = v * std::cos(th3) * dt;
    const double th4 = k3_th;
    w += 0.5 * w_alpha * dt;
    v += 0.5 * v_jerk * dt;
    const double k4_th = -w * dt, k4_x = v * std::cos(th4) * dt;
    const double th_next = th_2D + (1.0 / 6.0) * (k1_th + 2 * k2_th + 2 * k3_th + k4_th);
    const double x_next = x_2D + (1.0 / 6.0) * (k1_x + 2 * k2_x + 2 * k3_x + k4_x);
    double y_next;  // the RK4 value is overwritten by the closed form (:568-571)
    if (std::fabs(w1) < 0.0001)
      y_next = y_2D - v1 * std::sin(th_2D - w1 * dt) * dt;
    else
      y_next = y_2D - (v1 * (std::cos(th_2D - w1 * dt) - std::cos(th_2D))) / w1;
    Mat<1, 2> Hwn, Hvn;
    if (op->type == PLV_WHEEL2D_ANG) {
      Hwn = Mat<1, 2>{{rl / b, -rr / b}};
      Hvn = Mat<1, 2>{{-rl / 2, -rr / 2}};
    } else if (op->type == PLV_WHEEL2D_LIN) {
      Hwn = Mat<1, 2>{{1.0 / b, -1.0 / b}};
      Hvn = Mat<1, 2>{{-1.0 / 2, -1.0 / 2}};
    } else {
      Hwn = Mat<1, 2>{{1, 0}};
      Hvn = Mat<1, 2>{{0, 1}};
    }
    double h_thw = dt;
    double h_xth = (v1 * (std::cos(th_2D - w1 * dt) - std::cos(th_2D))) / w1;
    double h_yth = -(v1 * (std::sin(th_2D - w1 * dt) - std::sin(th_2D))) / w1;
    double h_xw = (v1 * (std::sin(th_2D - w1 * dt) - std::sin(th_2D))) / w1 / w1 + (v1 * std::cos(th_2D - w1 * dt) * dt) / w1;
    double h_yw = (v1 * (std::cos(th_2D - w1 * dt) - std::cos(th_2D))) / w1 / w1 - (v1 * std::sin(th_2D - w1 * dt) * dt) / w1;
    double h_xv = -(std::sin(th_2D - w1 * dt) - std::sin(th_2D)) / w1;
    double h_yv = -(std::cos(th_2D - w1 * dt) - std::cos(th_2D)) / w1;
    if (std::fabs(w1) < 0.0001) {
      h_xth = v1 * std::sin(th_2D) * dt;
      h_yth = v1 * std::cos(th_2D) * dt;
      h_xw = v1 * std::sin(th_2D) * dt * dt / 2;
      h_yw = v1 * std::cos(th_2D) * dt * dt / 2;
      h_xv = std::cos(th_2D) * dt;
      h_yv = -std::sin(th_2D) * dt;
    }
    M3 Phi_tr = M3::eye();
    Phi_tr(1, 0) = h_xth;
    Phi_tr(2, 0) = h_yth;
    Mat<3, 2> Phi_ns = Mat<3, 2>::zero();
    const Mat<1, 2> r0 = h_thw * Hwn, r1 = h_xw * Hwn + h_xv * Hvn, r2 = h_yw * Hwn + h_yv * Hvn;
    for (int c = 0; c < 2; ++c) Phi_ns(0, c) = r0.a[c], Phi_ns(1, c) = r1.a[c], Phi_ns(2, c) = r2.a[c];
    Mat<2, 2> Q = Mat<2, 2>::zero();
    if (op->type == PLV_WHEEL2D_ANG)
      Q(0, 0) = Q(1, 1) = op->noise_w * op->noise_w / dt;
    else if (op->type == PLV_WHEEL2D_LIN)
      Q(0, 0) = Q(1, 1) = op->noise_v * op->noise_v / dt;
    else
      Q(0, 0) = op->noise_w * op->noise_w / dt, Q(1, 1) = op->noise_v * op->noise_v / dt;
    C2 = (Phi_tr * C2) * T(Phi_tr) + (Phi_ns * Q) * T(Phi_ns);
    C2 = 0.5 * (C2 + T(C2));
    th_2D = th_next, x_2D = x_next, y_2D = y_next;
  }
  // compute_linear_system_2D
  V3 pI0 = v3(st->p0), pI1 = v3(st->p1);
  M3 RG0 = m3(st->R0), RG1 = m3(st->R1);
  const V3 pIinO = v3(st->p_IinO);
  const M3 RItoO = m3(st->R_ItoO);
  const V3 pOinI = (-1.0 * T(RItoO)) * pIinO;
  const double theta_est = log_so3(((RItoO * RG1) * T(RG0)) * T(RItoO)).a[2];
  res[0] = theta_est - th_2D;
  const V3 d_est3 = (RItoO * RG0) * (((pI1 + T(RG1) * pOinI) - pI0) - T(RG0) * pOinI);
  res[1] = x_2D - d_est3.a[0];
  res[2] = y_2D - d_est3.a[1];
  const int k = 12 + (op->do_calib_ext ? 6 : 0) + (op->do_calib_dt ? 1 : 0) + (op->do_calib_int ? 3 : 0);
  std::vector<double> Hr((size_t)3 * k, 0.0);
  pI0 = v3(st->p0_fej), pI1 = v3(st->p1_fej), RG0 = m3(st->R0_fej), RG1 = m3(st->R1_fej);
  const M3 RO0toO1 = ((RItoO * RG1) * T(RG0)) * T(RItoO), RO1toO0 = T(RO0toO1);
  const M3 A0 = ((-1.0 * RItoO) * RG1) * T(RG0);   // row 2 = -e3^T RItoO RG1 RG0^T
  const M3 Pth0 = RItoO * skew(RG0 * ((pI1 + T(RG1) * pOinI) - pI0));
  const M3 Pp0 = (-1.0 * RItoO) * RG0;
  const M3 Pth1 = (((-1.0 * RItoO) * RG0) * T(RG1)) * skew(pOinI);
  const M3 Pp1 = RItoO * RG0;
  auto row_of = [&](int r, int c0, const M3 &B, int br) {
    for (int c = 0; c < 3; ++c) Hr[(size_t)r * k + c0 + c] = B(br, c);
  };
  row_of(0, 0, A0, 2);
  row_of(0, 6, RItoO, 2);
  for (int r = 0; r < 2; ++r) {
    row_of(1 + r, 0, Pth0, r);
    row_of(1 + r, 3, Pp0, r);
    row_of(1 + r, 6, Pth1, r);
    row_of(1 + r, 9, Pp1, r);
  }
  int hc = 12, nc = 0;
  for (int i = 0; i < 6; ++i) col_to_state[nc++] = st->pose0_id + i;
  for (int i = 0; i < 6; ++i) col_to_state[nc++] = st->pose1_id + i;
  if (op->do_calib_ext) {
    row_of(0, hc, M3::eye() - RO0toO1, 2);
    const M3 Dth = skew((RItoO * RG0) * (pI1 - pI0) - RO1toO0 * pIinO) + RO1toO0 * skew(pIinO);
    const M3 Dp = (-1.0 * RO1toO0) + M3::eye();
    for (int r = 0; r < 2; ++r) {
      row_of(1 + r, hc, Dth, r);
      row_of(1 + r, hc + 3, Dp, r);
    }
    for (int i = 0; i < 6; ++i) col_to_state[nc++] = st->ext_id + i;
    hc += 6;
  }
  if (op->do_calib_dt) {
    const V3 w0 = v3(st->w0), v0 = v3(st->v0), w1 = v3(st->w1), v1 = v3(st->v1);
    const V3 a = A0 * w0 + RItoO * w1;
    const V3 c = ((Pth0 * w0 + Pp0 * v0) + Pth1 * w1) + Pp1 * v1;
    Hr[(size_t)0 * k + hc] = a.a[2];
    Hr[(size_t)1 * k + hc] = c.a[0];
    Hr[(size_t)2 * k + hc] = c.a[1];
    col_to_state[nc++] = st->dt_id;
    hc += 1;
  }
  if (op->do_calib_int) {
    for (int c = 0; c < 3; ++c) {
      Hr[(size_t)0 * k + hc + c] = -dth_di.a[c];
      Hr[(size_t)1 * k + hc + c] = -dx_di.a[c];
      Hr[(size_t)2 * k + hc + c] = -dy_di.a[c];
    }
    for (int i = 0; i < 3; ++i) col_to_state[nc++] = st->intr_id + i;
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < k; ++c) H[(size_t)c * 3 + r] = Hr[(size_t)r * k + c];
  std::memcpy(Cov, C2.a, sizeof(C2.a));
  meas[0] = th_2D, meas[1] = x_2D, meas[2] = y_2D;
  return k;
}

// StateHelper::clone: append `size` rows / columns copying the block at src_id.  P has room for (n + size) (ld >= n + size).
void orc_cov_clone(double *P, int n, int ld, int src_id, int size) {
  for (int c = 0; c < size; ++c)
    for (int r = 0; r < n; ++r) {
      P[(size_t)(n + c) * ld + r] = P[(size_t)(src_id + c) * ld + r];  // (0, new_loc) block
      P[(size_t)r * ld + n + c] = P[(size_t)r * ld + src_id + c];      // (new_loc, 0) block
    }
  for (int c = 0; c < size; ++c)
    for (int r = 0; r < size; ++r) P[(size_t)(n + c) * ld + n + r] = P[(size_t)(src_id + c) * ld + src_id + r];
}

}  // extern "C"
