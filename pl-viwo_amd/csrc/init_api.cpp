// init_api.cpp — state initialisation behind the C-ABI: the static IMU initialiser and the
// IMU-wheel initialiser.  A few hundred samples of scalar arithmetic once per run: host code, no kernel.
//   REF: PL-VIWO/src/init/imu/I_Initializer.cpp:44-150, PL-VIWO/src/init/imu_wheel/IW_Initializer.cpp:44-690,
//        PL-VIWO/src/init/Initializer.cpp:93-113 (caller), open_vins/ov_core/src/utils/quat_ops.h.
#include <algorithm>
#include <array>
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

#include "../../include/plviwo.h"

namespace {

using V3 = std::array<double, 3>;
using M3 = std::array<double, 9>;  // row-major
using Q4 = std::array<double, 4>;  // JPL x y z w

V3 add(V3 a, V3 b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
V3 sub(V3 a, V3 b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
V3 scl(double s, V3 a) { return {s * a[0], s * a[1], s * a[2]}; }
double dot(V3 a, V3 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
double nrm(V3 a) { return std::sqrt(dot(a, a)); }
V3 cross(V3 a, V3 b) { return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; }
V3 ld(const double *p) { return {p[0], p[1], p[2]}; }
M3 eye() { return {1, 0, 0, 0, 1, 0, 0, 0, 1}; }
M3 mm(const M3 &a, const M3 &b) {
  M3 c;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  return c;
}
M3 tr(const M3 &a) { return {a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8]}; }
V3 mv(const M3 &a, V3 v) {
  return {a[0] * v[0] + a[1] * v[1] + a[2] * v[2], a[3] * v[0] + a[4] * v[1] + a[5] * v[2], a[6] * v[0] + a[7] * v[1] + a[8] * v[2]};
}
M3 madd(const M3 &a, const M3 &b) {
  M3 c;
  for (int i = 0; i < 9; ++i) c[i] = a[i] + b[i];
  return c;
}
M3 mscl(double s, const M3 &a) {
  M3 c;
  for (int i = 0; i < 9; ++i) c[i] = s * a[i];
  return c;
}

M3 quat_2_Rot(Q4 q) {  // quat_ops.h:152-157
  const double x = q[0], y = q[1], z = q[2], w = q[3], s = 2 * w * w - 1;
  return {s + 2 * x * x,         2 * w * z + 2 * x * y,  -2 * w * y + 2 * x * z,
          -2 * w * z + 2 * x * y, s + 2 * y * y,          2 * w * x + 2 * y * z,
          2 * w * y + 2 * x * z,  -2 * w * x + 2 * y * z, s + 2 * z * z};
}

Q4 rot_2_quat(const M3 &R) {  // quat_ops.h:88-130
  const double T = R[0] + R[4] + R[8];
  Q4 q;
  if (R[0] >= T && R[0] >= R[4] && R[0] >= R[8]) {
    q[0] = std::sqrt((1 + 2 * R[0] - T) / 4);
    q[1] = (1 / (4 * q[0])) * (R[1] + R[3]);
    q[2] = (1 / (4 * q[0])) * (R[2] + R[6]);
    q[3] = (1 / (4 * q[0])) * (R[5] - R[7]);
  } else if (R[4] >= T && R[4] >= R[0] && R[4] >= R[8]) {
    q[1] = std::sqrt((1 + 2 * R[4] - T) / 4);
    q[0] = (1 / (4 * q[1])) * (R[1] + R[3]);
    q[2] = (1 / (4 * q[1])) * (R[5] + R[7]);
    q[3] = (1 / (4 * q[1])) * (R[6] - R[2]);
  } else if (R[8] >= T && R[8] >= R[0] && R[8] >= R[4]) {
    q[2] = std::sqrt((1 + 2 * R[8] - T) / 4);
    q[0] = (1 / (4 * q[2])) * (R[2] + R[6]);
    q[1] = (1 / (4 * q[2])) * (R[5] + R[7]);
    q[3] = (1 / (4 * q[2])) * (R[1] - R[3]);
  } else {
    q[3] = std::sqrt((1 + T) / 4);
    q[0] = (1 / (4 * q[3])) * (R[5] - R[7]);
    q[1] = (1 / (4 * q[3])) * (R[6] - R[2]);
    q[2] = (1 / (4 * q[3])) * (R[1] - R[3]);
  }
  if (q[3] < 0) q = {-q[0], -q[1], -q[2], -q[3]};
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  return {q[0] / n, q[1] / n, q[2] / n, q[3] / n};
}

Q4 quatnorm(Q4 q) {  // quat_ops.h:496-501
  if (q[3] < 0) q = {-q[0], -q[1], -q[2], -q[3]};
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  return {q[0] / n, q[1] / n, q[2] / n, q[3] / n};
}

Q4 half_omega_times(V3 w, Q4 q, double dt) {  // 0.5 * Omega(w) * q * dt, Omega of quat_ops.h:482-489
  const V3 v{q[0], q[1], q[2]};
  const V3 c = cross(w, v);
  return {0.5 * dt * (-c[0] + w[0] * q[3]), 0.5 * dt * (-c[1] + w[1] * q[3]), 0.5 * dt * (-c[2] + w[2] * q[3]), 0.5 * dt * (-dot(w, v))};
}

Q4 axpy(Q4 a, double s, Q4 b) { return {a[0] + s * b[0], a[1] + s * b[1], a[2] + s * b[2], a[3] + s * b[3]}; }

// IW_Initializer::IMU_prop_rk4 (REF: IW_Initializer.cpp:608-641): the rotation over dt for an angular velocity that changes
// linearly from w1 to w2, RK4 on the JPL quaternion starting from the identity.
Q4 prop_rk4(double dt, V3 w1, V3 w2) {
  V3 w = w1;
  const V3 alpha = scl(1.0 / dt, sub(w2, w1));
  const Q4 dq0{0, 0, 0, 1};
  const Q4 k1 = half_omega_times(w, dq0, dt);
  w = add(w, scl(0.5 * dt, alpha));
  const Q4 k2 = half_omega_times(w, quatnorm(axpy(dq0, 0.5, k1)), dt);
  const Q4 k3 = half_omega_times(w, quatnorm(axpy(dq0, 0.5, k2)), dt);
  w = add(w, scl(0.5 * dt, alpha));
  const Q4 k4 = half_omega_times(w, quatnorm(axpy(dq0, 1.0, k3)), dt);
  Q4 r = dq0;
  r = axpy(r, 1.0 / 6.0, k1), r = axpy(r, 1.0 / 3.0, k2), r = axpy(r, 1.0 / 3.0, k3), r = axpy(r, 1.0 / 6.0, k4);
  return quatnorm(r);
}

// The rotation whose third column is the direction of `g` (REF: IW_Initializer.cpp:643-681, I_Initializer.cpp:117-135).
M3 gram_schmidt(V3 g, bool normalise_y) {
  const V3 z = scl(1.0 / nrm(g), g);
  V3 x = sub(V3{1, 0, 0}, scl(z[0], z));
  x = scl(1.0 / nrm(x), x);
  V3 y = cross(z, x);
  if (normalise_y) y = scl(1.0 / nrm(y), y);
  return {x[0], y[0], z[0], x[1], y[1], z[1], x[2], y[2], z[2]};
}

struct Imu {
  std::vector<double> t, w, a;  // w / a [n][3]
  size_t size() const { return t.size(); }
};

bool select_imu(const Imu &buf, double t0, double t1, Imu &out) {
  const int n = (int)buf.size();
  out.t.assign(n + 2, 0), out.w.assign(3 * (n + 2), 0), out.a.assign(3 * (n + 2), 0);
  int m = 0, ok = 0;
  if (plv_select_imu_readings(n, buf.t.data(), buf.w.data(), buf.a.data(), t0, t1, n + 2, out.t.data(), out.w.data(), out.a.data(), &m, &ok) != PLV_OK || !ok)
    return false;
  out.t.resize(m), out.w.resize(3 * m), out.a.resize(3 * m);
  return true;
}

struct Wheel {  // one converted reading: time, angular and linear velocity of the odometry frame
  double t;
  V3 w, v;
};

struct IW {
  const plv_iw_init_options &op;
  const Imu &all;  // Propagator::imu_data
  M3 R_OtoI;
  V3 p_IinO, grav;
  double toff;

  // The quantities every stage accumulates over the wheel intervals (REF: IW_Initializer.cpp:220-262 and its copies at
  // :286-326, :444-486, :500-545): returns false where the reference's assert(success) would fire.
  template <class F>
  bool walk(V3 bg, const std::vector<Wheel> &wh, F &&per_interval) const {
    double sum_dt = 0;
    V3 sum_R_a_dt{0, 0, 0};
    M3 sum_R_dt{0, 0, 0, 0, 0, 0, 0, 0, 0}, R_IktoI0 = eye(), R_O0toOk = eye();
    for (size_t i = 1; i < wh.size(); ++i) {
      const double ts = wh[i - 1].t + toff, te = wh[i].t + toff;
      Imu pr;
      if (!select_imu(all, ts, te, pr)) return false;
      for (size_t j = 0; j + 1 < pr.size(); ++j) {
        const double dt = pr.t[j + 1] - pr.t[j];
        const V3 w0 = sub(ld(&pr.w[3 * j]), bg), w1 = sub(ld(&pr.w[3 * j + 3]), bg);
        const V3 a = scl(0.5, add(ld(&pr.a[3 * j]), ld(&pr.a[3 * j + 3])));
        sum_R_a_dt = add(sum_R_a_dt, scl(dt, mv(R_IktoI0, a)));
        sum_R_dt = madd(sum_R_dt, mscl(dt, R_IktoI0));
        sum_dt += dt;
        R_IktoI0 = mm(R_IktoI0, tr(quat_2_Rot(prop_rk4(dt, w0, w1))));
      }
      R_O0toOk = mm(quat_2_Rot(prop_rk4(te - ts, wh[i - 1].w, wh[i].w)), R_O0toOk);
      const V3 v_ItinI0 = mv(R_OtoI, mv(tr(R_O0toOk), add(wh[i].v, cross(wh[i].w, p_IinO))));
      per_interval(i, sum_dt, sum_R_a_dt, sum_R_dt, v_ItinI0);
    }
    return true;
  }
};

bool inv3(const M3 &a, M3 &o) {
  const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
  const double det = a[0] * c0 + a[1] * c1 + a[2] * c2;
  if (det == 0 || !std::isfinite(det)) return false;
  const double s = 1.0 / det;
  o = {s * c0, s * (a[2] * a[7] - a[1] * a[8]), s * (a[1] * a[5] - a[2] * a[4]),
       s * c1, s * (a[0] * a[8] - a[2] * a[6]), s * (a[2] * a[3] - a[0] * a[5]),
       s * c2, s * (a[1] * a[6] - a[0] * a[7]), s * (a[0] * a[4] - a[1] * a[3])};
  return true;
}

// Eigen's LLT (unblocked, lower) followed by solve(), including what it does on a matrix that is not positive definite: the
// factorisation stops at the first non-positive pivot and solve() runs on what is in the matrix at that point.  The
// reference evaluates every real root of the constraint polynomial through it (REF: IW_Initializer.cpp:393-394, 421).
V3 llt_solve(M3 A, V3 b) {
  for (int k = 0; k < 3; ++k) {
    double x = A[4 * k];
    for (int j = 0; j < k; ++j) x -= A[3 * k + j] * A[3 * k + j];
    if (x <= 0) break;
    A[4 * k] = x = std::sqrt(x);
    for (int i = k + 1; i < 3; ++i) {
      for (int j = 0; j < k; ++j) A[3 * i + k] -= A[3 * i + j] * A[3 * k + j];
      A[3 * i + k] /= x;
    }
  }
  V3 y;  // L y = b, L^T x = y on the lower triangle
  for (int i = 0; i < 3; ++i) {
    double s = b[i];
    for (int j = 0; j < i; ++j) s -= A[3 * i + j] * y[j];
    y[i] = s / A[4 * i];
  }
  for (int i = 2; i >= 0; --i) {
    double s = y[i];
    for (int j = i + 1; j < 3; ++j) s -= A[3 * j + i] * y[j];
    y[i] = s / A[4 * i];
  }
  return y;
}

using Poly = std::vector<double>;  // ascending powers
Poly pmul(const Poly &a, const Poly &b) {
  Poly c(a.size() + b.size() - 1, 0.0);
  for (size_t i = 0; i < a.size(); ++i)
    for (size_t j = 0; j < b.size(); ++j) c[i + j] += a[i] * b[j];
  return c;
}
Poly padd(const Poly &a, const Poly &b, double sb = 1.0) {
  Poly c(std::max(a.size(), b.size()), 0.0);
  for (size_t i = 0; i < a.size(); ++i) c[i] += a[i];
  for (size_t i = 0; i < b.size(); ++i) c[i] += sb * b[i];
  return c;
}

// The degree-6 polynomial in lambda whose real roots satisfy |(D - lambda I)^-1 d| = g (REF: IW_Initializer.cpp:683-690
// compute_dongsi_coeff, a generated closed form there): g^2 det(M)^2 - |adj(M) d|^2 with M = D - lambda I, built by polynomial
// arithmetic and scaled to a leading coefficient of 1.
Poly dongsi_poly(const M3 &D, V3 d, double g) {
  Poly M[9];
  for (int i = 0; i < 9; ++i) M[i] = (i % 4 == 0) ? Poly{D[i], -1.0} : Poly{D[i]};
  auto minor2 = [&](int a, int b, int c, int e) { return padd(pmul(M[a], M[e]), pmul(M[b], M[c]), -1.0); };
  // adjugate (transposed cofactors), row-major
  Poly adj[9] = {minor2(4, 5, 7, 8), minor2(2, 1, 8, 7), minor2(1, 2, 4, 5),
                 minor2(5, 3, 8, 6), minor2(0, 2, 6, 8), minor2(2, 0, 5, 3),
                 minor2(3, 4, 6, 7), minor2(1, 0, 7, 6), minor2(0, 1, 3, 4)};
  Poly det = padd(padd(pmul(M[0], adj[0]), pmul(M[1], adj[3])), pmul(M[2], adj[6]));
  Poly p = pmul(det, det);
  for (auto &c : p) c *= g * g;
  for (int i = 0; i < 3; ++i) {
    Poly r = padd(padd(Poly{0.0}, adj[3 * i], d[0]), padd(Poly{0.0}, adj[3 * i + 1], d[1]));
    r = padd(r, adj[3 * i + 2], d[2]);
    p = padd(p, pmul(r, r), -1.0);
  }
  p.resize(7, 0.0);
  const double lead = p[6];
  for (auto &c : p) c /= lead;
  return p;
}

// Real roots of a monic polynomial: Aberth-Ehrlich iterations on all roots, then Newton in real arithmetic on those whose
// imaginary part vanished (the reference takes the eigenvalues of the companion matrix whose imaginary part is exactly 0).
std::vector<double> real_roots(const Poly &p) {
  using cd = std::complex<double>;
  const int n = (int)p.size() - 1;
  auto eval = [&](cd z, cd &dz) {
    cd v = p[n];
    dz = 0;
    for (int i = n - 1; i >= 0; --i) dz = dz * z + v, v = v * z + p[i];
    return v;
  };
  double rad = 0;
  for (int i = 0; i < n; ++i) rad = std::max(rad, std::abs(p[i]));
  rad = 1 + rad;  // Cauchy bound
  double lo = 0;
  for (int i = 0; i < n; ++i) lo = std::max(lo, std::pow(std::abs(p[i]), 1.0 / (n - i)));
  rad = std::min(rad, 2 * lo + 1e-300);
  std::vector<cd> z(n);
  for (int i = 0; i < n; ++i) z[i] = std::polar(rad * (0.4 + 0.6 * (i + 1) / n), 2 * M_PI * i / n + 0.7);
  for (int it = 0; it < 400; ++it) {
    double moved = 0;
    for (int i = 0; i < n; ++i) {
      cd dz, v = eval(z[i], dz);
      if (v == cd(0)) continue;
      cd r = v / dz, s = 0;
      for (int j = 0; j < n; ++j)
        if (j != i) s += cd(1) / (z[i] - z[j]);
      const cd w = r / (cd(1) - r * s);
      z[i] -= w;
      moved = std::max(moved, std::abs(w) / std::max(1e-300, std::abs(z[i])));
    }
    if (moved < 1e-15) break;
  }
  std::vector<double> out;
  for (int i = 0; i < n; ++i) {
    if (std::abs(z[i].imag()) > 1e-7 * std::max(1.0, std::abs(z[i]))) continue;
    double x = z[i].real();
    for (int it = 0; it < 20; ++it) {
      double v = p[n], dv = 0;
      for (int k = n - 1; k >= 0; --k) dv = dv * x + v, v = v * x + p[k];
      if (dv == 0) break;
      const double step = v / dv;
      x -= step;
      if (std::abs(step) <= 1e-16 * std::abs(x)) break;
    }
    out.push_back(x);
  }
  std::sort(out.begin(), out.end());
  return out;
}

}  // namespace

extern "C" {

// ov_type::JPLQuat::update for a stack of orientations (REF: open_vins/ov_core/src/types/JPLQuat.h:62-73, utils/quat_ops.h:152-157,
// 232-252): q <- quatnorm([dth / 2, 1]) (x) q, w >= 0, renormalised; R (nullable) receives quat_2_Rot(q) row-major.  Host arithmetic
// for the driver's dx application (StateHelper::EKFUpdate :156-160 calls it per variable).
void plv_jpl_left_update(int n, double *q, const double *dth, double *R) {
  for (int k = 0; k < n; ++k) {
    double *Q = q + 4 * (size_t)k;
    if (dth) {
      const double *d = dth + 3 * (size_t)k;
      double a[3] = {0.5 * d[0], 0.5 * d[1], 0.5 * d[2]}, b = 1.0;
      const double nd = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + 1.0);
      a[0] /= nd, a[1] /= nd, a[2] /= nd, b /= nd;
      const double v[3] = {Q[0], Q[1], Q[2]}, w = Q[3];
      double r[4];
      r[0] = b * v[0] - (a[1] * v[2] - a[2] * v[1]) + a[0] * w;
      r[1] = b * v[1] - (a[2] * v[0] - a[0] * v[2]) + a[1] * w;
      r[2] = b * v[2] - (a[0] * v[1] - a[1] * v[0]) + a[2] * w;
      r[3] = -(a[0] * v[0] + a[1] * v[1] + a[2] * v[2]) + b * w;
      if (r[3] < 0)
        for (double &x : r) x = -x;
      const double nr = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
      for (int i = 0; i < 4; ++i) Q[i] = r[i] / nr;
    }
    if (R) {
      const double x = Q[0], y = Q[1], z = Q[2], w = Q[3], c = 2 * w * w - 1;
      double *M = R + 9 * (size_t)k;
      // (2 w^2 - 1) I - 2 w [q x] + 2 q q^T
      M[0] = c + 2 * x * x, M[1] = 2 * w * z + 2 * x * y, M[2] = -2 * w * y + 2 * x * z;
      M[3] = -2 * w * z + 2 * y * x, M[4] = c + 2 * y * y, M[5] = 2 * w * x + 2 * y * z;
      M[6] = 2 * w * y + 2 * z * x, M[7] = -2 * w * x + 2 * z * y, M[8] = c + 2 * z * z;
    }
  }
}

// x <- x [+] dx for every variable of the state in one call (REF: StateHelper::EKFUpdate, StateHelper.cpp:156-160, calls
// Type::update(dx.block(id, 0, size, 1)) per variable: Vec adds, JPLQuat composes on the left, PoseJPL is one of each).
// `out` of a quaternion receives its rotation matrix; `mirror` (nullable) a second copy of what the variable now holds (the value of
// a vector, the rotation matrix of a quaternion) — e.g. the field of a plv_state_view the caller keeps current.
int plv_state_boxplus(int n_var, const plv_state_var *vars, const double *dx, int n_dx) {
  if (n_var < 0 || (n_var > 0 && !vars) || !dx) return PLV_E_BADARG;
  for (int i = 0; i < n_var; ++i) {  // every entry is checked before the first one is applied: a bad list leaves the state as it was
    const plv_state_var &v = vars[i];
    if (v.kind != PLV_VAR_QUAT && v.kind != PLV_VAR_VEC) return PLV_E_BADARG;
    if (!v.val || v.id < 0 || v.size < 1 || v.id + (v.kind == PLV_VAR_QUAT ? 3 : v.size) > n_dx) return PLV_E_BADARG;
  }
  for (int i = 0; i < n_var; ++i) {
    const plv_state_var &v = vars[i];
    if (v.kind == PLV_VAR_QUAT) {
      double R[9];
      plv_jpl_left_update(1, v.val, dx + v.id, v.out || v.mirror ? R : nullptr);
      if (v.out) std::copy(R, R + 9, v.out);
      if (v.mirror) std::copy(R, R + 9, v.mirror);
    } else if (v.kind == PLV_VAR_VEC) {
      for (int j = 0; j < v.size; ++j) v.val[j] = v.val[j] + dx[v.id + j];
      if (v.mirror) std::copy(v.val, v.val + v.size, v.mirror);
    }
  }
  return PLV_OK;
}


int plv_init_imu_static(int n, const double *t, const double *wm, const double *am, double window_time, double imu_thresh,
                        const double *gravity, double *imustate, int *ok) {
  if (!ok || !imustate || !gravity || n < 0 || (n > 0 && (!t || !wm || !am))) return PLV_E_BADARG;
  *ok = 0;
  if (n < 2) return PLV_OK;  // I_Initializer.cpp:47-49
  const double newest = t[n - 1], oldest = t[0];
  if (newest - oldest < 2 * window_time) return PLV_OK;  // :56-59
  // the two windows (newest - 2W, newest - W] and (newest - W, newest]  :62-70
  int n1 = 0, n2 = 0, last2 = -1;
  V3 a1{0, 0, 0}, a2{0, 0, 0}, w2{0, 0, 0};
  for (int i = 0; i < n; ++i) {
    if (t[i] > newest - window_time && t[i] <= newest) a1 = add(a1, ld(am + 3 * i)), ++n1;
    if (t[i] > newest - 2 * window_time && t[i] <= newest - window_time) a2 = add(a2, ld(am + 3 * i)), w2 = add(w2, ld(wm + 3 * i)), ++n2, last2 = i;
  }
  if (n1 < 2 || n2 < 2) return PLV_OK;  // :73-76
  a1 = scl(1.0 / n1, a1), a2 = scl(1.0 / n2, a2), w2 = scl(1.0 / n2, w2);
  double v1 = 0, v2 = 0;
  for (int i = 0; i < n; ++i) {
    if (t[i] > newest - window_time && t[i] <= newest) v1 += dot(sub(ld(am + 3 * i), a1), sub(ld(am + 3 * i), a1));
    if (t[i] > newest - 2 * window_time && t[i] <= newest - window_time) v2 += dot(sub(ld(am + 3 * i), a2), sub(ld(am + 3 * i), a2));
  }
  v1 = std::sqrt(v1 / (n1 - 1)), v2 = std::sqrt(v2 / (n2 - 1));
  if (v1 < imu_thresh) return PLV_OK;  // no jerk yet            :104-107
  if (v2 > imu_thresh) return PLV_OK;  // was not standing still :111-114
  const M3 Ro = gram_schmidt(a2, false);  // :117-135
  const Q4 q = rot_2_quat(Ro);
  const V3 ba = sub(a2, mv(quat_2_Rot(q), ld(gravity)));
  imustate[0] = t[last2];
  for (int i = 0; i < 4; ++i) imustate[1 + i] = q[i];
  for (int i = 0; i < 6; ++i) imustate[5 + i] = 0;
  for (int i = 0; i < 3; ++i) imustate[11 + i] = w2[i], imustate[14 + i] = ba[i];
  *ok = 1;
  return PLV_OK;
}

void plv_iw_init_reset(plv_iw_init_state *s) {
  if (!s) return;
  std::memset(s, 0, sizeof(*s));
  s->cnt_smooth = -1;  // IW_Initializer.h: int cnt_smooth = -1
}

int plv_init_imu_wheel(const plv_iw_init_options *op, plv_iw_init_state *state, int n_imu, const double *t, const double *wm,
                       const double *am, int n_whl, const double *tw, const double *m1, const double *m2, double *imustate, int *ok,
                       int *mode, double *init12) {
  if (!op || !state || !ok || !imustate || n_imu < 0 || n_whl < 0 || (n_imu > 0 && (!t || !wm || !am)) || (n_whl > 0 && (!tw || !m1 || !m2)))
    return PLV_E_BADARG;
  if (op->wheel_type < PLV_WHEEL3D_ANG || op->wheel_type > PLV_WHEEL2D_CEN) return PLV_E_BADARG;
  *ok = 0;
  if (mode) *mode = -1;
  // ---- get_IMU_Wheel_data (REF: IW_Initializer.cpp:106-167)
  if (n_imu < 3 || n_whl < 3) return PLV_OK;
  const double toff = op->toff;
  const double min_t = std::max(t[1], tw[1] + toff), max_t = std::min(t[n_imu - 2], tw[n_whl - 2] + toff);
  Imu all;
  all.t.assign(t, t + n_imu), all.w.assign(wm, wm + 3 * (size_t)n_imu), all.a.assign(am, am + 3 * (size_t)n_imu);
  Imu imu;
  if (!select_imu(all, min_t, max_t, imu)) return PLV_OK;
  std::vector<double> st(n_whl + 2), s1(n_whl + 2), s2(n_whl + 2);
  int nw = 0, wok = 0;
  if (plv_select_wheel_data(n_whl, tw, m1, m2, min_t - toff, max_t - toff, n_whl + 2, st.data(), s1.data(), s2.data(), &nw, &wok) != PLV_OK || !wok)
    return PLV_OK;
  if (imu.size() < 20 || nw < 20) return PLV_OK;
  const double rl = op->intrinsics[0], rr = op->intrinsics[1], b = op->intrinsics[2];
  std::vector<Wheel> wh(nw);
  bool all_zero = true;
  for (int i = 0; i < nw; ++i) {
    double wz, vx;
    switch (op->wheel_type) {
      case PLV_WHEEL2D_ANG:
      case PLV_WHEEL3D_ANG: wz = (s2[i] * rr - s1[i] * rl) / b, vx = (s2[i] * rr + s1[i] * rl) / 2; break;
      case PLV_WHEEL2D_LIN:
      case PLV_WHEEL3D_LIN: wz = (s2[i] - s1[i]) / b, vx = (s2[i] + s1[i]) / 2; break;
      default: wz = s1[i], vx = s2[i];
    }
    wh[i] = {st[i], {0, 0, wz}, {vx, 0, 0}};
    if (wz != 0 || vx != 0) all_zero = false;  // wheel.second.norm() > 0   :52-58
  }
  M3 R_ItoO;
  std::copy(op->R_ItoO, op->R_ItoO + 9, R_ItoO.begin());
  const IW iw{*op, all, tr(R_ItoO), ld(op->p_IinO), ld(op->gravity), toff};
  const double gmag = nrm(iw.grav);
  if (mode) *mode = all_zero ? 0 : 1;

  // ---- init_bg_interpolate_imu (:169-197): bg = mean(w_imu(t_wheel) - R_OtoI w_O), IMU readings from the selected window
  V3 bg{0, 0, 0};
  int cnt = 0;
  for (const Wheel &w : wh) {
    const double tq = w.t + toff;
    if (tq > imu.t.back() || tq < imu.t.front()) continue;  // get_bounding_data (REF: Propagator.cpp:358-373)
    for (size_t i = 0; i + 1 < imu.size(); ++i)
      if (tq >= imu.t[i] && tq < imu.t[i + 1]) {
        const double lam = (tq - imu.t[i]) / (imu.t[i + 1] - imu.t[i]);
        const V3 wi = add(scl(1 - lam, ld(&imu.w[3 * i])), scl(lam, ld(&imu.w[3 * i + 3])));
        bg = add(bg, sub(wi, mv(iw.R_OtoI, w.w)));
        ++cnt;
        break;
      }
  }
  if (cnt < 1) return PLV_OK;
  bg = scl(1.0 / cnt, bg);
  // ---- init_vI_from_wheel (:199-204)
  const V3 v_I0 = mv(iw.R_OtoI, add(wh[0].v, cross(wh[0].w, iw.p_IinO)));
  // ---- gravity in {I0}
  V3 g_I0;
  if (op->imu_gravity_aligned) {
    g_I0 = iw.grav;  // :208-210, :266-269
  } else if (all_zero) {  // init_gI_simple :206-264
    V3 g{0, 0, 0};
    if (!iw.walk(bg, wh, [&](size_t, double sdt, V3 sRa, const M3 &, V3 v_It) { g = add(g, scl(1.0 / sdt, sub(add(v_I0, sRa), v_It))); }))
      return PLV_OK;
    g = scl(1.0 / (wh.size() - 1), g);
    g_I0 = scl(gmag / nrm(g), g);
  } else {  // init_gI_dongsi :266-432: min |A1 ba + A2 g - b| subject to |g| = gravity, A1 eliminated
    const size_t rows = 3 * wh.size();
    std::vector<double> A1(rows * 3, 0.0), A2(rows * 3, 0.0), bb(rows, 0.0);
    if (!iw.walk(bg, wh, [&](size_t i, double sdt, V3 sRa, const M3 &sR, V3 v_It) {
          const V3 r = sub(sub(v_It, v_I0), sRa);
          for (int a = 0; a < 3; ++a) {
            bb[3 * i + a] = r[a];
            for (int c = 0; c < 3; ++c) A1[(3 * i + a) * 3 + c] = -sR[3 * a + c];
            A2[(3 * i + a) * 3 + a] = -sdt;
          }
        }))
      return PLV_OK;
    auto gram = [&](const std::vector<double> &X, const std::vector<double> &Y) {  // X^T Y (3 x 3)
      M3 G{0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (size_t r = 0; r < rows; ++r)
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) G[3 * i + j] += X[3 * r + i] * Y[3 * r + j];
      return G;
    };
    auto gramv = [&](const std::vector<double> &X) {
      V3 g{0, 0, 0};
      for (size_t r = 0; r < rows; ++r)
        for (int i = 0; i < 3; ++i) g[i] += X[3 * r + i] * bb[r];
      return g;
    };
    const M3 A11 = gram(A1, A1), A12 = gram(A1, A2), A22 = gram(A2, A2);
    M3 A11inv;
    for (int c = 0; c < 3; ++c) {  // (A1^T A1).llt().solve(I)
      V3 e{0, 0, 0};
      e[c] = 1;
      const V3 x = llt_solve(A11, e);
      for (int r = 0; r < 3; ++r) A11inv[3 * r + c] = x[r];
    }
    // D = A2^T (I - A1 A11^-1 A1^T) A2,  d = A2^T (I - A1 A11^-1 A1^T) b
    const M3 Dm = madd(A22, mscl(-1.0, mm(tr(A12), mm(A11inv, A12))));
    const V3 dv = sub(gramv(A2), mv(tr(A12), mv(A11inv, gramv(A1))));
    const Poly poly = dongsi_poly(Dm, dv, gmag);
    bool found = false;
    double lam_min = -1, cost_min = INFINITY;
    for (double lam : real_roots(poly)) {
      M3 Ml = Dm;
      Ml[0] -= lam, Ml[4] -= lam, Ml[8] -= lam;
      const double cost = std::abs(nrm(llt_solve(Ml, dv)) - gmag);
      if (!found || cost < cost_min) found = true, lam_min = lam, cost_min = cost;
    }
    if (!found) return PLV_OK;
    M3 Ml = Dm;
    Ml[0] -= lam_min, Ml[4] -= lam_min, Ml[8] -= lam_min;
    g_I0 = llt_solve(Ml, dv);
    if (!(std::abs(nrm(g_I0) - gmag) <= 1e-3)) return PLV_OK;  // init_max_grav_difference :423-429
  }
  // ---- init_ba (:434-493)
  V3 ba{0, 0, 0};
  bool singular = false;
  if (!iw.walk(bg, wh, [&](size_t, double sdt, V3 sRa, const M3 &sR, V3 v_It) {
        M3 inv;
        if (!inv3(sR, inv)) {
          singular = true;
          return;
        }
        ba = add(ba, mv(inv, sub(sub(add(v_I0, sRa), scl(sdt, g_I0)), v_It)));
      }) ||
      singular)
    return PLV_OK;
  ba = scl(1.0 / (wh.size() - 1), ba);
  if (!all_zero && nrm(ba) > gmag) return PLV_OK;  // dynamic_initialization :586-587
  // ---- residual (:495-548): only its last block is looked at
  V3 res_tail{0, 0, 0};
  if (!iw.walk(bg, wh, [&](size_t, double sdt, V3 sRa, const M3 &sR, V3 v_It) {
        res_tail = add(add(sub(sub(v_It, v_I0), sRa), mv(sR, ba)), scl(sdt, g_I0));
      }))
    return PLV_OK;
  if (nrm(res_tail) / 3 > op->threshold * 100) {  // :567-571, :592-596
    state->cnt_smooth = 0;
    return PLV_OK;
  }
  double init[12];
  for (int i = 0; i < 3; ++i) init[i] = bg[i], init[3 + i] = ba[i], init[6 + i] = g_I0[i], init[9 + i] = v_I0[i];
  if (init12) std::copy(init, init + 12, init12);
  // ---- smoothness over consecutive calls (REF: IW_Initializer.cpp:70-104)
  if (state->cnt_smooth < 0) {
    state->cnt_smooth++;
    std::copy(init, init + 12, state->prev_init);
    return PLV_OK;
  }
  double diff = 0;
  for (int i = 0; i < 12; ++i) diff += (state->prev_init[i] - init[i]) * (state->prev_init[i] - init[i]);
  if (std::sqrt(diff) < op->threshold) state->cnt_smooth++;
  else state->cnt_smooth = 0;
  if (state->cnt_smooth > 3) {
    const M3 R_GtoI0 = gram_schmidt(g_I0, true);
    const Q4 q = rot_2_quat(R_GtoI0);
    const V3 v_G = mv(tr(R_GtoI0), v_I0);
    imustate[0] = wh[0].t + toff;
    for (int i = 0; i < 4; ++i) imustate[1 + i] = q[i];
    for (int i = 0; i < 3; ++i) imustate[5 + i] = 0, imustate[8 + i] = v_G[i], imustate[11 + i] = bg[i], imustate[14 + i] = ba[i];
    *ok = 1;
    return PLV_OK;
  }
  std::copy(init, init + 12, state->prev_init);
  return PLV_OK;
}

}  // extern "C"
