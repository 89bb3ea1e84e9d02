// wheel_api.hip — wheel odometry updater, 3D types.
//
//   UpdaterWheel::select_wheel_data / interpolate_data   REF: PL-VIWO/src/update/wheel/UpdaterWheel.cpp:142-215,784-794 (host)
//   UpdaterWheel::update                                 REF: UpdaterWheel.cpp:72-139
//   preintegration_3D / preintegration_intrinsics_3D     REF: UpdaterWheel.cpp:648-782, 472-500
//   compute_linear_system_3D                             REF: UpdaterWheel.cpp:327-424
//   Chi2Check + StateHelper::EKFUpdate with a full R     REF: UpdaterStatistics.cpp:94-117, StateHelper.cpp:94-173
//
// wheel_kernel: one workgroup.  The preintegration is a sequential recursion over the wheel samples between two clones
// (RK4 on a quaternion + a 6x6 covariance): lane 0 integrates the means, the 6x6 products Phi Cov Phi^T + Phi_n Q Phi_n^T
// are one element per lane.  The linear system is then laid down by lane 0, whitened by the Cholesky factor of the
// preintegrated covariance (so that the shared chi-square / EKF kernels, which take R = I, apply the full 6x6 noise), and
// handed to plv_slam_update's path: gate, then the covariance update on the resident P.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "plv_ctx.hpp"
#include "so3_dev.hpp"
#include "update_state.hpp"

namespace plv {
namespace {

using namespace so3;

#define TRY(expr)                  \
  do {                             \
    int _rc = (expr);              \
    if (_rc != PLV_OK) return _rc; \
  } while (0)

struct WheelArgs {
  plv_wheel_options op;
  plv_wheel_state st;
  int n_data, k;
  const double *t, *m1, *m2;  // device
  double *out;                // [H 6*k col-major][res 6][Cov 36][R 9][p 3][Hw 6*k][resw 6]
};

__device__ __forceinline__ void wheel_vel(const WheelArgs &A, double a1, double a2, D3 &w, D3 &v) {
  const double rl = A.st.intr[0], rr = A.st.intr[1], b = A.st.intr[2];
  if (A.op.type == PLV_WHEEL3D_ANG) {
    w = {0, 0, (a2 * rr - a1 * rl) / b};
    v = {(a2 * rr + a1 * rl) / 2, 0, 0};
  } else if (A.op.type == PLV_WHEEL3D_LIN) {
    w = {0, 0, (a2 - a1) / b};
    v = {(a2 + a1) / 2, 0, 0};
  } else {
    w = {0, 0, a1};
    v = {a2, 0, 0};
  }
}

__device__ __forceinline__ DM3 ldm(const double *p) {
  DM3 m;
#pragma unroll
  for (int i = 0; i < 9; ++i) m.m[i] = p[i];
  return m;
}
__device__ __forceinline__ void put3w(double *M, int ldm_, int r0, int c0, const DM3 &B) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) M[(r0 + r) * ldm_ + c0 + c] = B.m[3 * r + c];
}

__global__ void __launch_bounds__(64) wheel_kernel(WheelArgs A) {
  __shared__ double Cov[36], Ptr[36], Pns[36], Qd[6], X[36], Y[36];
  __shared__ double R3[9], p3[3], dRdi[9], dpdi[9];
  __shared__ double Hr[6 * 24], resv[6], L[36];
  const int tid = threadIdx.x, r = tid / 6, c = tid % 6;
  const bool el = tid < 36;
  if (el) Cov[tid] = 0.0;
  if (tid < 9) {
    R3[tid] = (tid % 4 == 0) ? 1.0 : 0.0;
    dRdi[tid] = 0.0;
    dpdi[tid] = 0.0;
  }
  if (tid < 3) p3[tid] = 0.0;
  __syncthreads();
  for (int i = 0; i < A.n_data - 1; ++i) {
    const double dt = A.t[i + 1] - A.t[i];
    if (tid == 0) {
      const DM3 R_3D = ldm(R3);
      const D3 p_3D = ld3(p3);
      if (A.op.do_calib_int) {  // preintegration_intrinsics_3D
        const double rl = A.st.intr[0], rr = A.st.intr[1], b = A.st.intr[2];
        const double w_l = A.m1[i], w_r = A.m2[i];
        const D3 w{0, 0, (w_r * rr - w_l * rl) / b}, v{(w_r * rr + w_l * rl) / 2, 0, 0};
        DM3 Hwx{{0, 0, 0, 0, 0, 0, -w_l / b, w_r / b, -(w_r * rr - w_l * rl) / (b * b)}};
        DM3 Hvx{{w_l / 2, w_r / 2, 0, 0, 0, 0, 0, 0, 0}};
        const DM3 R = exp3((-dt) * w);
        const DM3 Hth = mscale(dt, Jl((-dt) * w));
        const DM3 dR = ldm(dRdi), dp = ldm(dpdi);
        const DM3 ndp = madd(msub(dp, mmul(mmul(mtr(R_3D), skewm(dt * v)), dR)), mscale(dt, mmul(mtr(R_3D), Hvx)));
        const DM3 ndR = madd(mmul(R, dR), mmul(Hth, Hwx));
#pragma unroll
        for (int e = 0; e < 9; ++e) {
          dpdi[e] = ndp.m[e];
          dRdi[e] = ndR.m[e];
        }
      }
      // preintegration_3D: RK4 means
      D3 w1, v1, w2, v2;
      wheel_vel(A, A.m1[i], A.m2[i], w1, v1);
      wheel_vel(A, A.m1[i + 1], A.m2[i + 1], w2, v2);
      D3 w_hat = w1, v_hat = v1;
      const D3 w_alpha = (1.0 / dt) * (w2 - w1), v_jerk = (1.0 / dt) * (v2 - v1);
      const DQ q_local = R2q(R_3D);
      const DQ dq_0{0, 0, 0, 1};
      auto qdot = [&](DQ dq) {
        const DQ o = omega_times(w_hat, dq);
        return DQ{dt * (0.5 * o.x), dt * (0.5 * o.y), dt * (0.5 * o.z), dt * (0.5 * o.w)};
      };
      auto pdot = [&](DQ dq) { return dt * mvec(mtr(q2R(qmul(dq, q_local))), v_hat); };
      const DQ k1_q = qdot(dq_0);
      const D3 k1_p = pdot(dq_0);
      w_hat = w_hat + (0.5 * dt) * w_alpha;
      v_hat = v_hat + (0.5 * dt) * v_jerk;
      const DQ dq_1 = qnorm(qaxpy(dq_0, 0.5, k1_q));
      const DQ k2_q = qdot(dq_1);
      const D3 k2_p = pdot(dq_1);
      const DQ dq_2 = qnorm(qaxpy(dq_0, 0.5, k2_q));
      const DQ k3_q = qdot(dq_2);
      const D3 k3_p = pdot(dq_2);
      w_hat = w_hat + (0.5 * dt) * w_alpha;
      v_hat = v_hat + (0.5 * dt) * v_jerk;
      const DQ dq_3 = qnorm(qaxpy(dq_0, 1.0, k3_q));
      const DQ k4_q = qdot(dq_3);
      const D3 k4_p = pdot(dq_3);
      const DQ dq = qnorm(qaxpy(qaxpy(qaxpy(qaxpy(dq_0, 1.0 / 6.0, k1_q), 1.0 / 3.0, k2_q), 1.0 / 3.0, k3_q), 1.0 / 6.0, k4_q));
      const DM3 R_new = q2R(qmul(dq, q_local));
      const D3 new_p = (((p_3D + (1.0 / 6.0) * k1_p) + (1.0 / 3.0) * k2_p) + (1.0 / 3.0) * k3_p) + (1.0 / 6.0) * k4_p;
      // Phi_tr, Phi_ns, Q
      const double nw = A.op.noise_w * A.op.noise_w, nv = A.op.noise_v * A.op.noise_v, np = A.op.noise_p * A.op.noise_p, b = A.st.intr[2];
      double q0, q3;
      if (A.op.type == PLV_WHEEL3D_ANG) {
        q0 = nw / dt, q3 = nw / dt;
      } else if (A.op.type == PLV_WHEEL3D_LIN) {
        q0 = nv / b / b / dt, q3 = nv / 2 / 2 / dt;
      } else {
        q0 = nw / dt, q3 = nv / dt;
      }
      Qd[0] = q0, Qd[3] = q3, Qd[1] = Qd[2] = Qd[4] = Qd[5] = np / dt;
      for (int e = 0; e < 36; ++e) Ptr[e] = Pns[e] = 0.0;
      const DM3 RT = mtr(R_3D);
      put3w(Ptr, 6, 0, 0, mmul(R_new, RT));
      put3w(Ptr, 6, 3, 0, mmul(mscale(-1.0, RT), skewm(mvec(RT, new_p - p_3D))));
      put3w(Ptr, 6, 3, 3, eyem());
      put3w(Pns, 6, 0, 0, mscale(dt, eyem()));
      put3w(Pns, 6, 3, 3, mscale(dt, RT));
#pragma unroll
      for (int e = 0; e < 9; ++e) R3[e] = R_new.m[e];
      st3(p3, new_p);
    }
    __syncthreads();
    double a = 0.0, bq = 0.0;
    if (el) {  // X = Phi_tr Cov, Y = Phi_ns Q
#pragma unroll
      for (int k2 = 0; k2 < 6; ++k2) a += Ptr[r * 6 + k2] * Cov[k2 * 6 + c];
      bq = Pns[r * 6 + c] * Qd[c];
      X[tid] = a;
      Y[tid] = bq;
    }
    __syncthreads();
    double s = 0.0;
    if (el) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int k2 = 0; k2 < 6; ++k2) {
        s1 += X[r * 6 + k2] * Ptr[c * 6 + k2];
        s2 += Y[r * 6 + k2] * Pns[c * 6 + k2];
      }
      s = s1 + s2;
    }
    __syncthreads();
    if (el) X[tid] = s;
    __syncthreads();
    if (el) Cov[tid] = 0.5 * (X[tid] + X[c * 6 + r]);
    __syncthreads();
  }
  const int k = A.k;
  if (tid == 0) {  // compute_linear_system_3D
    const DM3 R_3D = ldm(R3);
    const D3 p_3D = ld3(p3);
    D3 pI0 = ld3(A.st.p0), pI1 = ld3(A.st.p1);
    DM3 RG0 = ldm(A.st.R0), RG1 = ldm(A.st.R1);
    const D3 pIinO = ld3(A.st.p_IinO);
    const DM3 RItoO = ldm(A.st.R_ItoO);
    const D3 pOinI = mvec(mscale(-1.0, mtr(RItoO)), pIinO);
    DM3 RO0toO1 = mmul(mmul(mmul(RItoO, RG1), mtr(RG0)), mtr(RItoO));
    const D3 r_ori = -1.0 * log3(mmul(R_3D, mtr(RO0toO1)));
    const D3 p_est = mvec(mmul(RItoO, RG0), ((pI1 + mvec(mtr(RG1), pOinI)) - pI0) - mvec(mtr(RG0), pOinI));
    const D3 r_pos = p_3D - p_est;
    st3(resv, r_ori);
    st3(resv + 3, r_pos);
    for (int e = 0; e < 6 * k; ++e) Hr[e] = 0.0;
    pI0 = ld3(A.st.p0_fej), pI1 = ld3(A.st.p1_fej), RG0 = ldm(A.st.R0_fej), RG1 = ldm(A.st.R1_fej);
    RO0toO1 = mmul(mmul(mmul(RItoO, RG1), mtr(RG0)), mtr(RItoO));
    const DM3 RO1toO0 = mtr(RO0toO1);
    const DM3 dzr_dth0 = mmul(mmul(mscale(-1.0, RItoO), RG1), mtr(RG0)), dzr_dth1 = RItoO;
    const DM3 dzp_dth0 = mmul(RItoO, skewm((mvec(RG0, pI1) + mvec(mmul(RG0, mtr(RG1)), pOinI)) - mvec(RG0, pI0)));
    const DM3 dzp_dp0 = mmul(mscale(-1.0, RItoO), RG0);
    const DM3 dzp_dth1 = mmul(mmul(mmul(mscale(-1.0, RItoO), RG0), mtr(RG1)), skewm(pOinI));
    const DM3 dzp_dp1 = mmul(RItoO, RG0);
    put3w(Hr, k, 0, 0, dzr_dth0), put3w(Hr, k, 0, 6, dzr_dth1);
    put3w(Hr, k, 3, 0, dzp_dth0), put3w(Hr, k, 3, 3, dzp_dp0), put3w(Hr, k, 3, 6, dzp_dth1), put3w(Hr, k, 3, 9, dzp_dp1);
    int hc = 12;
    if (A.op.do_calib_ext) {
      put3w(Hr, k, 0, hc, msub(eyem(), RO0toO1));
      put3w(Hr, k, 3, hc, madd(skewm(mvec(mmul(RItoO, RG0), pI1 - pI0) - mvec(RO1toO0, pIinO)), mmul(RO1toO0, skewm(pIinO))));
      put3w(Hr, k, 3, hc + 3, madd(mscale(-1.0, RO1toO0), eyem()));
      hc += 6;
    }
    if (A.op.do_calib_dt) {
      const D3 w0 = ld3(A.st.w0), v0 = ld3(A.st.v0), w1 = ld3(A.st.w1), v1 = ld3(A.st.v1);
      const D3 a = mvec(dzr_dth0, w0) + mvec(dzr_dth1, w1);
      const D3 cc = ((mvec(dzp_dth0, w0) + mvec(dzp_dp0, v0)) + mvec(dzp_dth1, w1)) + mvec(dzp_dp1, v1);
      Hr[0 * k + hc] = a.x, Hr[1 * k + hc] = a.y, Hr[2 * k + hc] = a.z;
      Hr[3 * k + hc] = cc.x, Hr[4 * k + hc] = cc.y, Hr[5 * k + hc] = cc.z;
      hc += 1;
    }
    if (A.op.do_calib_int) {
      put3w(Hr, k, 0, hc, mscale(-1.0, ldm(dRdi)));
      put3w(Hr, k, 3, hc, mscale(-1.0, ldm(dpdi)));
    }
    // Cholesky of the preintegrated covariance (6 x 6, lower), for the whitening below
    for (int e = 0; e < 36; ++e) L[e] = 0.0;
    for (int j = 0; j < 6; ++j) {
      double d = Cov[j * 6 + j];
      for (int q = 0; q < j; ++q) d -= L[j * 6 + q] * L[j * 6 + q];
      d = sqrt(d);
      L[j * 6 + j] = d;
      for (int i2 = j + 1; i2 < 6; ++i2) {
        double v = Cov[i2 * 6 + j];
        for (int q = 0; q < j; ++q) v -= L[i2 * 6 + q] * L[j * 6 + q];
        L[i2 * 6 + j] = v / d;
      }
    }
  }
  __syncthreads();
  double *oH = A.out, *ores = oH + 6 * k, *oC = ores + 6, *oR = oC + 36, *op = oR + 9, *oHw = op + 3, *oresw = oHw + 6 * k;
  for (int e = tid; e < 6 * k; e += 64) {
    const int cc = e / 6, rr = e % 6;
    oH[e] = Hr[rr * k + cc];  // col-major
  }
  if (tid < 6) ores[tid] = resv[tid];
  if (el) oC[tid] = Cov[tid];
  if (tid < 9) oR[tid] = R3[tid];
  if (tid < 3) op[tid] = p3[tid];
  // whitened system: forward substitution with L, one column per lane (column k = the residual)
  for (int col = tid; col <= k; col += 64) {
    double y[6];
#pragma unroll
    for (int i2 = 0; i2 < 6; ++i2) {
      double v = col < k ? Hr[i2 * k + col] : resv[i2];
      for (int q = 0; q < i2; ++q) v -= L[i2 * 6 + q] * y[q];
      y[i2] = v / L[i2 * 6 + i2];
    }
#pragma unroll
    for (int i2 = 0; i2 < 6; ++i2) {
      if (col < k) oHw[col * 6 + i2] = y[i2];
      else oresw[i2] = y[i2];
    }
  }
}

// Sensitivities of one constant-rate planar arc: heading `a`, turn rate `r` (the heading advances by -r dt) and forward speed `s`
// held over dt.  d(x, y) by the heading (_a), the rate (_r) and the speed (_s); below |r| = 1e-4 the straight-line limits.  The
// closed forms are the model's (REF: UpdaterWheel.cpp:449-461, 574-611, evaluated there twice: once per intrinsic chain and once
// for the covariance), written here once over the two differences every one of them contains; operation order as there.
struct ArcSens {
  double x_a, y_a, x_r, y_r, x_s, y_s;
};
__device__ inline ArcSens arc_sensitivities(double a, double r, double s, double dt) {
  ArcSens d;
  if (fabs(r) < 0.0001) {
    const double sa = sin(a), ca = cos(a);
    d.x_a = s * sa * dt, d.y_a = s * ca * dt;
    d.x_r = s * sa * dt * dt / 2, d.y_r = s * ca * dt * dt / 2;
    d.x_s = ca * dt, d.y_s = -sa * dt;
    return d;
  }
  const double a_end = a - r * dt;
  const double dcos = cos(a_end) - cos(a), dsin = sin(a_end) - sin(a);
  d.x_a = (s * dcos) / r;
  d.y_a = -(s * dsin) / r;
  d.x_r = (s * dsin) / r / r + (s * cos(a_end) * dt) / r;
  d.y_r = (s * dcos) / r / r - (s * sin(a_end) * dt) / r;
  d.x_s = -dsin / r;
  d.y_s = -dcos / r;
  return d;
}

// The 2D types (REF: preintegration_2D :502-646, preintegration_intrinsics_2D :426-470, compute_linear_system_2D :217-325):
// scalar recursions and a 3x3 covariance, all on lane 0; the lanes then whiten one column each.
// out: [H 3*k col-major][res 3][Cov 9][R 9 = I][meas 3 = theta x y][Hw 3*k][resw 3]
__global__ void __launch_bounds__(64) wheel2d_kernel(WheelArgs A) {
  __shared__ double Hr[3 * 24], resv[3], L[9], C2[9], meas[3];
  const int tid = threadIdx.x, k = A.k;
  if (tid == 0) {
    double head = 0, px = 0, py = 0;   // the preintegrated planar pose: heading, position
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double g_head[3] = {0, 0, 0}, g_px[3] = {0, 0, 0}, g_py[3] = {0, 0, 0};   // its gradients by (r_left, r_right, base)
    const double rl = A.st.intr[0], rr = A.st.intr[1], b = A.st.intr[2];
    for (int i = 0; i < A.n_data - 1; ++i) {
      const double dt = A.t[i + 1] - A.t[i];
      if (A.op.do_calib_int) {
        const double enc_l = A.m1[i], enc_r = A.m2[i];
        const double rate = (enc_r * rr - enc_l * rl) / b, speed = (enc_r * rr + enc_l * rl) / 2;
        const double rate_by[3] = {-enc_l / b, enc_r / b, -(enc_r * rr - enc_l * rl) / (b * b)}, speed_by[3] = {enc_l / 2, enc_r / 2, 0};
        const ArcSens d = arc_sensitivities(head, rate, speed, dt);
        for (int c = 0; c < 3; ++c) {
          g_px[c] = ((g_px[c] + d.x_a * g_head[c]) + d.x_r * rate_by[c]) + d.x_s * speed_by[c];
          g_py[c] = ((g_py[c] + d.y_a * g_head[c]) + d.y_r * rate_by[c]) + d.y_s * speed_by[c];
        }
        for (int c = 0; c < 3; ++c) g_head[c] = g_head[c] + dt * rate_by[c];
      }
      // rate and speed at both ends of the interval, from what the type measures
      double rate0, rate1, speed0, speed1;
      if (A.op.type == PLV_WHEEL2D_ANG) {
        rate0 = (A.m2[i] * rr - A.m1[i] * rl) / b, speed0 = (A.m2[i] * rr + A.m1[i] * rl) / 2;
        rate1 = (A.m2[i + 1] * rr - A.m1[i + 1] * rl) / b, speed1 = (A.m2[i + 1] * rr + A.m1[i + 1] * rl) / 2;
      } else if (A.op.type == PLV_WHEEL2D_LIN) {
        rate0 = (A.m2[i] - A.m1[i]) / b, speed0 = (A.m2[i] + A.m1[i]) / 2;
        rate1 = (A.m2[i + 1] - A.m1[i + 1]) / b, speed1 = (A.m2[i + 1] + A.m1[i + 1]) / 2;
      } else {
        rate0 = A.m1[i], speed0 = A.m2[i], rate1 = A.m1[i + 1], speed1 = A.m2[i + 1];
      }
      // RK4 over the interval with rate and speed interpolated linearly; the stages' headings are relative to the interval's start
      const double rate_slope = (rate1 - rate0) / dt, speed_slope = (speed1 - speed0) / dt;
      double rate = rate0, speed = speed0;
      const double s1_head = -rate * dt, s1_x = speed * 1 * dt;
      const double mid_a = 0.5 * s1_head;
      rate += 0.5 * rate_slope * dt;
      speed += 0.5 * speed_slope * dt;
      const double s2_head = -rate * dt, s2_x = speed * cos(mid_a) * dt;
      const double mid_b = 0.5 * s2_head;
      const double s3_head = -rate * dt, s3_x = speed * cos(mid_b) * dt;
      const double end_c = s3_head;
      rate += 0.5 * rate_slope * dt;
      speed += 0.5 * speed_slope * dt;
      const double s4_head = -rate * dt, s4_x = speed * cos(end_c) * dt;
      const double head_next = head + (1.0 / 6.0) * (s1_head + 2 * s2_head + 2 * s3_head + s4_head);
      const double px_next = px + (1.0 / 6.0) * (s1_x + 2 * s2_x + 2 * s3_x + s4_x);
      double py_next;   // (the lateral coordinate takes the arc's closed form on the interval's first sample, as the reference does)
      if (fabs(rate0) < 0.0001)
        py_next = py - speed0 * sin(head - rate0 * dt) * dt;
      else
        py_next = py - (speed0 * (cos(head - rate0 * dt) - cos(head))) / rate0;
      // how the measurement noise enters rate and speed
      double rate_n[2], speed_n[2];
      if (A.op.type == PLV_WHEEL2D_ANG) {
        rate_n[0] = rl / b, rate_n[1] = -rr / b, speed_n[0] = -rl / 2, speed_n[1] = -rr / 2;
      } else if (A.op.type == PLV_WHEEL2D_LIN) {
        rate_n[0] = 1.0 / b, rate_n[1] = -1.0 / b, speed_n[0] = -1.0 / 2, speed_n[1] = -1.0 / 2;
      } else {
        rate_n[0] = 1, rate_n[1] = 0, speed_n[0] = 0, speed_n[1] = 1;
      }
      const ArcSens d = arc_sensitivities(head, rate0, speed0, dt);
      const double Ptr[9] = {1, 0, 0, d.x_a, 1, 0, d.y_a, 0, 1};
      double Pns[6];
      for (int c = 0; c < 2; ++c) {
        Pns[c] = dt * rate_n[c];
        Pns[2 + c] = d.x_r * rate_n[c] + d.x_s * speed_n[c];
        Pns[4 + c] = d.y_r * rate_n[c] + d.y_s * speed_n[c];
      }
      double Q0, Q1;
      if (A.op.type == PLV_WHEEL2D_ANG)
        Q0 = Q1 = A.op.noise_w * A.op.noise_w / dt;
      else if (A.op.type == PLV_WHEEL2D_LIN)
        Q0 = Q1 = A.op.noise_v * A.op.noise_v / dt;
      else
        Q0 = A.op.noise_w * A.op.noise_w / dt, Q1 = A.op.noise_v * A.op.noise_v / dt;
      double X[9], N[9];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          double a = 0;
          for (int q = 0; q < 3; ++q) a += Ptr[3 * r + q] * C[3 * q + c];
          X[3 * r + c] = a;
        }
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          double a = 0;
          for (int q = 0; q < 3; ++q) a += X[3 * r + q] * Ptr[3 * c + q];
          const double nn = (Pns[2 * r] * Q0) * Pns[2 * c] + (Pns[2 * r + 1] * Q1) * Pns[2 * c + 1];
          N[3 * r + c] = a + nn;
        }
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[3 * r + c] = 0.5 * (N[3 * r + c] + N[3 * c + r]);
      head = head_next, px = px_next, py = py_next;
    }
    // compute_linear_system_2D
    D3 pI0 = ld3(A.st.p0), pI1 = ld3(A.st.p1);
    DM3 RG0 = ldm(A.st.R0), RG1 = ldm(A.st.R1);
    const D3 pIinO = ld3(A.st.p_IinO);
    const DM3 RItoO = ldm(A.st.R_ItoO);
    const D3 pOinI = mvec(mscale(-1.0, mtr(RItoO)), pIinO);
    const double theta_est = log3(mmul(mmul(mmul(RItoO, RG1), mtr(RG0)), mtr(RItoO))).z;
    const D3 d_est = mvec(mmul(RItoO, RG0), ((pI1 + mvec(mtr(RG1), pOinI)) - pI0) - mvec(mtr(RG0), pOinI));
    resv[0] = theta_est - head;
    resv[1] = px - d_est.x;
    resv[2] = py - d_est.y;
    for (int e = 0; e < 3 * k; ++e) Hr[e] = 0.0;
    pI0 = ld3(A.st.p0_fej), pI1 = ld3(A.st.p1_fej), RG0 = ldm(A.st.R0_fej), RG1 = ldm(A.st.R1_fej);
    const DM3 RO0toO1 = mmul(mmul(mmul(RItoO, RG1), mtr(RG0)), mtr(RItoO)), RO1toO0 = mtr(RO0toO1);
    const DM3 A0 = mmul(mmul(mscale(-1.0, RItoO), RG1), mtr(RG0));
    const DM3 Pth0 = mmul(RItoO, skewm(mvec(RG0, (pI1 + mvec(mtr(RG1), pOinI)) - pI0)));
    const DM3 Pp0 = mmul(mscale(-1.0, RItoO), RG0);
    const DM3 Pth1 = mmul(mmul(mmul(mscale(-1.0, RItoO), RG0), mtr(RG1)), skewm(pOinI));
    const DM3 Pp1 = mmul(RItoO, RG0);
    auto row_of = [&](int r, int c0, const DM3 &B, int br) {
      for (int c = 0; c < 3; ++c) Hr[r * k + c0 + c] = B.m[3 * br + c];
    };
    row_of(0, 0, A0, 2);
    row_of(0, 6, RItoO, 2);
    for (int r = 0; r < 2; ++r) {
      row_of(1 + r, 0, Pth0, r);
      row_of(1 + r, 3, Pp0, r);
      row_of(1 + r, 6, Pth1, r);
      row_of(1 + r, 9, Pp1, r);
    }
    int hc = 12;
    if (A.op.do_calib_ext) {
      row_of(0, hc, msub(eyem(), RO0toO1), 2);
      const DM3 Dth = madd(skewm(mvec(mmul(RItoO, RG0), pI1 - pI0) - mvec(RO1toO0, pIinO)), mmul(RO1toO0, skewm(pIinO)));
      const DM3 Dp = madd(mscale(-1.0, RO1toO0), eyem());
      for (int r = 0; r < 2; ++r) {
        row_of(1 + r, hc, Dth, r);
        row_of(1 + r, hc + 3, Dp, r);
      }
      hc += 6;
    }
    if (A.op.do_calib_dt) {
      const D3 w0 = ld3(A.st.w0), v0 = ld3(A.st.v0), w1 = ld3(A.st.w1), v1 = ld3(A.st.v1);
      const D3 a = mvec(A0, w0) + mvec(RItoO, w1);
      const D3 cc = ((mvec(Pth0, w0) + mvec(Pp0, v0)) + mvec(Pth1, w1)) + mvec(Pp1, v1);
      Hr[0 * k + hc] = a.z;
      Hr[1 * k + hc] = cc.x;
      Hr[2 * k + hc] = cc.y;
      hc += 1;
    }
    if (A.op.do_calib_int)
      for (int c = 0; c < 3; ++c) {
        Hr[0 * k + hc + c] = -g_head[c];
        Hr[1 * k + hc + c] = -g_px[c];
        Hr[2 * k + hc + c] = -g_py[c];
      }
    for (int e = 0; e < 9; ++e) {
      C2[e] = C[e];
      L[e] = 0.0;
    }
    for (int j = 0; j < 3; ++j) {
      double d = C[j * 3 + j];
      for (int q = 0; q < j; ++q) d -= L[j * 3 + q] * L[j * 3 + q];
      d = sqrt(d);
      L[j * 3 + j] = d;
      for (int i2 = j + 1; i2 < 3; ++i2) {
        double v = C[i2 * 3 + j];
        for (int q = 0; q < j; ++q) v -= L[i2 * 3 + q] * L[j * 3 + q];
        L[i2 * 3 + j] = v / d;
      }
    }
    meas[0] = head, meas[1] = px, meas[2] = py;
  }
  __syncthreads();
  double *oH = A.out, *ores = oH + 3 * k, *oC = ores + 3, *oR = oC + 9, *op = oR + 9, *oHw = op + 3, *oresw = oHw + 3 * k;
  for (int e = tid; e < 3 * k; e += 64) oH[e] = Hr[(e % 3) * k + e / 3];
  if (tid < 3) {
    ores[tid] = resv[tid];
    op[tid] = meas[tid];
  }
  if (tid < 9) {
    oC[tid] = C2[tid];
    oR[tid] = (tid % 4 == 0) ? 1.0 : 0.0;
  }
  for (int col = tid; col <= k; col += 64) {
    double y[3];
#pragma unroll
    for (int i2 = 0; i2 < 3; ++i2) {
      double v = col < k ? Hr[i2 * k + col] : resv[i2];
      for (int q = 0; q < i2; ++q) v -= L[i2 * 3 + q] * y[q];
      y[i2] = v / L[i2 * 3 + i2];
    }
#pragma unroll
    for (int i2 = 0; i2 < 3; ++i2) {
      if (col < k) oHw[col * 3 + i2] = y[i2];
      else oresw[i2] = y[i2];
    }
  }
}

int wheel_columns(const plv_wheel_options *op, const plv_wheel_state *st, int *cols) {
  int nc = 0;
  for (int i = 0; i < 6; ++i) cols[nc++] = st->pose0_id + i;
  for (int i = 0; i < 6; ++i) cols[nc++] = st->pose1_id + i;
  if (op->do_calib_ext)
    for (int i = 0; i < 6; ++i) cols[nc++] = st->ext_id + i;
  if (op->do_calib_dt) cols[nc++] = st->dt_id;
  if (op->do_calib_int)
    for (int i = 0; i < 3; ++i) cols[nc++] = st->intr_id + i;
  return nc;
}

// runs the kernel; host copies of every output block
int wheel_system(plv_ctx *ctx, const plv_wheel_options *op, const plv_wheel_state *st, int n_data, const double *t, const double *m1,
                 const double *m2, std::vector<double> &out, int &k, int &rows) {
  if (!ctx || !op || !st || n_data < 2 || !t || !m1 || !m2 || op->type < 0 || op->type > PLV_WHEEL2D_CEN) return PLV_E_BADARG;
  k = 12 + (op->do_calib_ext ? 6 : 0) + (op->do_calib_dt ? 1 : 0) + (op->do_calib_int ? 3 : 0);
  rows = op->type >= PLV_WHEEL2D_ANG ? 3 : 6;
  (void)hipSetDevice(ctx->device);
  auto *us = plv_update_state(ctx);
  const size_t nd = (size_t)n_data, n_out = (size_t)2 * rows * k + rows + (size_t)rows * rows + 9 + 3 + rows;
  TRY(us->eval.reserve((3 * nd + n_out) * 8));
  double *d = us->eval.as<double>();
  std::vector<double> h(3 * nd);
  std::copy(t, t + nd, h.begin());
  std::copy(m1, m1 + nd, h.begin() + nd);
  std::copy(m2, m2 + nd, h.begin() + 2 * nd);
  PLV_HIP_CHECK(plv::memcpy_async(d, h.data(), 3 * nd * 8, hipMemcpyHostToDevice, ctx->stream));
  WheelArgs A{};
  A.op = *op;
  A.st = *st;
  A.n_data = n_data;
  A.k = k;
  A.t = d, A.m1 = d + nd, A.m2 = d + 2 * nd;
  A.out = d + 3 * nd;
  {
    ProfScope ps(ctx->prof, "wheel_kernel", ctx->stream);
    if (rows == 6)
      hipLaunchKernelGGL(wheel_kernel, dim3(1), dim3(64), 0, ctx->stream, A);
    else
      hipLaunchKernelGGL(wheel2d_kernel, dim3(1), dim3(64), 0, ctx->stream, A);
  }
  PLV_HIP_CHECK(hipGetLastError());
  out.resize(n_out);
  PLV_HIP_CHECK(plv::memcpy_async(out.data(), A.out, n_out * 8, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  ctx->prof.collect();
  return PLV_OK;
}

}  // namespace
}  // namespace plv

using namespace plv;

extern "C" {

int plv_select_wheel_data(int n, const double *t, const double *m1, const double *m2, double time0, double time1, int cap, double *ot,
                          double *o1, double *o2, int *n_out, int *ok) {
  if (!n_out || !ok || n < 0 || (n > 0 && (!t || !m1 || !m2))) return PLV_E_BADARG;
  *n_out = 0;
  *ok = 0;
  if (n < 1) return PLV_OK;                                  // :144-147
  if (t[n - 1] <= time1 || t[0] > time0) return PLV_OK;      // :150-154
  std::vector<double> vt, v1, v2;
  auto push = [&](double a, double b, double c) {
    vt.push_back(a);
    v1.push_back(b);
    v2.push_back(c);
  };
  auto interp = [&](int a, int b, double ts) {  // interpolate_data :784-794
    const double lambda = (ts - t[a]) / (t[b] - t[a]);
    push(ts, (1 - lambda) * m1[a] + lambda * m1[b], (1 - lambda) * m2[a] + lambda * m2[b]);
  };
  for (int i = 0; i < n - 1; i++) {
    if (t[i + 1] > time0 && t[i] < time0) {  // :162-166 split at time0
      interp(i, i + 1, time0);
      continue;
    }
    if (t[i] >= time0 && t[i + 1] <= time1) {  // :170-173 whole interval
      push(t[i], m1[i], m2[i]);
      continue;
    }
    if (t[i + 1] > time1) {  // :179-197 the last one, cut at time1
      if (t[i] > time1)
        interp(i - 1, i, time1);
      else
        push(t[i], m1[i], m2[i]);
      if (vt.back() != time1) interp(i, i + 1, time1);
      break;
    }
  }
  if (vt.size() < 2) return PLV_OK;  // :200-203
  for (size_t i = 0; i + 1 < vt.size(); i++)  // :207-212 zero-length steps removed
    if (std::fabs(vt[i + 1] - vt[i]) < 1e-12) {
      vt.erase(vt.begin() + i);
      v1.erase(v1.begin() + i);
      v2.erase(v2.begin() + i);
      i--;
    }
  *n_out = (int)vt.size();
  if ((int)vt.size() > cap) return PLV_E_CAPACITY;
  if (!ot || !o1 || !o2) return PLV_E_BADARG;
  std::copy(vt.begin(), vt.end(), ot);
  std::copy(v1.begin(), v1.end(), o1);
  std::copy(v2.begin(), v2.end(), o2);
  *ok = 1;
  return PLV_OK;
}

int plv_wheel_linear_system(plv_ctx *ctx, const plv_wheel_options *opt, const plv_wheel_state *st, int n_data, const double *t,
                            const double *m1, const double *m2, double *H, double *res, double *Cov, int *col_to_state, int *k_out,
                            int *rows_out, double *R_3D, double *p_3D) {
  if (!H || !res || !Cov || !col_to_state || !k_out) return PLV_E_BADARG;
  std::vector<double> out;
  int k = 0, rows = 0;
  TRY(wheel_system(ctx, opt, st, n_data, t, m1, m2, out, k, rows));
  const size_t o_res = (size_t)rows * k, o_cov = o_res + rows, o_R = o_cov + (size_t)rows * rows, o_p = o_R + 9;
  std::copy(out.begin(), out.begin() + o_res, H);
  std::copy(out.begin() + o_res, out.begin() + o_cov, res);
  std::copy(out.begin() + o_cov, out.begin() + o_R, Cov);
  if (R_3D) std::copy(out.begin() + o_R, out.begin() + o_p, R_3D);
  if (p_3D) std::copy(out.begin() + o_p, out.begin() + o_p + 3, p_3D);
  *k_out = wheel_columns(opt, st, col_to_state);
  if (rows_out) *rows_out = rows;
  return PLV_OK;
}

int plv_wheel_update(plv_ctx *ctx, const plv_wheel_options *opt, const plv_wheel_state *st, int n_data, const double *t,
                     const double *m1, const double *m2, uint8_t *accepted, double *dx) {
  if (!accepted || !dx || !ctx || ctx->cov_n < 1) return PLV_E_BADARG;
  std::vector<double> out;
  int k = 0, rows = 0;
  TRY(wheel_system(ctx, opt, st, n_data, t, m1, m2, out, k, rows));
  int cols[24];
  wheel_columns(opt, st, cols);
  for (int i = 0; i < k; ++i)
    if (cols[i] < 0 || cols[i] >= ctx->cov_n) {
      set_last_error("plv_wheel_update: column %d maps to state %d outside the covariance (%d)", i, cols[i], ctx->cov_n);
      return PLV_E_BADARG;
    }
  const double *Hw = out.data() + (size_t)rows * k + rows + (size_t)rows * rows + 12, *resw = Hw + (size_t)rows * k;
  for (int i = 0; i < rows * k + rows; ++i)
    if (!std::isfinite(Hw[i])) {
      set_last_error("plv_wheel_update: the preintegrated covariance is not positive definite");
      return PLV_E_NUMERIC;
    }
  // Chi2Check(H, res, Cov) + EKFUpdate(H, res, Cov) == the same two steps on the whitened system with R = I
  return plv_slam_update(ctx, rows, k, rows, Hw, resw, cols, opt->chi2_mult, accepted, dx);
}

}  // extern "C"
