// propagate_api.hip — IMU propagation + window maintenance on the device-resident covariance.
//
//   Propagator::select_imu_readings / interpolate_data   REF: PL-VIWO/src/state/Propagator.cpp:93-152,320-331  (host)
//   Propagator::propagate                                REF: Propagator.cpp:30-91
//   Propagator::predict_and_compute / predict_mean_rk4   REF: Propagator.cpp:154-318
//   Propagator::reset_cpi                                REF: Propagator.cpp:333-357                            (host)
//   CpiV1::feed_IMU (means + RK4 measurement covariance) REF: open_vins/ov_core/src/cpi/CpiV1.cpp:32-315
//   StateHelper::EKFPropagation                          REF: PL-VIWO/src/state/StateHelper.cpp:20-92
//   StateHelper::clone (augment_clone)                   REF: StateHelper.cpp:175-201,305-355
//
// propagate_kernel: ONE workgroup walks the IMU intervals in order (the recursion is sequential).  Per interval
// wave 0 / lane 0 integrates the mean (RK4) and lays F and G down in LDS while wave 1 / lane 0 advances the CPI means;
// every 15x15 product after that (Phi = F Phi, Qd = F Qd F^T + G Qc G^T, the four RK4 stages of the CPI
// measurement covariance) is one output element per thread.  Then two small launches apply Phi / Qd to the n x n
// covariance: P[:, imu] Phi^T into a staging strip, and the write-back of the row / column strips + the 15x15 block.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "plv_ctx.hpp"
#include "so3_dev.hpp"
#include "update_state.hpp"

namespace plv {
namespace {

#define TRY(expr)                  \
  do {                             \
    int _rc = (expr);              \
    if (_rc != PLV_OK) return _rc; \
  } while (0)

using namespace so3;

// LDS image of plv_imu_state: q4 p3 v3 bg3 ba3 qf4 pf3 vf3
enum { IQ = 0, IP = 4, IV = 7, IBG = 10, IBA = 13, IQF = 16, IPF = 20, IVF = 23, IMU_N = 26 };

struct PropArgs {
  int n_data;
  const double *t, *wm, *am;  // device
  double *imu;                // [26] in/out
  double sw, swb, sa, sab, g[3];
  double *cpi;                // plv_cpi_accum image in/out, or null
  double *records;            // [n_data-1][sizeof(plv_cpi_record)/8] or null
  double *Phi, *Qd;           // [225] out, row-major
  int mode;                   // 0 = Propagator::propagate, 1 = State::create_new_cpi_integrate (CPI only, one record at the end)
  double R0[9], t_given;      // mode 1: clones.at(clone_t)->Rot(), the requested time
};

__device__ __forceinline__ void put3(double *M, int ldm, int r0, int c0, const DM3 &B) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) M[(r0 + r) * ldm + c0 + c] = B.m[3 * r + c];
}

// predict_mean_rk4 + the blocks of predict_and_compute; one thread
__device__ void mean_and_jacobians(double *imu, const PropArgs &A, int i, double *F, double *G) {
  const double dt = A.t[i + 1] - A.t[i];
  const D3 g{A.g[0], A.g[1], A.g[2]};
  const D3 bg = ld3(imu + IBG), ba = ld3(imu + IBA);
  const D3 w1 = ld3(A.wm + 3 * i) - bg, a1 = ld3(A.am + 3 * i) - ba, w2 = ld3(A.wm + 3 * (i + 1)) - bg, a2 = ld3(A.am + 3 * (i + 1)) - ba;
  D3 w_hat = w1, a_hat = a1;
  const D3 w_alpha = (1.0 / dt) * (w2 - w1), a_jerk = (1.0 / dt) * (a2 - a1);
  const DQ q_0{imu[IQ], imu[IQ + 1], imu[IQ + 2], imu[IQ + 3]};
  const D3 p_0 = ld3(imu + IP), v_0 = ld3(imu + IV);
  const DQ dq_0{0, 0, 0, 1};
  auto vdot = [&](DQ dq) { return mvec(mtr(q2R(qmul(dq, q_0))), a_hat) - g; };
  auto qdot = [&](DQ dq) {
    const DQ o = omega_times(w_hat, dq);
    return DQ{0.5 * o.x, 0.5 * o.y, 0.5 * o.z, 0.5 * o.w};
  };
  const DQ q0d = qdot(dq_0);
  const D3 v0d = vdot(dq_0);
  const DQ k1_q{dt * q0d.x, dt * q0d.y, dt * q0d.z, dt * q0d.w};
  const D3 k1_p = dt * v_0, k1_v = dt * v0d;
  w_hat = w_hat + (0.5 * dt) * w_alpha;
  a_hat = a_hat + (0.5 * dt) * a_jerk;
  const DQ dq_1 = qnorm(qaxpy(dq_0, 0.5, k1_q));
  const D3 v_1 = v_0 + 0.5 * k1_v;
  const DQ q1d = qdot(dq_1);
  const D3 v1d = vdot(dq_1);
  const DQ k2_q{dt * q1d.x, dt * q1d.y, dt * q1d.z, dt * q1d.w};
  const D3 k2_p = dt * v_1, k2_v = dt * v1d;
  const DQ dq_2 = qnorm(qaxpy(dq_0, 0.5, k2_q));
  const D3 v_2 = v_0 + 0.5 * k2_v;
  const DQ q2d = qdot(dq_2);
  const D3 v2d = vdot(dq_2);
  const DQ k3_q{dt * q2d.x, dt * q2d.y, dt * q2d.z, dt * q2d.w};
  const D3 k3_p = dt * v_2, k3_v = dt * v2d;
  w_hat = w_hat + (0.5 * dt) * w_alpha;
  a_hat = a_hat + (0.5 * dt) * a_jerk;
  const DQ dq_3 = qnorm(qaxpy(dq_0, 1.0, k3_q));
  const D3 v_3 = v_0 + k3_v;
  const DQ q3d = qdot(dq_3);
  const D3 v3d = vdot(dq_3);
  const DQ k4_q{dt * q3d.x, dt * q3d.y, dt * q3d.z, dt * q3d.w};
  const D3 k4_p = dt * v_3, k4_v = dt * v3d;
  const DQ dq = qnorm(qaxpy(qaxpy(qaxpy(qaxpy(dq_0, 1.0 / 6.0, k1_q), 1.0 / 3.0, k2_q), 1.0 / 3.0, k3_q), 1.0 / 6.0, k4_q));
  const DQ new_q = qmul(dq, q_0);
  const D3 new_p = (((p_0 + (1.0 / 6.0) * k1_p) + (1.0 / 3.0) * k2_p) + (1.0 / 3.0) * k3_p) + (1.0 / 6.0) * k4_p;
  const D3 new_v = (((v_0 + (1.0 / 6.0) * k1_v) + (1.0 / 3.0) * k2_v) + (1.0 / 3.0) * k3_v) + (1.0 / 6.0) * k4_v;
  // ---- Jacobians with the first estimates (Propagator.cpp:189-220); local ids theta 0, p 3, v 6, bg 9, ba 12
  for (int e = 0; e < 225; ++e) F[e] = 0.0;
  for (int e = 0; e < 180; ++e) G[e] = 0.0;
  const DM3 Rfej = q2R(DQ{imu[IQF], imu[IQF + 1], imu[IQF + 2], imu[IQF + 3]}), RfT = mtr(Rfej);
  const DM3 dR = mmul(q2R(new_q), RfT);
  const D3 v_fej = ld3(imu + IVF), p_fej = ld3(imu + IPF);
  const DM3 thbg = mscale(dt, mmul(mscale(-1.0, dR), Jl(dt * w1)));  // -dR * Jr_so3(-w_hat dt) * dt, Jr(x) = Jl(-x)
  put3(F, 15, 0, 0, dR);
  put3(F, 15, 0, 9, thbg);
  put3(F, 15, 9, 9, eyem());
  put3(F, 15, 6, 0, mmul(mscale(-1.0, skewm((new_v - v_fej) + dt * g)), RfT));
  put3(F, 15, 6, 6, eyem());
  put3(F, 15, 6, 12, mscale(-dt, RfT));
  put3(F, 15, 12, 12, eyem());
  put3(F, 15, 3, 0, mmul(mscale(-1.0, skewm(((new_p - p_fej) - dt * v_fej) + (0.5 * dt * dt) * g)), RfT));
  put3(F, 15, 3, 6, mscale(dt, eyem()));
  put3(F, 15, 3, 12, mscale(-0.5 * dt * dt, RfT));
  put3(F, 15, 3, 3, eyem());
  put3(G, 12, 0, 0, thbg);
  put3(G, 12, 6, 3, mscale(-dt, RfT));
  put3(G, 12, 3, 3, mscale(-0.5 * dt * dt, RfT));
  put3(G, 12, 9, 6, eyem());
  put3(G, 12, 12, 9, eyem());
  // value and fej := propagated mean (:229-237)
  imu[IQ] = imu[IQF] = new_q.x, imu[IQ + 1] = imu[IQF + 1] = new_q.y, imu[IQ + 2] = imu[IQF + 2] = new_q.z, imu[IQ + 3] = imu[IQF + 3] = new_q.w;
  st3(imu + IP, new_p), st3(imu + IPF, new_p), st3(imu + IV, new_v), st3(imu + IVF, new_v);
}

// plv_cpi_accum image (doubles): clone_t 0, DT 1, R 2..10, alpha 11, beta 14, bw 17, ba 20, v_clone 23, P_meas 26..250
enum { CA_CLONE = 0, CA_DT = 1, CA_R = 2, CA_AL = 11, CA_BE = 14, CA_BW = 17, CA_BA = 20, CA_V = 23, CA_P = 26, CA_N = 251 };
// plv_cpi_record image: t 0, dt 1, clone_t 2, R 3..11, alpha 12, v 15, w 18, Q 21..56
enum { CR_T = 0, CR_DT = 1, CR_CL = 2, CR_R = 3, CR_AL = 12, CR_V = 15, CR_W = 18, CR_Q = 21, CR_N = 57 };

// CpiV1::feed_IMU means (:36-123); leaves w_x, a_x, R_k2tau (old), R_mid, R_k2tau1 for the covariance stages.  One thread.
__device__ void cpi_means(double *cpi, const PropArgs &A, int i, double *wx, double *ax, double *Rold, double *Rmid, double *Rnew) {
  const double delta_t = A.t[i + 1] - A.t[i];
  cpi[CA_DT] += delta_t;
  const D3 bw = ld3(cpi + CA_BW), bal = ld3(cpi + CA_BA);
  D3 w_hat = ld3(A.wm + 3 * i) - bw, a_hat = ld3(A.am + 3 * i) - bal;
  w_hat = w_hat + (ld3(A.wm + 3 * (i + 1)) - bw);
  w_hat = 0.5 * w_hat;
  a_hat = a_hat + (ld3(A.am + 3 * (i + 1)) - bal);
  a_hat = 0.5 * a_hat;
  const double mag_w = nrm3(w_hat), w_dt = mag_w * delta_t;
  const bool small_w = mag_w < 0.008726646;
  const double dt_2 = delta_t * delta_t, cos_wt = cos(w_dt), sin_wt = sin(w_dt);
  const DM3 w_x = skewm(w_hat), a_x = skewm(a_hat), w_x_2 = mmul(w_x, w_x), I = eyem();
  DM3 R_k2tau;
#pragma unroll
  for (int e = 0; e < 9; ++e) R_k2tau.m[e] = cpi[CA_R + e];
  const DM3 R_t2t1 = small_w ? madd(msub(I, mscale(delta_t, w_x)), mscale(dt_2 / 2, w_x_2))
                             : madd(msub(I, mscale(sin_wt / mag_w, w_x)), mscale((1.0 - cos_wt) / (mag_w * mag_w), w_x_2));
  const DM3 R_k2tau1 = mmul(R_t2t1, R_k2tau), R_tau12k = mtr(R_k2tau1);
  double f_1, f_2, f_3, f_4;
  if (small_w) {
    f_1 = -(pow(delta_t, 3) / 3);
    f_2 = pow(delta_t, 4) / 8;
    f_3 = -(dt_2 / 2);
    f_4 = pow(delta_t, 3) / 6;
  } else {
    f_1 = (w_dt * cos_wt - sin_wt) / pow(mag_w, 3);
    f_2 = (w_dt * w_dt - 2 * cos_wt - 2 * w_dt * sin_wt + 2) / (2 * pow(mag_w, 4));
    f_3 = -(1 - cos_wt) / (mag_w * mag_w);
    f_4 = (w_dt - sin_wt) / pow(mag_w, 3);
  }
  const DM3 alpha_arg = madd(madd(mscale(dt_2 / 2.0, I), mscale(f_1, w_x)), mscale(f_2, w_x_2));
  const DM3 Beta_arg = madd(madd(mscale(delta_t, I), mscale(f_3, w_x)), mscale(f_4, w_x_2));
  const DM3 H_al = mmul(R_tau12k, alpha_arg), H_be = mmul(R_tau12k, Beta_arg);
  D3 alpha = ld3(cpi + CA_AL), beta = ld3(cpi + CA_BE);
  alpha = alpha + (delta_t * beta + mvec(H_al, a_hat));
  beta = beta + mvec(H_be, a_hat);
  st3(cpi + CA_AL, alpha);
  st3(cpi + CA_BE, beta);
  const double hw = mag_w * .5 * delta_t;
  DM3 R_mid = small_w ? madd(msub(I, mscale(.5 * delta_t, w_x)), mscale(pow(.5 * delta_t, 2) / 2, w_x_2))
                      : madd(msub(I, mscale(sin(hw) / mag_w, w_x)), mscale((1.0 - cos(hw)) / (mag_w * mag_w), w_x_2));
  R_mid = mmul(R_mid, R_k2tau);
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    wx[e] = w_x.m[e], ax[e] = a_x.m[e], Rold[e] = R_k2tau.m[e], Rmid[e] = R_mid.m[e], Rnew[e] = R_k2tau1.m[e];
  }
}

// element (r, c) of CpiV1's continuous-time F for rotation R (:214-219): blocks (0,0) -w_x, (0,3) -I, (6,0) -R^T a_x, (6,9) -R^T, (12,6) I
__device__ __forceinline__ double cpi_F(int r, int c, const double *wx, const double *RTa, const double *R) {
  const int br = r / 3, bc = c / 3, i = r % 3, j = c % 3;
  if (br == 0 && bc == 0) return -wx[3 * i + j];
  if (br == 0 && bc == 1) return i == j ? -1.0 : 0.0;
  if (br == 2 && bc == 0) return -RTa[3 * i + j];
  if (br == 2 && bc == 3) return -R[3 * j + i];
  if (br == 4 && bc == 2) return i == j ? 1.0 : 0.0;
  return 0.0;
}
// element (r, c) of G Q_c G^T (:222-229): diag blocks sw^2 I, swb^2 I, sa^2 R^T R, sab^2 I, 0
__device__ __forceinline__ double cpi_GQG(int r, int c, const double *R, const PropArgs &A) {
  const int br = r / 3, bc = c / 3, i = r % 3, j = c % 3;
  if (br != bc) return 0.0;
  if (br == 0) return i == j ? A.sw * A.sw : 0.0;
  if (br == 1) return i == j ? A.swb * A.swb : 0.0;
  if (br == 2) return (-R[i]) * (A.sa * A.sa) * (-R[j]) + (-R[3 + i]) * (A.sa * A.sa) * (-R[3 + j]) + (-R[6 + i]) * (A.sa * A.sa) * (-R[6 + j]);
  if (br == 3) return i == j ? A.sab * A.sab : 0.0;
  return 0.0;
}

__global__ void __launch_bounds__(256) propagate_kernel(PropArgs A) {
  __shared__ double F[225], G[180], Phi[225], Qd[225], X[225], Y[225];
  __shared__ double Pm[225], Pk[225], Pd[4][225], RTa[3][9];
  __shared__ double imu[IMU_N], cpi[CA_N], wx[9], ax[9], Rs[3][9], RGtoIk[9];
  const int tid = threadIdx.x;
  const int r = tid / 15, c = tid % 15;
  const bool el = tid < 225;
  if (tid < IMU_N) imu[tid] = A.imu[tid];
  if (A.cpi)
    for (int e = tid; e < CA_N; e += 256) cpi[e] = A.cpi[e];
  if (el) {
    Phi[tid] = r == c ? 1.0 : 0.0;
    Qd[tid] = 0.0;
  }
  __syncthreads();
  if (tid == 0) {  // R_GtoIk = state->imu->Rot() at entry (:46) / clones.at(clone_t)->Rot() (State.cpp:389)
    const DM3 R0 = q2R(DQ{imu[IQ], imu[IQ + 1], imu[IQ + 2], imu[IQ + 3]});
    for (int e = 0; e < 9; ++e) RGtoIk[e] = A.mode == 1 ? A.R0[e] : R0.m[e];
  }
  for (int i = 0; i < A.n_data - 1; ++i) {
    __syncthreads();
    if (A.mode == 1) {
      if (tid == 0) {  // State.cpp:393 — the orientation advances BEFORE the feed, with the previous R_k2tau
        double nr[9];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) nr[3 * a + b] = cpi[CA_R + 3 * a] * RGtoIk[b] + cpi[CA_R + 3 * a + 1] * RGtoIk[3 + b] + cpi[CA_R + 3 * a + 2] * RGtoIk[6 + b];
        for (int e = 0; e < 9; ++e) RGtoIk[e] = nr[e];
      }
    } else if (tid == 0) {
      mean_and_jacobians(imu, A, i, F, G);
    }
    const double delta_t = A.t[i + 1] - A.t[i];
    const bool do_cpi = A.cpi != nullptr;
    if (do_cpi && tid == 64) {
      if (delta_t == 0) cpi[CA_DT] += 0.0;  // feed_IMU returns after DT += 0 (:37-43)
      else cpi_means(cpi, A, i, wx, ax, Rs[0], Rs[1], Rs[2]);
    }
    if (do_cpi && el) Pm[tid] = cpi[CA_P + tid];
    __syncthreads();
    // ---- Qdi = sym(G Qc G^T), Phi = F Phi, Qd = sym(F Qd F^T + Qdi)   (:56-58, :222-224)
    double q = 0.0, ph = 0.0, fq = 0.0;
    if (el && A.mode == 0) {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const double qc = k < 3 ? A.sw * A.sw / delta_t : (k < 6 ? A.sa * A.sa / delta_t : (k < 9 ? A.swb * A.swb * delta_t : A.sab * A.sab * delta_t));
        q += (G[r * 12 + k] * qc) * G[c * 12 + k];
      }
#pragma unroll
      for (int k = 0; k < 15; ++k) {
        ph += F[r * 15 + k] * Phi[k * 15 + c];
        fq += F[r * 15 + k] * Qd[k * 15 + c];
      }
      X[tid] = q;   // G Qc G^T
      Y[tid] = fq;  // F Qd
    }
    __syncthreads();
    double s = 0.0, qdi = 0.0;
    if (el && A.mode == 0) {
      Phi[tid] = ph;
#pragma unroll
      for (int k = 0; k < 15; ++k) s += Y[r * 15 + k] * F[c * 15 + k];
      qdi = 0.5 * (X[tid] + X[c * 15 + r]);
    }
    __syncthreads();
    if (el && A.mode == 0) X[tid] = s + qdi;
    __syncthreads();
    if (el && A.mode == 0) Qd[tid] = 0.5 * (X[tid] + X[c * 15 + r]);
    // ---- CPI measurement covariance, RK4 (:197-306), and the State::CPI record (Propagator.cpp:62-82)
    if (do_cpi && delta_t != 0) {
      __syncthreads();
      if (tid < 27) {  // -R^T a_x is formed as (R^T a_x) per stage rotation
        const int s = tid / 9, e = tid % 9, i3 = e / 3, j3 = e % 3;
        const double *R = Rs[s];
        RTa[s][e] = R[i3] * ax[j3] + R[3 + i3] * ax[3 + j3] + R[6 + i3] * ax[6 + j3];
      }
      __syncthreads();
      for (int stage = 0; stage < 4; ++stage) {
        const int rs = stage == 0 ? 0 : (stage == 3 ? 2 : 1);
        if (el) {
          double pk = Pm[tid];
          if (stage == 1) pk += Pd[0][tid] * delta_t / 2.0;
          if (stage == 2) pk += Pd[1][tid] * delta_t / 2.0;
          if (stage == 3) pk += Pd[2][tid] * delta_t;
          Pk[tid] = pk;
        }
        __syncthreads();
        if (el) {
          double s1 = 0.0, s2 = 0.0;
#pragma unroll
          for (int k = 0; k < 15; ++k) {
            s1 += cpi_F(r, k, wx, RTa[rs], Rs[rs]) * Pk[k * 15 + c];
            s2 += Pk[r * 15 + k] * cpi_F(c, k, wx, RTa[rs], Rs[rs]);
          }
          Pd[stage][tid] = (s1 + s2) + cpi_GQG(r, c, Rs[rs], A);
        }
        __syncthreads();
      }
      if (el) X[tid] = Pm[tid] + (delta_t / 6.0) * (((Pd[0][tid] + 2.0 * Pd[1][tid]) + 2.0 * Pd[2][tid]) + Pd[3][tid]);
      __syncthreads();
      if (el) cpi[CA_P + tid] = 0.5 * (X[tid] + X[c * 15 + r]);
      if (tid < 9) cpi[CA_R + tid] = Rs[2][tid];
      __syncthreads();
    }
    if (do_cpi && A.mode == 0) {
      __syncthreads();
      if (A.records && tid == 0) {
        double *rec = A.records + (size_t)i * CR_N;
        rec[CR_T] = A.t[i + 1];
        rec[CR_DT] = cpi[CA_DT];
        rec[CR_CL] = cpi[CA_CLONE];
        for (int e = 0; e < 9; ++e) rec[CR_R + e] = cpi[CA_R + e];
        for (int e = 0; e < 3; ++e) {
          rec[CR_AL + e] = cpi[CA_AL + e];
          rec[CR_W + e] = A.wm[3 * (i + 1) + e] - imu[IBG + e];
          // v = v_clone - g DT + R_GtoIk^T beta   (:73)
          const double rb = RGtoIk[e] * cpi[CA_BE] + RGtoIk[3 + e] * cpi[CA_BE + 1] + RGtoIk[6 + e] * cpi[CA_BE + 2];
          rec[CR_V + e] = (cpi[CA_V + e] - A.g[e] * cpi[CA_DT]) + rb;
        }
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) {
            rec[CR_Q + 6 * a + b] = cpi[CA_P + 15 * a + b];
            rec[CR_Q + 6 * a + 3 + b] = cpi[CA_P + 15 * a + 12 + b];
            rec[CR_Q + 6 * (3 + a) + b] = cpi[CA_P + 15 * (12 + a) + b];
            rec[CR_Q + 6 * (3 + a) + 3 + b] = cpi[CA_P + 15 * (12 + a) + 12 + b];
          }
      }
      __syncthreads();
      if (tid == 0) {  // R_GtoIk = R_k2tau * R_GtoIk (:80)
        double nr[9];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) nr[3 * a + b] = cpi[CA_R + 3 * a] * RGtoIk[b] + cpi[CA_R + 3 * a + 1] * RGtoIk[3 + b] + cpi[CA_R + 3 * a + 2] * RGtoIk[6 + b];
        for (int e = 0; e < 9; ++e) RGtoIk[e] = nr[e];
      }
    }
  }
  __syncthreads();
  if (A.mode == 1 && tid == 0) {  // State.cpp:398-411
    double *rec = A.records;
    const int last = A.n_data - 1;
    rec[CR_T] = A.t_given;
    rec[CR_DT] = A.t_given - cpi[CA_CLONE];
    rec[CR_CL] = cpi[CA_CLONE];
    for (int e = 0; e < 9; ++e) rec[CR_R + e] = cpi[CA_R + e];
    for (int e = 0; e < 3; ++e) {
      rec[CR_AL + e] = cpi[CA_AL + e];
      rec[CR_W + e] = A.wm[3 * last + e] - cpi[CA_BW + e];
      const double rb = RGtoIk[e] * cpi[CA_BE] + RGtoIk[3 + e] * cpi[CA_BE + 1] + RGtoIk[6 + e] * cpi[CA_BE + 2];
      rec[CR_V + e] = (cpi[CA_V + e] - A.g[e] * cpi[CA_DT]) + rb;
    }
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        rec[CR_Q + 6 * a + b] = cpi[CA_P + 15 * a + b];
        rec[CR_Q + 6 * a + 3 + b] = cpi[CA_P + 15 * a + 12 + b];
        rec[CR_Q + 6 * (3 + a) + b] = cpi[CA_P + 15 * (12 + a) + b];
        rec[CR_Q + 6 * (3 + a) + 3 + b] = cpi[CA_P + 15 * (12 + a) + 12 + b];
      }
  }
  if (tid < IMU_N) A.imu[tid] = imu[tid];
  if (A.cpi)
    for (int e = tid; e < CA_N; e += 256) A.cpi[e] = cpi[e];
  if (el) {
    A.Phi[tid] = Phi[tid];
    A.Qd[tid] = Qd[tid];
  }
}

// Cov_PhiT = P[:, imu] Phi^T  (n x 15, row-major strip)   REF: StateHelper.cpp:58-63
__global__ void __launch_bounds__(256) ekf_prop_strip_kernel(const double *__restrict__ P, int n, int imu_id, const double *__restrict__ Phi,
                                                             double *__restrict__ strip) {
  __shared__ double ph[225];
  if (threadIdx.x < 225) ph[threadIdx.x] = Phi[threadIdx.x];
  __syncthreads();
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * 15) return;
  const int c = idx / n, r = idx - c * n;  // consecutive threads walk a column of P: coalesced
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 15; ++k) s += P[(size_t)(imu_id + k) * n + r] * ph[c * 15 + k];
  strip[(size_t)r * 15 + c] = s;
}
// write-back: row strip, column strip, Phi Cov_PhiT[imu rows] + Q (Q read through its upper triangle)   REF: :66-78
__global__ void __launch_bounds__(256) ekf_prop_write_kernel(double *__restrict__ P, int n, int imu_id, const double *__restrict__ Phi,
                                                             const double *__restrict__ Qd, const double *__restrict__ strip) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * 15) return;
  const int c = idx / n, r = idx - c * n;
  if (r >= imu_id && r < imu_id + 15) {
    const int rr = r - imu_id;
    double s = rr <= c ? Qd[rr * 15 + c] : Qd[c * 15 + rr];
#pragma unroll
    for (int k = 0; k < 15; ++k) s += Phi[rr * 15 + k] * strip[(size_t)(imu_id + k) * 15 + c];
    P[(size_t)(imu_id + c) * n + r] = s;
    return;
  }
  const double v = strip[(size_t)r * 15 + c];
  P[(size_t)(imu_id + c) * n + r] = v;  // column block
  P[(size_t)r * n + imu_id + c] = v;    // row block
}

// StateHelper::clone: P (n x n) -> P2 ((n+size) x (n+size))
__global__ void __launch_bounds__(256) cov_clone_kernel(const double *__restrict__ P, int n, int src, int size, double *__restrict__ P2) {
  const int m = n + size;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < m * m; idx += gridDim.x * 256) {
    const int c = idx / m, r = idx - c * m;
    const int sr = r < n ? r : src + (r - n), sc = c < n ? c : src + (c - n);
    P2[idx] = P[(size_t)sc * n + sr];
  }
}

}  // namespace
}  // namespace plv

using namespace plv;

extern "C" {

int plv_select_imu_readings(int n, const double *t, const double *wm, const double *am, double time0, double time1, int cap,
                            double *ot, double *owm, double *oam, int *n_out, int *ok) {
  if (!n_out || !ok || n < 0 || (n > 0 && (!t || !wm || !am)) || cap < 0 || (cap > 0 && (!ot || !owm || !oam))) return PLV_E_BADARG;
  *n_out = 0;
  *ok = 0;
  if (n < 2 || time1 <= time0 || t[0] > time0 || t[n - 1] < time1) return PLV_OK;  // :96-123
  int m = 0;
  auto push = [&](double tt, const double *w, const double *a) {
    if (m < cap) {
      ot[m] = tt;
      std::copy(w, w + 3, owm + 3 * (size_t)m);
      std::copy(a, a + 3, oam + 3 * (size_t)m);
    }
    ++m;
  };
  auto interp = [&](size_t i, double ts) {  // interpolate_data :320-331
    const double lambda = (ts - t[i]) / (t[i + 1] - t[i]);
    double w[3], a[3];
    for (int c = 0; c < 3; ++c) {
      a[c] = (1 - lambda) * am[3 * i + c] + lambda * am[3 * (i + 1) + c];
      w[c] = (1 - lambda) * wm[3 * i + c] + lambda * wm[3 * (i + 1) + c];
    }
    push(ts, w, a);
  };
  const size_t N = (size_t)n;
  size_t i = 0;
  for (; i < N - 1; i++)  // the first sample :126-132
    if (t[i] <= time0 && time0 <= t[i + 1]) {
      interp(i, time0);
      break;
    }
  for (i == 0 ? i = 0 : i--; i < N - 1; i++) {  // the middle :135-141
    if (time0 < t[i] && t[i + 1] < time1) push(t[i], wm + 3 * i, am + 3 * i);
    if (t[i + 1] > time1) break;
  }
  for (i == 0 ? i = 0 : i--; i < N - 1; i++)  // the last :144-150
    if (t[i] <= time1 && time1 <= t[i + 1]) {
      interp(i, time1);
      break;
    }
  *n_out = m;
  if (m > cap) return PLV_E_CAPACITY;
  *ok = 1;
  return PLV_OK;
}

void plv_reset_cpi(plv_cpi_accum *acc, const plv_imu_state *imu, double clone_t) {
  if (!acc || !imu) return;
  std::memset(acc, 0, sizeof(*acc));
  acc->clone_t = clone_t;
  acc->R_k2tau[0] = acc->R_k2tau[4] = acc->R_k2tau[8] = 1.0;
  std::copy(imu->bg, imu->bg + 3, acc->b_w_lin);  // setLinearizationPoints(bias_g, bias_a)
  std::copy(imu->ba, imu->ba + 3, acc->b_a_lin);
  std::copy(imu->v, imu->v + 3, acc->v_clone);    // cpi.v = state->imu->vel()
}

int plv_propagate(plv_ctx *ctx, plv_imu_state *imu, const plv_imu_noise *nz, int n_data, const double *t, const double *wm,
                  const double *am, plv_cpi_accum *acc, plv_cpi_record *records, int n, int imu_id, double *Phi_out, double *Qd_out) {
  static_assert(sizeof(plv_imu_state) == IMU_N * 8 && sizeof(plv_cpi_accum) == CA_N * 8 && sizeof(plv_cpi_record) == CR_N * 8,
                "struct images");
  if (!ctx || !imu || !nz || n_data < 2 || !t || !wm || !am) return PLV_E_BADARG;
  if (n != 0 && (n != ctx->cov_n || imu_id < 0 || imu_id + 15 > n)) return PLV_E_BADARG;
  for (int i = 1; i < n_data; ++i)
    if (t[i] < t[i - 1]) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  auto *us = plv_update_state(ctx);
  const size_t nd = (size_t)n_data, nrec = (acc && records) ? (nd - 1) * CR_N : 0;
  // packed: [t nd][wm 3nd][am 3nd][imu 26][cpi 251][Phi 225][Qd 225][records][strip n*15]
  const size_t o_imu = 7 * nd, o_cpi = o_imu + IMU_N, o_phi = o_cpi + CA_N, o_qd = o_phi + 225, o_rec = o_qd + 225, o_strip = o_rec + nrec,
               total = o_strip + (size_t)n * 15;
  TRY(us->eval.reserve(total * 8));
  double *d = us->eval.as<double>();
  std::vector<double> h(o_phi);
  std::copy(t, t + nd, h.begin());
  std::copy(wm, wm + 3 * nd, h.begin() + nd);
  std::copy(am, am + 3 * nd, h.begin() + 4 * nd);
  std::memcpy(h.data() + o_imu, imu, sizeof(*imu));
  if (acc) std::memcpy(h.data() + o_cpi, acc, sizeof(*acc));
  PLV_HIP_CHECK(plv::memcpy_async(d, h.data(), o_phi * 8, hipMemcpyHostToDevice, ctx->stream));
  PropArgs A{};
  A.n_data = n_data;
  A.t = d, A.wm = d + nd, A.am = d + 4 * nd;
  A.imu = d + o_imu;
  A.cpi = acc ? d + o_cpi : nullptr;
  A.records = nrec ? d + o_rec : nullptr;
  A.Phi = d + o_phi, A.Qd = d + o_qd;
  A.sw = nz->sigma_w, A.swb = nz->sigma_wb, A.sa = nz->sigma_a, A.sab = nz->sigma_ab;
  std::copy(nz->gravity, nz->gravity + 3, A.g);
  {
    ProfScope ps(ctx->prof, "propagate_kernel", ctx->stream);
    hipLaunchKernelGGL(propagate_kernel, dim3(1), dim3(256), 0, ctx->stream, A);
  }
  if (n > 0) {
    ++ctx->gather_stamp;  // EKFPropagation rewrites the covariance
    const int blocks = (n * 15 + 255) / 256;
    {
      ProfScope ps(ctx->prof, "ekf_prop_strip_kernel", ctx->stream);
      hipLaunchKernelGGL(ekf_prop_strip_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->d_P.as<double>(), n, imu_id, A.Phi, d + o_strip);
    }
    {
      ProfScope ps(ctx->prof, "ekf_prop_write_kernel", ctx->stream);
      hipLaunchKernelGGL(ekf_prop_write_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->d_P.as<double>(), n, imu_id, A.Phi, A.Qd,
                         d + o_strip);
    }
  }
  PLV_HIP_CHECK(hipGetLastError());
  std::vector<double> back(IMU_N + CA_N + 450 + nrec);
  PLV_HIP_CHECK(plv::memcpy_async(back.data(), d + o_imu, back.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  ctx->prof.collect();
  std::memcpy(imu, back.data(), sizeof(*imu));
  if (acc) std::memcpy(acc, back.data() + IMU_N, sizeof(*acc));
  if (Phi_out) std::copy(back.begin() + IMU_N + CA_N, back.begin() + IMU_N + CA_N + 225, Phi_out);
  if (Qd_out) std::copy(back.begin() + IMU_N + CA_N + 225, back.begin() + IMU_N + CA_N + 450, Qd_out);
  if (nrec) std::memcpy(records, back.data() + IMU_N + CA_N + 450, nrec * 8);
  return PLV_OK;
}

int plv_next_clone_time(const plv_clone_schedule *in, double *clone_time, int *ok) {
  if (!in || !clone_time || !ok || in->clone_freq < 1 || in->n_sensor_times < 0 || (in->n_sensor_times > 0 && !in->sensor_times))
    return PLV_E_BADARG;
  *ok = 0;
  *clone_time = -1;
  if (in->n_clones == 0) {  // SystemManager.cpp:178-181 create a clone RIGHT NOW
    *clone_time = in->state_time;
    *ok = 1;
    return PLV_OK;
  }
  const int freq = in->clone_freq;
  // :189-197 desired time: one period after the newest real clone, never before the state time
  double ct = (in->newest_is_imu_pose ? in->second_newest_clone_time : in->newest_clone_time) + 1.0 / freq;
  ct = ct < in->state_time ? in->meas_t : ct;
  // :200-222 the nearest sensor measurement, at most 10 % of a period early, not before the state time
  double min_t_diff = INFINITY, tmp = ct;
  bool have_meas = false;
  for (int i = in->n_sensor_times - 1; i >= 0; --i) {  // newest first (rbegin -> rend)
    const double sensor_t = in->sensor_times[i] + in->sensor_dt;
    if (in->newest_clone_time < sensor_t) have_meas = true;
    if (sensor_t < ct - 0.1 / freq) break;
    if (std::fabs(sensor_t - ct) < min_t_diff && sensor_t >= in->state_time) {
      min_t_diff = std::fabs(sensor_t - ct);
      tmp = sensor_t;
    }
  }
  if (in->imu_newest_t < tmp || in->imu_oldest_t > tmp) return PLV_OK;             // :241-246
  if (std::isinf(min_t_diff) && !have_meas && !in->wheel_enabled) return PLV_OK;  // :248-253
  *clone_time = tmp;
  *ok = 1;
  return PLV_OK;
}

int plv_closest_clone_time(const plv_state_view *st, int exclude_newest, double t_given, double *clone_t, int *found) {
  if (!st || !clone_t || !found || st->n_clones < 0) return PLV_E_BADARG;
  *found = 0;
  double best = INFINITY;
  const int N = st->n_clones - (exclude_newest ? 1 : 0);
  for (int i = 0; i < N; ++i) {
    const double d = std::fabs(t_given - st->clone_time[i]);
    if (d < best) {
      best = d;
      *clone_t = st->clone_time[i];
      *found = 1;
    }
  }
  return PLV_OK;
}

int plv_cpi_integrate(plv_ctx *ctx, const plv_imu_noise *nz, double t_given, double clone_t, const double *R_GtoI_clone,
                      const double *v_clone, const double *bg, const double *ba, int n_imu, const double *t, const double *wm,
                      const double *am, plv_cpi_record *out, int *ok) {
  if (!ctx || !nz || !R_GtoI_clone || !v_clone || !bg || !ba || !out || !ok || n_imu < 0) return PLV_E_BADARG;
  *ok = 0;
  // State.cpp:378-384: the readings between the two times, reversed when the clone is the later one
  std::vector<double> st_((size_t)n_imu + 2), sw_(3 * ((size_t)n_imu + 2)), sa_(3 * ((size_t)n_imu + 2));
  int m = 0, sel_ok = 0;
  const double lo = clone_t <= t_given ? clone_t : t_given, hi = clone_t <= t_given ? t_given : clone_t;
  TRY(plv_select_imu_readings(n_imu, t, wm, am, lo, hi, n_imu + 2, st_.data(), sw_.data(), sa_.data(), &m, &sel_ok));
  if (!sel_ok || m < 2) return PLV_OK;
  if (!(clone_t <= t_given)) {
    std::reverse(st_.begin(), st_.begin() + m);
    for (int i = 0; i < m / 2; ++i)
      for (int c = 0; c < 3; ++c) {
        std::swap(sw_[3 * (size_t)i + c], sw_[3 * (size_t)(m - 1 - i) + c]);
        std::swap(sa_[3 * (size_t)i + c], sa_[3 * (size_t)(m - 1 - i) + c]);
      }
  }
  (void)hipSetDevice(ctx->device);
  auto *us = plv_update_state(ctx);
  plv_cpi_accum acc;
  std::memset(&acc, 0, sizeof(acc));
  acc.clone_t = clone_t;
  acc.R_k2tau[0] = acc.R_k2tau[4] = acc.R_k2tau[8] = 1.0;
  std::copy(bg, bg + 3, acc.b_w_lin);
  std::copy(ba, ba + 3, acc.b_a_lin);
  std::copy(v_clone, v_clone + 3, acc.v_clone);
  const size_t nd = (size_t)m, o_imu = 7 * nd, o_cpi = o_imu + IMU_N, o_phi = o_cpi + CA_N, o_qd = o_phi + 225, o_rec = o_qd + 225,
               total = o_rec + CR_N;
  TRY(us->eval.reserve(total * 8));
  double *d = us->eval.as<double>();
  std::vector<double> h(o_phi, 0.0);
  std::copy(st_.begin(), st_.begin() + nd, h.begin());
  std::copy(sw_.begin(), sw_.begin() + 3 * nd, h.begin() + nd);
  std::copy(sa_.begin(), sa_.begin() + 3 * nd, h.begin() + 4 * nd);
  h[o_imu + IQ + 3] = 1.0;  // the IMU image is not used in this mode
  std::memcpy(h.data() + o_cpi, &acc, sizeof(acc));
  PLV_HIP_CHECK(plv::memcpy_async(d, h.data(), o_phi * 8, hipMemcpyHostToDevice, ctx->stream));
  PropArgs A{};
  A.n_data = m;
  A.t = d, A.wm = d + nd, A.am = d + 4 * nd;
  A.imu = d + o_imu;
  A.cpi = d + o_cpi;
  A.records = d + o_rec;
  A.Phi = d + o_phi, A.Qd = d + o_qd;
  A.sw = nz->sigma_w, A.swb = nz->sigma_wb, A.sa = nz->sigma_a, A.sab = nz->sigma_ab;
  std::copy(nz->gravity, nz->gravity + 3, A.g);
  A.mode = 1;
  std::copy(R_GtoI_clone, R_GtoI_clone + 9, A.R0);
  A.t_given = t_given;
  {
    ProfScope ps(ctx->prof, "propagate_kernel", ctx->stream);
    hipLaunchKernelGGL(propagate_kernel, dim3(1), dim3(256), 0, ctx->stream, A);
  }
  PLV_HIP_CHECK(hipGetLastError());
  PLV_HIP_CHECK(plv::memcpy_async(out, d + o_rec, sizeof(*out), hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  ctx->prof.collect();
  *ok = 1;
  return PLV_OK;
}

int plv_cov_clone(plv_ctx *ctx, int n, int src_id, int size) {
  if (!ctx || n != ctx->cov_n || n < 1 || src_id < 0 || size < 1 || src_id + size > n) return PLV_E_BADARG;
  if (ctx->cfg.max_state_dim > 0 && n + size > ctx->cfg.max_state_dim) return PLV_E_CAPACITY;
  (void)hipSetDevice(ctx->device);
  const int m = n + size;
  TRY(ctx->d_P2.reserve((size_t)m * m * 8));
  {
    ProfScope ps(ctx->prof, "cov_clone_kernel", ctx->stream);
    hipLaunchKernelGGL(cov_clone_kernel, dim3(std::min(64, (m * m + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_P.as<double>(), n, src_id,
                       size, ctx->d_P2.as<double>());
  }
  PLV_HIP_CHECK(hipGetLastError());
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  ctx->prof.collect();
  std::swap(ctx->d_P, ctx->d_P2);
  ctx->cov_n = m;
  ++ctx->gather_stamp;
  return PLV_OK;
}

}  // extern "C"
