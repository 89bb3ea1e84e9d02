// eval_api.hip — trajectory I/O and the ATE evaluator: the accuracy half of the metric.
//
//   State_Logger::save_trajectory_to_file          REF: PL-VIWO/src/utils/State_Logger.h:166-205
//   ov_eval::Loader::load_data / get_total_length  REF: open_vins/ov_eval/src/utils/Loader.cpp:26-90,388-398
//   AlignUtils::perform_association / align_umeyama REF: open_vins/ov_eval/src/alignment/AlignUtils.cpp:26-91,101-189
//   AlignUtils::get_best_yaw / get_mean            REF: open_vins/ov_eval/src/alignment/AlignUtils.h:53-72
//   AlignTrajectory::align_*                       REF: open_vins/ov_eval/src/alignment/AlignTrajectory.cpp:26-166
//   ResultTrajectory ctor / calculate_ate          REF: open_vins/ov_eval/src/calc/ResultTrajectory.cpp:26-121
//   Statistics::calculate                          REF: open_vins/ov_eval/src/utils/Statistics.h:72-119
//
// File parsing, association and the order statistics are host logic; everything that touches every pose (the two
// Umeyama passes and the per-pose error) runs on the device: a KAIST sequence is ~1e4-1e5 poses, an evaluation
// sweep over many runs is where this is called in a loop.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "plv_ctx.hpp"
#include "update_state.hpp"

namespace plv {
namespace {

#define TRY(expr)                  \
  do {                             \
    int _rc = (expr);              \
    if (_rc != PLV_OK) return _rc; \
  } while (0)

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// block-wide sums of NV values per thread -> out[blockIdx.x][NV]
template <int NV> __device__ void block_sums(double (&v)[NV], double *out) {
  __shared__ double sh[4][NV];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const double s = wsum(v[i]);
    if (lane == 0) sh[w][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) out[(size_t)blockIdx.x * NV + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// pass 1: sums of the positions (get_mean)
__global__ void __launch_bounds__(256) traj_mean_kernel(int n, const double *__restrict__ data, const double *__restrict__ model,
                                                        double *__restrict__ partial) {
  double v[6] = {0, 0, 0, 0, 0, 0};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      v[c] += data[7 * (size_t)i + c];
      v[3 + c] += model[7 * (size_t)i + c];
    }
  }
  block_sums<6>(v, partial);
}

// pass 2: C = sum (m - mu_M)(d - mu_D)^T and sigma2 = sum |d - mu_D|^2   (align_umeyama :33-52)
struct Means {
  double d[3], m[3];
};
__global__ void __launch_bounds__(256) traj_corr_kernel(int n, const double *__restrict__ data, const double *__restrict__ model, Means mu,
                                                        double *__restrict__ partial) {
  double v[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    double d[3], m[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      d[c] = data[7 * (size_t)i + c] - mu.d[c];
      m[c] = model[7 * (size_t)i + c] - mu.m[c];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) v[3 * r + c] += m[r] * d[c];
    v[9] += d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  }
  block_sums<10>(v, partial);
}

struct Align {
  double R[9], t[3], s, qinv[4];  // qinv = Inv(rot_2_quat(R))
};

__device__ void quat_2_rot(const double *q, double *R) {  // quat_ops.h:152-157
  const double a = 2 * q[3] * q[3] - 1;
  const double sk[9] = {0, -q[2], q[1], q[2], 0, -q[0], -q[1], q[0], 0};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) R[3 * r + c] = (r == c ? a : 0.0) - 2 * q[3] * sk[3 * r + c] + 2 * q[r] * q[c];
}

__device__ void quat_mul(const double *q, const double *p, double *o) {  // quat_ops.h:180-195 (the code's L matrix: q4 I - [q x])
  double r[4];
  r[0] = q[3] * p[0] + q[2] * p[1] - q[1] * p[2] + q[0] * p[3];
  r[1] = -q[2] * p[0] + q[3] * p[1] + q[0] * p[2] + q[1] * p[3];
  r[2] = q[1] * p[0] - q[0] * p[1] + q[3] * p[2] + q[2] * p[3];
  r[3] = -q[0] * p[0] - q[1] * p[1] - q[2] * p[2] + q[3] * p[3];
  if (r[3] < 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = -r[i];
  }
  const double nrm = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = r[i] / nrm;
}

__device__ void log_so3_dev(const double *R, double *w) {  // quat_ops.h:273-313
  const double R11 = R[0], R12 = R[1], R13 = R[2], R21 = R[3], R22 = R[4], R23 = R[5], R31 = R[6], R32 = R[7], R33 = R[8];
  const double tr = R11 + R22 + R33;
  if (tr + 1.0 < 1e-10) {
    double k;
    if (fabs(R33 + 1.0) > 1e-5) {
      k = M_PI / sqrt(2.0 + 2.0 * R33);
      w[0] = k * R13, w[1] = k * R23, w[2] = k * (1.0 + R33);
    } else if (fabs(R22 + 1.0) > 1e-5) {
      k = M_PI / sqrt(2.0 + 2.0 * R22);
      w[0] = k * R12, w[1] = k * (1.0 + R22), w[2] = k * R32;
    } else {
      k = M_PI / sqrt(2.0 + 2.0 * R11);
      w[0] = k * (1.0 + R11), w[1] = k * R21, w[2] = k * R31;
    }
    return;
  }
  double mag;
  const double tr_3 = tr - 3.0;
  if (tr_3 < -1e-7) {
    const double theta = acos((tr - 1.0) / 2.0);
    mag = theta / (2.0 * sin(theta));
  } else {
    mag = 0.5 - tr_3 / 12.0;
  }
  w[0] = mag * (R32 - R23), w[1] = mag * (R13 - R31), w[2] = mag * (R21 - R12);
}

// ResultTrajectory ctor :72-82 (the aligned estimate) + calculate_ate :91-115 (errors), thread per pose
__global__ void __launch_bounds__(256) traj_ate_kernel(int n, const double *__restrict__ est, const double *__restrict__ gt, Align A,
                                                       double *__restrict__ aligned, double *__restrict__ ori_err,
                                                       double *__restrict__ pos_err) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double *e = est + 7 * (size_t)i, *g = gt + 7 * (size_t)i;
  double p[3], q[4];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    // s * R * p + t: Eigen evaluates (s * R) * p
    p[r] = (A.s * A.R[3 * r] * e[0] + A.s * A.R[3 * r + 1] * e[1] + A.s * A.R[3 * r + 2] * e[2]) + A.t[r];
  }
  quat_mul(e + 3, A.qinv, q);
  double Re[9], Rg[9], eR[9], w[3];
  quat_2_rot(q, Re);
  quat_2_rot(g + 3, Rg);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) eR[3 * r + c] = Re[r] * Rg[c] + Re[3 + r] * Rg[3 + c] + Re[6 + r] * Rg[6 + c];  // Re^T Rg
  log_so3_dev(eR, w);
  ori_err[i] = 180.0 / M_PI * sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const double dx = g[0] - p[0], dy = g[1] - p[1], dz = g[2] - p[2];
  pos_err[i] = sqrt(dx * dx + dy * dy + dz * dz);
  if (aligned) {
#pragma unroll
    for (int r = 0; r < 3; ++r) aligned[7 * (size_t)i + r] = p[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) aligned[7 * (size_t)i + 3 + r] = q[r];
  }
}

// ---------------------------------------------------------------------------------------- host 3x3 helpers
void h_quat_2_rot(const double *q, double *R) {
  const double a = 2 * q[3] * q[3] - 1;
  const double sk[9] = {0, -q[2], q[1], q[2], 0, -q[0], -q[1], q[0], 0};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R[3 * r + c] = (r == c ? a : 0.0) - 2 * q[3] * sk[3 * r + c] + 2 * q[r] * q[c];
}

void h_rot_2_quat(const double *rot, double *q) {  // quat_ops.h:88-120
  auto R = [&](int r, int c) { return rot[3 * r + c]; };
  const double T = R(0, 0) + R(1, 1) + R(2, 2);
  if (R(0, 0) >= T && R(0, 0) >= R(1, 1) && R(0, 0) >= R(2, 2)) {
    q[0] = std::sqrt((1 + 2 * R(0, 0) - T) / 4);
    q[1] = (1 / (4 * q[0])) * (R(0, 1) + R(1, 0));
    q[2] = (1 / (4 * q[0])) * (R(0, 2) + R(2, 0));
    q[3] = (1 / (4 * q[0])) * (R(1, 2) - R(2, 1));
  } else if (R(1, 1) >= T && R(1, 1) >= R(0, 0) && R(1, 1) >= R(2, 2)) {
    q[1] = std::sqrt((1 + 2 * R(1, 1) - T) / 4);
    q[0] = (1 / (4 * q[1])) * (R(0, 1) + R(1, 0));
    q[2] = (1 / (4 * q[1])) * (R(1, 2) + R(2, 1));
    q[3] = (1 / (4 * q[1])) * (R(2, 0) - R(0, 2));
  } else if (R(2, 2) >= T && R(2, 2) >= R(0, 0) && R(2, 2) >= R(1, 1)) {
    q[2] = std::sqrt((1 + 2 * R(2, 2) - T) / 4);
    q[0] = (1 / (4 * q[2])) * (R(0, 2) + R(2, 0));
    q[1] = (1 / (4 * q[2])) * (R(1, 2) + R(2, 1));
    q[3] = (1 / (4 * q[2])) * (R(0, 1) - R(1, 0));
  } else {
    q[3] = std::sqrt((1 + T) / 4);
    q[0] = (1 / (4 * q[3])) * (R(1, 2) - R(2, 1));
    q[1] = (1 / (4 * q[3])) * (R(2, 0) - R(0, 2));
    q[2] = (1 / (4 * q[3])) * (R(0, 1) - R(1, 0));
  }
  if (q[3] < 0)
    for (int i = 0; i < 4; ++i) q[i] = -q[i];
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= n;
}

double det3(const double *a) {
  return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}

// Singular value decomposition C = U diag(D) V^T of a 3x3 by one-sided Jacobi (Hestenes), singular values descending
// like Eigen::JacobiSVD.  A null singular direction is completed by the cross product; U S V^T with the reference's
// S = diag(1, 1, sign(det U det V)) does not depend on that choice.
void svd3(const double *C, double *U, double *D, double *V) {
  double A[9];
  std::copy(C, C + 9, A);
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int r = 0; r < 3; ++r) {
          al += A[3 * r + p] * A[3 * r + p];
          be += A[3 * r + q] * A[3 * r + q];
          ga += A[3 * r + p] * A[3 * r + q];
        }
        if (ga == 0.0 || std::fabs(ga) <= 1e-300) continue;
        off = std::max(off, std::fabs(ga) / std::sqrt(std::max(al * be, 1e-300)));
        const double zeta = (be - al) / (2 * ga);
        const double tt = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
        const double c = 1 / std::sqrt(1 + tt * tt), s = c * tt;
        for (int r = 0; r < 3; ++r) {
          const double ap = A[3 * r + p], aq = A[3 * r + q];
          A[3 * r + p] = c * ap - s * aq;
          A[3 * r + q] = s * ap + c * aq;
          const double vp = V[3 * r + p], vq = V[3 * r + q];
          V[3 * r + p] = c * vp - s * vq;
          V[3 * r + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-16) break;
  }
  double sig[3];
  for (int c = 0; c < 3; ++c) sig[c] = std::sqrt(A[c] * A[c] + A[3 + c] * A[3 + c] + A[6 + c] * A[6 + c]);
  int ord[3] = {0, 1, 2};
  std::sort(ord, ord + 3, [&](int a, int b) { return sig[a] > sig[b]; });
  double Vs[9];
  const double big = std::max(sig[ord[0]], 1e-300);
  for (int k = 0; k < 3; ++k) {
    const int c = ord[k];
    D[k] = sig[c];
    for (int r = 0; r < 3; ++r) {
      Vs[3 * r + k] = V[3 * r + c];
      U[3 * r + k] = sig[c] > 1e-14 * big ? A[3 * r + c] / sig[c] : 0.0;
    }
  }
  std::copy(Vs, Vs + 9, V);
  // complete null columns of U (rank-deficient C: planar or collinear trajectories)
  auto col = [&](double *M, int k, double *o) { o[0] = M[k], o[1] = M[3 + k], o[2] = M[6 + k]; };
  auto setc = [&](double *M, int k, const double *o) { M[k] = o[0], M[3 + k] = o[1], M[6 + k] = o[2]; };
  auto nrm = [](const double *o) { return std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]); };
  double u0[3], u1[3], u2[3];
  col(U, 0, u0), col(U, 1, u1);
  if (nrm(u0) == 0) u0[0] = 1, u0[1] = 0, u0[2] = 0, setc(U, 0, u0);
  if (nrm(u1) == 0) {  // any unit vector orthogonal to u0
    const double e[3] = {std::fabs(u0[0]) < 0.9 ? 1.0 : 0.0, std::fabs(u0[0]) < 0.9 ? 0.0 : 1.0, 0.0};
    u1[0] = u0[1] * e[2] - u0[2] * e[1], u1[1] = u0[2] * e[0] - u0[0] * e[2], u1[2] = u0[0] * e[1] - u0[1] * e[0];
    const double k = nrm(u1);
    for (double &x : u1) x /= k;
    setc(U, 1, u1);
  }
  col(U, 2, u2);
  if (nrm(u2) == 0) {
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1], u2[1] = u0[2] * u1[0] - u0[0] * u1[2], u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
    setc(U, 2, u2);
  }
}

void stats_of(std::vector<double> v, plv_stats *st) {  // Statistics::calculate
  *st = plv_stats{0, 0, 0, 0, 0, 0, 0};
  std::sort(v.begin(), v.end());
  if (v.empty()) return;
  const size_t n = v.size();
  st->min = v.front();
  st->max = v.back();
  st->median = n == 1 ? v[0] : (n % 2 == 1 ? v[n / 2] : 0.5 * (v[n / 2 - 1] + v[n / 2]));
  double mean = 0, rmse = 0;
  for (double x : v) {
    mean += x;
    rmse += x * x;
  }
  mean /= n;
  st->mean = mean;
  st->rmse = std::sqrt(rmse / n);
  double sd = 0;
  for (double x : v) sd += std::pow(x - mean, 2);
  st->std = std::sqrt(sd / (n - 1));  // n == 1: 0 / 0 like the reference
  st->ninetynine = mean + 2.326 * st->std;
}

}  // namespace
}  // namespace plv

using namespace plv;

extern "C" {

int plv_traj_header(char *buf, int cap) {
  static const char *h = "# timestamp(s) tx ty tz qx qy qz qw Pr11 Pr12 Pr13 Pr22 Pr23 Pr33 Pt11 Pt12 Pt13 Pt22 Pt23 Pt33\n";
  const int need = (int)std::strlen(h);
  if (!buf || cap <= need) return PLV_E_BADARG;
  std::memcpy(buf, h, need + 1);
  return need;
}

int plv_traj_format(char *buf, int cap, double t, const double *p, const double *q, const double *P) {
  if (!buf || !p || !q || cap < 1) return PLV_E_BADARG;
  // ios::fixed, precision 6 for time / position / quaternion, precision 10 for the covariance terms (:192-204)
  int n = std::snprintf(buf, cap, "%.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f", t, p[0], p[1], p[2], q[0], q[1], q[2], q[3]);
  if (n < 0 || n >= cap) return PLV_E_CAPACITY;
  if (P) {  // 6x6 row-major marginal of the IMU pose [ori; pos]
    auto at = [&](int r, int c) { return P[6 * r + c]; };
    const int m = std::snprintf(buf + n, cap - n, " %.10f %.10f %.10f %.10f %.10f %.10f %.10f %.10f %.10f %.10f %.10f %.10f", at(0, 0),
                                at(0, 1), at(0, 2), at(1, 1), at(1, 2), at(2, 2), at(3, 3), at(3, 4), at(3, 5), at(4, 4), at(4, 5), at(5, 5));
    if (m < 0 || m >= cap - n) return PLV_E_CAPACITY;
    n += m;
  }
  if (n + 1 >= cap) return PLV_E_CAPACITY;
  buf[n++] = '\n';
  buf[n] = 0;
  return n;
}

int plv_traj_load(const char *path, int cap, double *times, double *poses, double *cov_ori, double *cov_pos, int *n_out, int *n_cov_out) {
  if (!path || !n_out) return PLV_E_BADARG;
  std::ifstream file(path);
  if (!file.is_open()) {
    set_last_error("plv_traj_load: unable to open %s", path);
    return PLV_E_BADARG;
  }
  int n = 0, ncov = 0;
  std::string line;
  while (std::getline(file, line)) {
    if (!line.find("#")) continue;  // '#' in the first column only (:44-45)
    int i = 0;
    std::istringstream s(line);
    std::string field;
    double data[20] = {0};
    while (std::getline(s, field, ' ')) {
      if (field.empty() || i >= 20) continue;
      data[i++] = std::atof(field.c_str());
    }
    if (i < 8) continue;
    if (times && n < cap) {
      times[n] = data[0];
      if (poses) std::copy(data + 1, data + 8, poses + 7 * (size_t)n);
    }
    if (i >= 20) {
      if (cov_ori && cov_pos && ncov < cap) {
        // symmetric fill, then 0.5 (c + c^T) which leaves it unchanged
        const double o[9] = {data[8], data[9], data[10], data[9], data[11], data[12], data[10], data[12], data[13]};
        const double q[9] = {data[14], data[15], data[16], data[15], data[17], data[18], data[16], data[18], data[19]};
        std::copy(o, o + 9, cov_ori + 9 * (size_t)ncov);
        std::copy(q, q + 9, cov_pos + 9 * (size_t)ncov);
      }
      ++ncov;
    }
    ++n;
  }
  *n_out = n;
  if (n_cov_out) *n_cov_out = ncov;
  if (n == 0) {
    set_last_error("plv_traj_load: could not parse any data from %s", path);
    return PLV_E_BADARG;  // the reference exits (:78-82)
  }
  return (times && n > cap) ? PLV_E_CAPACITY : PLV_OK;
}

double plv_traj_length(int n, const double *poses) {
  double d = 0;
  for (int i = 1; i < n; ++i) {
    const double *a = poses + 7 * (size_t)i, *b = a - 7;
    d += std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
  }
  return d;
}

int plv_traj_associate(double offset, double max_difference, int n_est, const double *est_times, int n_gt, const double *gt_times,
                       int *est_idx, int *gt_idx, int *n_match) {
  if (n_est < 0 || n_gt < 0 || !n_match || (n_est && !est_times) || (n_gt && !gt_times) || !est_idx || !gt_idx) return PLV_E_BADARG;
  int gp = 0, m = 0;
  for (int i = 0; i < n_est; ++i) {
    double best = max_difference;
    int best_gt = -1;
    const double te = est_times[i] + offset;
    while (gp < n_gt && gt_times[gp] < te && std::fabs(gt_times[gp] - te) > max_difference) ++gp;
    while (gp < n_gt && std::fabs(gt_times[gp] - te) <= max_difference) {
      if (std::fabs(gt_times[gp] - te) >= best) break;
      best = std::fabs(gt_times[gp] - te);
      best_gt = gp;
      ++gp;
    }
    if (best_gt != -1) {
      est_idx[m] = i;
      gt_idx[m] = best_gt;
      ++m;
    }
  }
  *n_match = m;
  return PLV_OK;
}

static int device_align(plv_ctx *ctx, int method, int n, const double *d_est, const double *d_gt, const double *h_est, const double *h_gt,
                        int n_aligned, double *R, double *t, double *s) {
  *s = 1;
  auto single = [&](bool yaw) {  // align_posyaw_single / align_se3_single
    double g[9], e[9];
    h_quat_2_rot(h_gt + 3, g);
    h_quat_2_rot(h_est + 3, e);
    // g_rot = quat_2_Rot(q_gt)^T, est_rot = quat_2_Rot(q_es)^T
    if (yaw) {
      double CR[9];  // est_rot * g_rot^T = Re^T Rg
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) CR[3 * r + c] = e[r] * g[c] + e[3 + r] * g[3 + c] + e[6 + r] * g[6 + c];
      const double th = std::atan2(CR[1] - CR[3], CR[0] + CR[4]);
      const double ct = std::cos(th), st = std::sin(th);
      const double Rz[9] = {ct, -st, 0, st, ct, 0, 0, 0, 1};
      std::copy(Rz, Rz + 9, R);
    } else {
      for (int r = 0; r < 3; ++r)  // g_rot * est_rot^T = Rg^T Re
        for (int c = 0; c < 3; ++c) R[3 * r + c] = g[r] * e[c] + g[3 + r] * e[3 + c] + g[6 + r] * e[6 + c];
    }
    for (int r = 0; r < 3; ++r) t[r] = h_gt[r] - (R[3 * r] * h_est[0] + R[3 * r + 1] * h_est[1] + R[3 * r + 2] * h_est[2]);
  };
  if (method == PLV_ALIGN_NONE) {
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::copy(I, I + 9, R);
    t[0] = t[1] = t[2] = 0;
    return PLV_OK;
  }
  if (method == PLV_ALIGN_POSYAW_SINGLE || (method == PLV_ALIGN_POSYAW && n_aligned == 1)) return single(true), PLV_OK;
  if (method == PLV_ALIGN_SE3_SINGLE || (method == PLV_ALIGN_SE3 && n_aligned == 1)) return single(false), PLV_OK;
  if (method != PLV_ALIGN_POSYAW && method != PLV_ALIGN_SE3 && method != PLV_ALIGN_SIM3) return PLV_E_BADARG;
  // ---- align_umeyama(data = est, model = gt)
  auto *us = plv_update_state(ctx);
  const int blocks = std::min(256, (n + 255) / 256);
  TRY(us->tri.reserve((size_t)blocks * 16 * sizeof(double) + 64));
  double *d_part = us->tri.as<double>();
  std::vector<double> part((size_t)blocks * 10);
  {
    ProfScope ps(ctx->prof, "traj_mean_kernel", ctx->stream);
    hipLaunchKernelGGL(traj_mean_kernel, dim3(blocks), dim3(256), 0, ctx->stream, n, d_est, d_gt, d_part);
  }
  PLV_HIP_CHECK(plv::memcpy_async(part.data(), d_part, (size_t)blocks * 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  Means mu{};
  for (int b = 0; b < blocks; ++b)
    for (int c = 0; c < 3; ++c) {
      mu.d[c] += part[6 * (size_t)b + c];
      mu.m[c] += part[6 * (size_t)b + 3 + c];
    }
  for (int c = 0; c < 3; ++c) {
    mu.d[c] /= n;
    mu.m[c] /= n;
  }
  {
    ProfScope ps(ctx->prof, "traj_corr_kernel", ctx->stream);
    hipLaunchKernelGGL(traj_corr_kernel, dim3(blocks), dim3(256), 0, ctx->stream, n, d_est, d_gt, mu, d_part);
  }
  PLV_HIP_CHECK(plv::memcpy_async(part.data(), d_part, (size_t)blocks * 10 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  double C[9] = {0}, sigma2 = 0;
  for (int b = 0; b < blocks; ++b) {
    for (int c = 0; c < 9; ++c) C[c] += part[10 * (size_t)b + c];
    sigma2 += part[10 * (size_t)b + 9];
  }
  for (double &x : C) x *= 1.0 / n;
  sigma2 *= 1.0 / n;
  double U[9], D[3], V[9];
  svd3(C, U, D, V);
  const double S22 = det3(U) * det3(V) < 0 ? -1.0 : 1.0;
  if (method == PLV_ALIGN_POSYAW) {
    // rot_C = n * C^T; theta = atan2(rot_C(0,1) - rot_C(1,0), rot_C(0,0) + rot_C(1,1))
    const double th = std::atan2(n * C[3] - n * C[1], n * C[0] + n * C[4]);
    const double ct = std::cos(th), st = std::sin(th);
    const double Rz[9] = {ct, -st, 0, st, ct, 0, 0, 0, 1};
    std::copy(Rz, Rz + 9, R);
  } else {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) R[3 * r + c] = U[3 * r] * V[3 * c] + U[3 * r + 1] * V[3 * c + 1] + S22 * U[3 * r + 2] * V[3 * c + 2];
  }
  if (method == PLV_ALIGN_SIM3) *s = 1.0 / sigma2 * (D[0] + D[1] + S22 * D[2]);
  for (int r = 0; r < 3; ++r) t[r] = mu.m[r] - (*s * R[3 * r] * mu.d[0] + *s * R[3 * r + 1] * mu.d[1] + *s * R[3 * r + 2] * mu.d[2]);
  return PLV_OK;
}

int plv_traj_ate(plv_ctx *ctx, int method, int n, const double *est_poses, const double *gt_poses, int n_aligned, double *R_out,
                 double *t_out, double *s_out, double *aligned, double *ori_err, double *pos_err, plv_stats *ori, plv_stats *pos) {
  if (!ctx || !est_poses || !gt_poses || n < 1) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  auto *us = plv_update_state(ctx);
  const size_t bytes = (size_t)n * 7 * sizeof(double);
  TRY(us->eval.reserve(bytes * 3 + (size_t)n * 2 * sizeof(double)));
  double *d_est = us->eval.as<double>(), *d_gt = d_est + (size_t)n * 7, *d_al = d_gt + (size_t)n * 7, *d_oe = d_al + (size_t)n * 7,
         *d_pe = d_oe + n;
  PLV_HIP_CHECK(plv::memcpy_async(d_est, est_poses, bytes, hipMemcpyHostToDevice, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(d_gt, gt_poses, bytes, hipMemcpyHostToDevice, ctx->stream));
  Align A{};
  TRY(device_align(ctx, method, n, d_est, d_gt, est_poses, gt_poses, n_aligned, A.R, A.t, &A.s));
  double q[4];
  h_rot_2_quat(A.R, q);
  A.qinv[0] = -q[0], A.qinv[1] = -q[1], A.qinv[2] = -q[2], A.qinv[3] = q[3];
  {
    ProfScope ps(ctx->prof, "traj_ate_kernel", ctx->stream);
    hipLaunchKernelGGL(traj_ate_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, d_est, d_gt, A, d_al, d_oe, d_pe);
  }
  PLV_HIP_CHECK(hipGetLastError());
  std::vector<double> oe(n), pe(n);
  PLV_HIP_CHECK(plv::memcpy_async(oe.data(), d_oe, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(pe.data(), d_pe, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (aligned) PLV_HIP_CHECK(plv::memcpy_async(aligned, d_al, bytes, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  ctx->prof.collect();
  if (R_out) std::copy(A.R, A.R + 9, R_out);
  if (t_out) std::copy(A.t, A.t + 3, t_out);
  if (s_out) *s_out = A.s;
  if (ori_err) std::copy(oe.begin(), oe.end(), ori_err);
  if (pos_err) std::copy(pe.begin(), pe.end(), pos_err);
  if (ori) stats_of(oe, ori);
  if (pos) stats_of(pe, pos);
  return PLV_OK;
}

}  // extern "C"
