// so3_dev.hpp — small fp64 vector / rotation / JPL-quaternion helpers shared by the propagation and wheel kernels.
// REF for every formula: open_vins/ov_core/src/utils/quat_ops.h (line ranges at the functions).
#pragma once
#include <hip/hip_runtime.h>

namespace plv {
namespace so3 {

struct D3 {
  double x, y, z;
};
struct DM3 {
  double m[9];
};
struct DQ {
  double x, y, z, w;
};
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ D3 ld3(const double *p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(double *p, D3 a) { p[0] = a.x, p[1] = a.y, p[2] = a.z; }
__device__ __forceinline__ double nrm3(D3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ DM3 skewm(D3 w) { return {{0, -w.z, w.y, w.z, 0, -w.x, -w.y, w.x, 0}}; }
__device__ __forceinline__ DM3 eyem() { return {{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
__device__ __forceinline__ DM3 mmul(const DM3 &a, const DM3 &b) {
  DM3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return c;
}
__device__ __forceinline__ DM3 mtr(const DM3 &a) { return {{a.m[0], a.m[3], a.m[6], a.m[1], a.m[4], a.m[7], a.m[2], a.m[5], a.m[8]}}; }
__device__ __forceinline__ DM3 madd(const DM3 &a, const DM3 &b) {
  DM3 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.m[i] = a.m[i] + b.m[i];
  return c;
}
__device__ __forceinline__ DM3 msub(const DM3 &a, const DM3 &b) {
  DM3 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.m[i] = a.m[i] - b.m[i];
  return c;
}
__device__ __forceinline__ DM3 mscale(double s, const DM3 &a) {
  DM3 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.m[i] = s * a.m[i];
  return c;
}
__device__ __forceinline__ D3 mvec(const DM3 &a, D3 v) {
  return {a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z, a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
__device__ __forceinline__ DM3 outer(D3 a, D3 b) { return {{a.x * b.x, a.x * b.y, a.x * b.z, a.y * b.x, a.y * b.y, a.y * b.z, a.z * b.x, a.z * b.y, a.z * b.z}}; }

__device__ inline DM3 q2R(DQ q) {  // quat_ops.h:152-157
  const D3 v{q.x, q.y, q.z};
  return madd(msub(mscale(2 * q.w * q.w - 1, eyem()), mscale(2 * q.w, skewm(v))), mscale(2.0, outer(v, v)));
}
__device__ inline DQ qmul(DQ q, DQ p) {  // quat_ops.h:180-195
  DQ r;
  r.x = q.w * p.x + q.z * p.y - q.y * p.z + q.x * p.w;
  r.y = -q.z * p.x + q.w * p.y + q.x * p.z + q.y * p.w;
  r.z = q.y * p.x - q.x * p.y + q.w * p.z + q.z * p.w;
  r.w = -q.x * p.x - q.y * p.y - q.z * p.z + q.w * p.w;
  if (r.w < 0) r = {-r.x, -r.y, -r.z, -r.w};
  const double n = sqrt(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
  return {r.x / n, r.y / n, r.z / n, r.w / n};
}
__device__ inline DQ qnorm(DQ q) {  // quat_ops.h:496-501
  if (q.w < 0) q = {-q.x, -q.y, -q.z, -q.w};
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return {q.x / n, q.y / n, q.z / n, q.w / n};
}
__device__ inline DQ omega_times(D3 w, DQ q) {  // Omega(w) * q, quat_ops.h:482-489
  return {-(-w.z * q.y + w.y * q.z) + w.x * q.w, -(w.z * q.x - w.x * q.z) + w.y * q.w, -(-w.y * q.x + w.x * q.y) + w.z * q.w,
          -w.x * q.x - w.y * q.y - w.z * q.z};
}
__device__ inline DQ qaxpy(DQ a, double s, DQ b) { return {a.x + s * b.x, a.y + s * b.y, a.z + s * b.z, a.w + s * b.w}; }
__device__ inline DM3 Jl(D3 w) {  // quat_ops.h:515-525
  const double th = nrm3(w);
  if (th < 1e-6) return eyem();
  const D3 a = (1.0 / th) * w;
  const double sth = sin(th) / th;
  return madd(madd(mscale(sth, eyem()), mscale(1 - sth, outer(a, a))), mscale((1 - cos(th)) / th, skewm(a)));
}

__device__ inline DM3 exp3(D3 w) {  // exp_so3, quat_ops.h:231-251
  const DM3 wx = skewm(w);
  const double theta = nrm3(w);
  double A, B;
  if (theta < 1e-7) {
    A = 1;
    B = 0.5;
  } else {
    A = sin(theta) / theta;
    B = (1 - cos(theta)) / (theta * theta);
  }
  if (theta == 0) return eyem();
  return madd(madd(eyem(), mscale(A, wx)), mscale(B, mmul(wx, wx)));
}
__device__ inline D3 log3(const DM3 &Rm) {  // log_so3, quat_ops.h:273-313
  const double *R = Rm.m;
  const double R11 = R[0], R12 = R[1], R13 = R[2], R21 = R[3], R22 = R[4], R23 = R[5], R31 = R[6], R32 = R[7], R33 = R[8];
  const double tr = R11 + R22 + R33;
  if (tr + 1.0 < 1e-10) {
    if (fabs(R33 + 1.0) > 1e-5) return (M_PI / sqrt(2.0 + 2.0 * R33)) * D3{R13, R23, 1.0 + R33};
    if (fabs(R22 + 1.0) > 1e-5) return (M_PI / sqrt(2.0 + 2.0 * R22)) * D3{R12, 1.0 + R22, R32};
    return (M_PI / sqrt(2.0 + 2.0 * R11)) * D3{1.0 + R11, R21, R31};
  }
  double mag;
  const double tr_3 = tr - 3.0;
  if (tr_3 < -1e-7) {
    const double theta = acos((tr - 1.0) / 2.0);
    mag = theta / (2.0 * sin(theta));
  } else {
    mag = 0.5 - tr_3 / 12.0;
  }
  return mag * D3{R32 - R23, R13 - R31, R21 - R12};
}
__device__ inline DQ R2q(const DM3 &rot) {  // rot_2_quat, quat_ops.h:88-120
  auto R = [&](int r, int c) { return rot.m[3 * r + c]; };
  const double T = R(0, 0) + R(1, 1) + R(2, 2);
  double q[4];
  if (R(0, 0) >= T && R(0, 0) >= R(1, 1) && R(0, 0) >= R(2, 2)) {
    q[0] = sqrt((1 + (2 * R(0, 0)) - T) / 4);
    q[1] = (1 / (4 * q[0])) * (R(0, 1) + R(1, 0));
    q[2] = (1 / (4 * q[0])) * (R(0, 2) + R(2, 0));
    q[3] = (1 / (4 * q[0])) * (R(1, 2) - R(2, 1));
  } else if (R(1, 1) >= T && R(1, 1) >= R(0, 0) && R(1, 1) >= R(2, 2)) {
    q[1] = sqrt((1 + (2 * R(1, 1)) - T) / 4);
    q[0] = (1 / (4 * q[1])) * (R(0, 1) + R(1, 0));
    q[2] = (1 / (4 * q[1])) * (R(1, 2) + R(2, 1));
    q[3] = (1 / (4 * q[1])) * (R(2, 0) - R(0, 2));
  } else if (R(2, 2) >= T && R(2, 2) >= R(0, 0) && R(2, 2) >= R(1, 1)) {
    q[2] = sqrt((1 + (2 * R(2, 2)) - T) / 4);
    q[0] = (1 / (4 * q[2])) * (R(0, 2) + R(2, 0));
    q[1] = (1 / (4 * q[2])) * (R(1, 2) + R(2, 1));
    q[3] = (1 / (4 * q[2])) * (R(0, 1) - R(1, 0));
  } else {
    q[3] = sqrt((1 + T) / 4);
    q[0] = (1 / (4 * q[3])) * (R(1, 2) - R(2, 1));
    q[1] = (1 / (4 * q[3])) * (R(2, 0) - R(0, 2));
    q[2] = (1 / (4 * q[3])) * (R(0, 1) - R(1, 0));
  }
  if (q[3] < 0) {
    q[0] = -q[0], q[1] = -q[1], q[2] = -q[2], q[3] = -q[3];
  }
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  return {q[0] / n, q[1] / n, q[2] / n, q[3] / n};
}

}  // namespace so3
}  // namespace plv
