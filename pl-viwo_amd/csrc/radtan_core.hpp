// radtan_core.hpp — cv::undistortPoints for the radtan model (5 fixed-point iterations in fp64, float in / out), one source for the
// device kernels (lk_kernel tail, undistort_kernel) and for host code that has a handful of points at hand (the end points of
// the line segments kept in a frame): plain IEEE add / multiply / divide, so host and device agree to the bit.
//   REF: open_vins/ov_core/src/cam/CamRadtan.h:99-120 (undistort_f -> cv::undistortPoints), SURVEY Appendix A.
#pragma once
#include <hip/hip_runtime.h>

namespace plv {

__host__ __device__ inline void undistort_radtan(const double *K, float u, float v, float &xn, float &yn) {
  const double fx = K[0], fy = K[1], cx = K[2], cy = K[3], k1 = K[4], k2 = K[5], p1 = K[6], p2 = K[7];
  const double ifx = 1. / fx, ify = 1. / fy;
  double x = ((double)u - cx) * ifx, y = ((double)v - cy) * ify;
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; ++j) {
    double r2 = x * x + y * y;
    double icdist = 1. / (1 + ((0. * r2 + k2) * r2 + k1) * r2);
    if (icdist < 0) {
      x = x0;
      y = y0;
      break;
    }
    double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
    double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  xn = (float)x;
  yn = (float)y;
}


}  // namespace plv
