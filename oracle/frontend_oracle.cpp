// ORACLE — TEST INFRASTRUCTURE ONLY (see dense.h).  PARITY UNPINNED.
//
// CPU restatement of the point front-end TrackKLT drives through OpenCV 4.2:
//   cv::equalizeHist                REF call site: open_vins/ov_core/src/track/TrackKLT.cpp:59
//   cv::buildOpticalFlowPyramid     REF call site: TrackKLT.cpp:71   (15x15, maxLevel 5, with derivatives)
//   cv::calcOpticalFlowPyrLK        REF call site: TrackKLT.cpp:857-858 (30 it / 0.01, USE_INITIAL_FLOW)
//   cv::undistortPoints (radtan)    REF call site: ov_core/src/cam/CamRadtan.h:99-120
//   cv::findFundamentalMat RANSAC   REF call site: TrackKLT.cpp:870-873 (2/f_max, 0.999)
//   TrackKLT::perform_matching      REF: TrackKLT.cpp:829-886
// OpenCV (4.2, un-vendored: README.md:19) is NOT under /root/reference and not installed, so these
// follow OpenCV's published algorithms and the call contracts in SURVEY.md Appendix A.  Two
// deliberate, documented choices (DESIGN.md §"Front-end arithmetic"):
//   * LK's 2x2 normal-equation sums are accumulated exactly in int64 and rounded to float once
//     (OpenCV's scalar path accumulates in float, its SIMD path in int32 lanes — all three agree to
//     float rounding); this makes the result independent of summation order, so the wave-parallel
//     HIP kernel can be compared BIT-EXACTLY with this oracle;
//   * RANSAC draws its 7-point subsets from a counter-based hash RNG (hypothesis h, draw t), not
//     from cv::RNG's sequential stream, so all hypotheses can be evaluated in parallel and the
//     adaptive stopping rule is replayed afterwards in hypothesis order.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * len - 2 - p;
  }
  return p;
}
inline int cv_round(float v) { return (int)lrintf(v); }
inline int cv_round_d(double v) { return (int)lrint(v); }
inline int cv_floor(float v) { return (int)floorf(v); }

// ---------------------------------------------------------------- equalizeHist
void equalize_hist(const uint8_t *src, int w, int h, int stride, uint8_t *dst, int dstride) {
  int hist[256] = {0};
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) hist[src[y * stride + x]]++;
  int total = w * h;
  int i = 0;
  while (!hist[i]) ++i;
  uint8_t lut[256];
  if (hist[i] == total) {
    for (int y = 0; y < h; ++y) memset(dst + y * dstride, i, w);
    return;
  }
  float scale = (256 - 1.f) / (total - hist[i]);
  int sum = 0;
  for (int j = 0; j <= i; ++j) lut[j] = 0;
  for (++i; i < 256; ++i) {
    sum += hist[i];
    int v = cv_round(sum * scale);
    lut[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
  }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) dst[y * dstride + x] = lut[src[y * stride + x]];
}

// ---------------------------------------------------------------- pyrDown (5x5 [1 4 6 4 1]^2 / 256)
void pyr_down(const uint8_t *src, int w, int h, uint8_t *dst, int dw, int dh) {
  std::vector<int> rowbuf((size_t)h * dw);
  for (int y = 0; y < h; ++y) {
    const uint8_t *s = src + (size_t)y * w;
    for (int x = 0; x < dw; ++x) {
      int c = 2 * x;
      rowbuf[(size_t)y * dw + x] = s[reflect101(c - 2, w)] + 4 * s[reflect101(c - 1, w)] + 6 * s[reflect101(c, w)] +
                                   4 * s[reflect101(c + 1, w)] + s[reflect101(c + 2, w)];
    }
  }
  for (int y = 0; y < dh; ++y) {
    int c = 2 * y;
    const int *r0 = &rowbuf[(size_t)reflect101(c - 2, h) * dw], *r1 = &rowbuf[(size_t)reflect101(c - 1, h) * dw],
              *r2 = &rowbuf[(size_t)reflect101(c, h) * dw], *r3 = &rowbuf[(size_t)reflect101(c + 1, h) * dw],
              *r4 = &rowbuf[(size_t)reflect101(c + 2, h) * dw];
    for (int x = 0; x < dw; ++x) dst[(size_t)y * dw + x] = (uint8_t)((r0[x] + 4 * r1[x] + 6 * r2[x] + 4 * r3[x] + r4[x] + 128) >> 8);
  }
}

// Scharr derivatives (int16 dx,dy interleaved), reflect-101 inside the image (calcSharrDeriv).
void scharr(const uint8_t *img, int w, int h, int16_t *d) {
  for (int y = 0; y < h; ++y) {
    const uint8_t *r0 = img + (size_t)reflect101(y - 1, h) * w, *r1 = img + (size_t)y * w,
                  *r2 = img + (size_t)reflect101(y + 1, h) * w;
    for (int x = 0; x < w; ++x) {
      int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
      int t0m = (r0[xm] + r2[xm]) * 3 + r1[xm] * 10, t0p = (r0[xp] + r2[xp]) * 3 + r1[xp] * 10;
      int t1m = r2[xm] - r0[xm], t1c = r2[x] - r0[x], t1p = r2[xp] - r0[xp];
      d[((size_t)y * w + x) * 2] = (int16_t)(t0p - t0m);
      d[((size_t)y * w + x) * 2 + 1] = (int16_t)((t1m + t1p) * 3 + t1c * 10);
    }
  }
}

struct Pyramid {
  int levels = 0;
  std::vector<int> w, h;
  std::vector<std::vector<uint8_t>> img;
  std::vector<std::vector<int16_t>> der;
};

// buildOpticalFlowPyramid(img, pyr, win, maxLevel): level l+1 exists only while both of its
// dimensions exceed the window.
void build_pyramid(const uint8_t *img0, int w, int h, int stride, int win, int max_level, bool with_deriv, Pyramid &P) {
  P = Pyramid();
  P.w.push_back(w);
  P.h.push_back(h);
  P.img.emplace_back((size_t)w * h);
  for (int y = 0; y < h; ++y) memcpy(&P.img[0][(size_t)y * w], img0 + (size_t)y * stride, w);
  for (int l = 0; l < max_level; ++l) {
    int nw = (P.w[l] + 1) / 2, nh = (P.h[l] + 1) / 2;
    if (nw <= win || nh <= win) break;
    P.w.push_back(nw);
    P.h.push_back(nh);
    P.img.emplace_back((size_t)nw * nh);
    pyr_down(P.img[l].data(), P.w[l], P.h[l], P.img[l + 1].data(), nw, nh);
  }
  P.levels = (int)P.w.size();
  if (with_deriv) {
    P.der.resize(P.levels);
    for (int l = 0; l < P.levels; ++l) {
      P.der[l].resize((size_t)P.w[l] * P.h[l] * 2);
      scharr(P.img[l].data(), P.w[l], P.h[l], P.der[l].data());
    }
  }
}

// image sample with the pyramid's REFLECT_101 padding; derivative sample with CONSTANT(0) padding
inline int px(const Pyramid &P, int l, int x, int y) {
  return P.img[l][(size_t)reflect101(y, P.h[l]) * P.w[l] + reflect101(x, P.w[l])];
}
inline void dpx(const Pyramid &P, int l, int x, int y, int &dx, int &dy) {
  if (x < 0 || y < 0 || x >= P.w[l] || y >= P.h[l]) {
    dx = dy = 0;
    return;
  }
  const int16_t *d = &P.der[l][((size_t)y * P.w[l] + x) * 2];
  dx = d[0];
  dy = d[1];
}

#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

// LKTrackerInvoker for one point, all levels (coarse to fine).
void lk_point(const Pyramid &I, const Pyramid &J, int win, int max_iters, float eps, float px0, float py0, float &nx,
              float &ny, uint8_t &status, int *iters_done) {
  const int W_BITS = 14;
  const float FLT_SCALE = 1.f / (1 << 20);
  const float min_eig_thr = 1e-4f;
  const float half = (win - 1) * 0.5f;
  const double eps2 = (double)std::min(std::max(eps, 0.f), 10.f) * (double)std::min(std::max(eps, 0.f), 10.f);
  const int maxLevel = I.levels - 1;
  status = 1;
  float nextx = nx, nexty = ny;
  std::vector<int16_t> Ibuf((size_t)win * win), dIbuf((size_t)win * win * 2);
  for (int level = maxLevel; level >= 0; --level) {
    const float sc = (float)(1. / (1 << level));
    float prevx = px0 * sc, prevy = py0 * sc;
    if (level == maxLevel) {
      nextx = nx * sc;
      nexty = ny * sc;
    } else {
      nextx = nextx * 2.f;
      nexty = nexty * 2.f;
    }
    const int cols = I.w[level], rows = I.h[level];
    prevx -= half;
    prevy -= half;
    int ipx = cv_floor(prevx), ipy = cv_floor(prevy);
    if (ipx < -win || ipx >= cols || ipy < -win || ipy >= rows) {
      if (level == 0) status = 0;
      continue;
    }
    float a = prevx - ipx, b = prevy - ipy;
    int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
    int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
    int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
    int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
    int64_t iA11 = 0, iA12 = 0, iA22 = 0;
    for (int y = 0; y < win; ++y)
      for (int x = 0; x < win; ++x) {
        int X = ipx + x, Y = ipy + y;
        int ival = DESCALE(px(I, level, X, Y) * iw00 + px(I, level, X + 1, Y) * iw01 + px(I, level, X, Y + 1) * iw10 +
                               px(I, level, X + 1, Y + 1) * iw11,
                           W_BITS - 5);
        int dx00, dy00, dx01, dy01, dx10, dy10, dx11, dy11;
        dpx(I, level, X, Y, dx00, dy00);
        dpx(I, level, X + 1, Y, dx01, dy01);
        dpx(I, level, X, Y + 1, dx10, dy10);
        dpx(I, level, X + 1, Y + 1, dx11, dy11);
        int ixval = DESCALE(dx00 * iw00 + dx01 * iw01 + dx10 * iw10 + dx11 * iw11, W_BITS);
        int iyval = DESCALE(dy00 * iw00 + dy01 * iw01 + dy10 * iw10 + dy11 * iw11, W_BITS);
        Ibuf[y * win + x] = (int16_t)ival;
        dIbuf[(y * win + x) * 2] = (int16_t)ixval;
        dIbuf[(y * win + x) * 2 + 1] = (int16_t)iyval;
        iA11 += (int64_t)ixval * ixval;
        iA12 += (int64_t)ixval * iyval;
        iA22 += (int64_t)iyval * iyval;
      }
    float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE, A22 = (float)iA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
    if (minEig < min_eig_thr || D < FLT_EPSILON) {
      if (level == 0) status = 0;
      continue;
    }
    D = 1.f / D;
    nextx -= half;
    nexty -= half;
    float pdx = 0.f, pdy = 0.f;
    const int jc = J.w[level], jr = J.h[level];
    // the value reported when the loop exits: OpenCV stores nextPts = nextPt + halfWin each iteration
    float outx = nextx + half, outy = nexty + half;
    for (int j = 0; j < max_iters; ++j) {
      int inx = cv_floor(nextx), iny = cv_floor(nexty);
      if (inx < -win || inx >= jc || iny < -win || iny >= jr) {
        if (level == 0) status = 0;
        break;
      }
      if (iters_done) ++*iters_done;
      a = nextx - inx;
      b = nexty - iny;
      iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
      iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
      iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
      iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      int64_t ib1 = 0, ib2 = 0;
      for (int y = 0; y < win; ++y)
        for (int x = 0; x < win; ++x) {
          int X = inx + x, Y = iny + y;
          int diff = DESCALE(px(J, level, X, Y) * iw00 + px(J, level, X + 1, Y) * iw01 + px(J, level, X, Y + 1) * iw10 +
                                 px(J, level, X + 1, Y + 1) * iw11,
                             W_BITS - 5) -
                     Ibuf[y * win + x];
          ib1 += (int64_t)diff * dIbuf[(y * win + x) * 2];
          ib2 += (int64_t)diff * dIbuf[(y * win + x) * 2 + 1];
        }
      float b1 = (float)ib1 * FLT_SCALE, b2 = (float)ib2 * FLT_SCALE;
      float ddx = (A12 * b2 - A22 * b1) * D;
      float ddy = (A12 * b1 - A11 * b2) * D;
      nextx += ddx;
      nexty += ddy;
      outx = nextx + half;
      outy = nexty + half;
      if ((double)ddx * ddx + (double)ddy * ddy <= eps2) break;
      if (j > 0 && std::fabs(ddx + pdx) < 0.01 && std::fabs(ddy + pdy) < 0.01) {
        outx -= ddx * 0.5f;
        outy -= ddy * 0.5f;
        break;
      }
      pdx = ddx;
      pdy = ddy;
    }
    nextx = outx;
    nexty = outy;
  }
  nx = nextx;
  ny = nexty;
}

// ---------------------------------------------------------------- undistortPoints (radtan, 5 fixed-point iterations)
void undistort_radtan(const double *K8, float u, float v, float &xn, float &yn) {
  const double fx = K8[0], fy = K8[1], cx = K8[2], cy = K8[3], k1 = K8[4], k2 = K8[5], p1 = K8[6], p2 = K8[7];
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

// ---------------------------------------------------------------- RANSAC fundamental matrix
inline uint32_t hash32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
inline uint32_t rng_draw(uint32_t seed, uint32_t hyp, uint32_t t) { return hash32(seed ^ hash32(hyp * 0x9E3779B9U + hash32(t + 0x85EBCA6BU))); }

inline double det3(const double *m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// cos(acos(x) / 3) and the cube root from +, -, *, / and sqrt only (all correctly rounded on the host and on gfx950), so that the
// CPU restatement and the HIP kernel produce the same fundamental-matrix candidates bit for bit; libm's acos / cos / cbrt differ
// in the last place between the two.  cos(acos(x)/3) is the root of 4 t^3 - 3 t = x in [1/2, 1] (monotone there): 64 bisection
// steps.  Near x = -1 the root is double (t = 1/2) and only sqrt(eps) of it is determined — as ill-conditioned as the cubic's own
// roots are there.  The cube root: bit-level first guess, six Newton steps.
double det_cos_third(double x) {
  double lo = 0.5, hi = 1.0;
  for (int it = 0; it < 64; ++it) {
    const double mid = 0.5 * (lo + hi);
    const double f = ((4.0 * mid) * mid) * mid - 3.0 * mid - x;
    if (f < 0)
      lo = mid;
    else
      hi = mid;
  }
  return 0.5 * (lo + hi);
}
double det_cbrt(double x) {  // x >= 0
  if (!(x > 0)) return 0.0;
  unsigned long long i;
  memcpy(&i, &x, 8);
  i = i / 3 + 0x2A9F7893782DA1CEull;
  double y;
  memcpy(&y, &i, 8);
  for (int it = 0; it < 6; ++it) y = y - (y * y * y - x) / (3.0 * y * y);
  return y;
}

// real roots of c3 x^3 + c2 x^2 + c1 x + c0 (cv::solveCubic's case analysis, trigonometric form)
int solve_cubic(double c3, double c2, double c1, double c0, double *roots) {
  if (c3 == 0) {
    if (c2 == 0) {
      if (c1 == 0) return 0;
      roots[0] = -c0 / c1;
      return 1;
    }
    double d = c1 * c1 - 4 * c2 * c0;
    if (d < 0) return 0;
    d = std::sqrt(d);
    double q = 1. / (2 * c2);
    roots[0] = (-c1 - d) * q;
    roots[1] = (-c1 + d) * q;
    return d > 0 ? 2 : 1;
  }
  double a1 = c2 / c3, a2 = c1 / c3, a3 = c0 / c3;
  double Q = (a1 * a1 - 3 * a2) * (1. / 9);
  double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
  double Qcubed = Q * Q * Q;
  double d = Qcubed - R * R;
  if (d > 0) {
    // theta = acos(R / sqrt(Q^3)); the roots are -2 sqrt(Q) cos(theta / 3 + 2 pi k / 3) - a1 / 3
    double xr = R / std::sqrt(Qcubed);
    xr = xr < -1.0 ? -1.0 : (xr > 1.0 ? 1.0 : xr);
    const double ct = det_cos_third(xr), st = std::sqrt(1.0 - ct * ct);
    double sqrtQ = std::sqrt(Q);
    double t0 = -2 * sqrtQ, t2 = a1 * (1. / 3);
    roots[0] = t0 * ct - t2;
    roots[1] = t0 * (-0.5 * ct - 0.8660254037844386 * st) - t2;
    roots[2] = t0 * (-0.5 * ct + 0.8660254037844386 * st) - t2;
    return 3;
  } else if (d == 0) {
    if (R >= 0) {
      roots[0] = -2 * det_cbrt(R) - a1 / 3;
      roots[1] = det_cbrt(R) - a1 / 3;
    } else {
      roots[0] = 2 * det_cbrt(-R) - a1 / 3;
      roots[1] = -det_cbrt(-R) - a1 / 3;
    }
    return 2;
  } else {
    d = std::sqrt(-d);
    double e = det_cbrt(d + std::fabs(R));
    if (R > 0) e = -e;
    roots[0] = (e + Q / e) - a1 * (1. / 3);
    return 1;
  }
}

// Null space (2 vectors, each 9) of the 7x9 epipolar constraint matrix by Gauss-Jordan with full
// pivoting.  Returns false when the matrix has rank < 7.
bool nullspace_7x9(double A[7][9], double f1[9], double f2[9]) {
  int colperm[9];
  for (int j = 0; j < 9; ++j) colperm[j] = j;
  for (int i = 0; i < 7; ++i) {
    int pr = i, pc = i;
    double best = 0;
    for (int r = i; r < 7; ++r)
      for (int c = i; c < 9; ++c)
        if (std::fabs(A[r][c]) > best) best = std::fabs(A[r][c]), pr = r, pc = c;
    if (!(best > 1e-14)) return false;
    if (pr != i)
      for (int c = 0; c < 9; ++c) std::swap(A[pr][c], A[i][c]);
    if (pc != i) {
      for (int r = 0; r < 7; ++r) std::swap(A[r][pc], A[r][i]);
      std::swap(colperm[pc], colperm[i]);
    }
    double inv = 1.0 / A[i][i];
    for (int c = 0; c < 9; ++c) A[i][c] *= inv;
    for (int r = 0; r < 7; ++r) {
      if (r == i) continue;
      double f = A[r][i];
      if (f == 0) continue;
      for (int c = 0; c < 9; ++c) A[r][c] -= f * A[i][c];
    }
  }
  // free variables: permuted columns 7 and 8
  double v1[9], v2[9];
  for (int i = 0; i < 7; ++i) {
    v1[i] = -A[i][7];
    v2[i] = -A[i][8];
  }
  v1[7] = 1, v1[8] = 0, v2[7] = 0, v2[8] = 1;
  for (int j = 0; j < 9; ++j) {
    f1[colperm[j]] = v1[j];
    f2[colperm[j]] = v2[j];
  }
  return true;
}

// 7-point algorithm: up to 3 fundamental matrices (row-major 3x3 each, normalised like cv::run7Point)
int run7point(const float *m1, const float *m2, const int *idx, double *F) {
  double A[7][9];
  for (int i = 0; i < 7; ++i) {
    double x0 = m1[2 * idx[i]], y0 = m1[2 * idx[i] + 1], x1 = m2[2 * idx[i]], y1 = m2[2 * idx[i] + 1];
    A[i][0] = x1 * x0, A[i][1] = x1 * y0, A[i][2] = x1, A[i][3] = y1 * x0, A[i][4] = y1 * y0, A[i][5] = y1, A[i][6] = x0,
    A[i][7] = y0, A[i][8] = 1;
  }
  double f1[9], f2[9];
  if (!nullspace_7x9(A, f1, f2)) return 0;
  // det(lambda*f1 + (1-lambda)*f2) = 0  ->  with g = f1 - f2: det(f2 + lambda*g)
  double g[9];
  for (int i = 0; i < 9; ++i) g[i] = f1[i] - f2[i];
  double c0 = det3(f2), c3 = det3(g), c1 = 0, c2 = 0;
  for (int r = 0; r < 3; ++r) {
    double m[9], q[9];
    memcpy(m, f2, sizeof(m));
    memcpy(q, g, sizeof(q));
    for (int c = 0; c < 3; ++c) {
      m[3 * r + c] = g[3 * r + c];
      q[3 * r + c] = f2[3 * r + c];
    }
    c1 += det3(m);
    c2 += det3(q);
  }
  double roots[3];
  int n = solve_cubic(c3, c2, c1, c0, roots);
  int nout = 0;
  for (int k = 0; k < n; ++k) {
    double lambda = roots[k], mu = 1.;
    double s = g[8] * lambda + f2[8];
    double *Fk = F + 9 * nout;
    if (std::fabs(s) > DBL_EPSILON) {
      mu = 1. / s;
      lambda *= mu;
      Fk[8] = 1.;
    } else
      Fk[8] = 0.;
    for (int i = 0; i < 8; ++i) Fk[i] = g[i] * lambda + f2[i] * mu;
    bool finite = true;
    for (int i = 0; i < 9; ++i) finite = finite && std::isfinite(Fk[i]);
    if (finite) ++nout;
  }
  return nout;
}

inline float epi_err(const double *F, const float *m1, const float *m2, int i) {
  double x1 = m1[2 * i], y1 = m1[2 * i + 1], x2 = m2[2 * i], y2 = m2[2 * i + 1];
  double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
  double s2 = 1. / (a * a + b * b);
  double d2 = x2 * a + y2 * b + c;
  a = F[0] * x2 + F[3] * y2 + F[6];
  b = F[1] * x2 + F[4] * y2 + F[7];
  c = F[2] * x2 + F[5] * y2 + F[8];
  double s1 = 1. / (a * a + b * b);
  double d1 = x1 * a + y1 * b + c;
  return (float)std::max(d1 * d1 * s1, d2 * d2 * s2);
}

bool collinear_last(const float *m, const int *idx, int count) {
  int i = count - 1;
  for (int j = 0; j < i; ++j) {
    double dx1 = m[2 * idx[j]] - m[2 * idx[i]], dy1 = m[2 * idx[j] + 1] - m[2 * idx[i] + 1];
    for (int k = 0; k < j; ++k) {
      double dx2 = m[2 * idx[k]] - m[2 * idx[i]], dy2 = m[2 * idx[k] + 1] - m[2 * idx[i] + 1];
      if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2)))
        return true;
    }
  }
  return false;
}

// subset of hypothesis h: up to 16 attempts of 7 distinct draws + the collinearity check
bool get_subset(const float *m1, const float *m2, int n, uint32_t seed, uint32_t h, int *idx) {
  uint32_t t = 0;
  for (int attempt = 0; attempt < 16; ++attempt) {
    int i = 0;
    int guard = 0;
    while (i < 7 && guard < 64) {
      ++guard;
      int cand = (int)(((uint64_t)rng_draw(seed, h, t++) * (uint64_t)n) >> 32);
      bool dup = false;
      for (int j = 0; j < i; ++j) dup = dup || idx[j] == cand;
      if (dup) continue;
      idx[i++] = cand;
    }
    if (i < 7) continue;
    if (collinear_last(m1, idx, 7) || collinear_last(m2, idx, 7)) continue;
    return true;
  }
  return false;
}

int ransac_update_iters(double p, double ep, int model_points, int max_iters) {
  p = std::max(p, 0.);
  p = std::min(p, 1.);
  ep = std::max(ep, 0.);
  ep = std::min(ep, 1.);
  double num = std::max(1. - p, DBL_MIN);
  double denom = 1. - std::pow(1. - ep, model_points);
  if (denom < DBL_MIN) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : cv_round_d(num / denom);
}

// returns the number of inliers (0 => empty mask, as cv::findFundamentalMat on failure)
int ransac_fundamental(const float *m1, const float *m2, int n, double thr, double conf, int max_iters, uint32_t seed,
                       uint8_t *mask, int *iters_used) {
  memset(mask, 0, n);
  if (iters_used) *iters_used = 0;
  if (n < 7) return 0;
  const float t = (float)(thr * thr);
  int niters = max_iters, best = 0;
  double bestF[9] = {0};
  if (n == 7) niters = 1;
  int it = 0;
  for (; it < niters; ++it) {
    int idx[7];
    if (n == 7) {
      for (int i = 0; i < 7; ++i) idx[i] = i;
    } else if (!get_subset(m1, m2, n, seed, (uint32_t)it, idx)) {
      continue;  // this hypothesis yields no model
    }
    double F[27];
    int nm = run7point(m1, m2, idx, F);
    for (int k = 0; k < nm; ++k) {
      int good = 0;
      for (int i = 0; i < n; ++i) good += epi_err(F + 9 * k, m1, m2, i) <= t;
      if (good > std::max(best, 6)) {
        best = good;
        memcpy(bestF, F + 9 * k, sizeof(bestF));
        niters = ransac_update_iters(conf, (double)(n - good) / n, 7, niters);
      }
    }
  }
  if (iters_used) *iters_used = it;
  if (best <= 0) return 0;
  for (int i = 0; i < n; ++i) mask[i] = epi_err(bestF, m1, m2, i) <= t;
  return best;
}

}  // namespace

// ================================================================================ C API
extern "C" {

void orc_equalize_hist(const uint8_t *src, int w, int h, int stride, uint8_t *dst) { equalize_hist(src, w, h, stride, dst, w); }

// cv::createCLAHE(clip_limit, Size(tiles, tiles))->apply(src, dst) on 8-bit images.
// REF call sites: open_vins/ov_core/src/track/TrackKLT.cpp:60-64, PL-VIWO/src/update/cam/TrackLSD.cpp:84-88
// (clip 10.0, 8x8 tiles).  OpenCV's algorithm (imgproc/clahe.cpp): per-tile clipped histogram with the
// excess redistributed (batch + strided residual), cumulative LUT scaled by 255 / tile area, bilinear
// blend of the four surrounding tile LUTs in float, saturate_cast<uchar> (round to nearest even).
void orc_clahe(const uint8_t *src, int w, int h, double clip_limit, int tiles, uint8_t *dst) {
  const int tx = tiles, ty = tiles;
  int ew = w, eh = h;
  std::vector<uint8_t> ext;
  const uint8_t *lsrc = src;
  if (w % tx != 0 || h % ty != 0) {  // copyMakeBorder(.., 0, ty - h % ty, 0, tx - w % tx, BORDER_REFLECT_101)
    ew = w + (tx - (w % tx));
    eh = h + (ty - (h % ty));
    ext.resize((size_t)ew * eh);
    auto refl = [](int p, int n) {
      if (n == 1) return 0;
      while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - p - 2;
      return p;
    };
    for (int y = 0; y < eh; ++y)
      for (int x = 0; x < ew; ++x) ext[(size_t)y * ew + x] = src[(size_t)refl(y, h) * w + refl(x, w)];
    lsrc = ext.data();
  }
  const int tw = ew / tx, th = eh / ty, area = tw * th;
  const float lut_scale = (float)255 / area;
  int clip = 0;
  if (clip_limit > 0.0) {
    clip = (int)(clip_limit * area / 256);
    clip = std::max(clip, 1);
  }
  std::vector<uint8_t> lut((size_t)tx * ty * 256);
  for (int j = 0; j < ty; ++j)
    for (int i = 0; i < tx; ++i) {
      int hist[256] = {0};
      for (int y = 0; y < th; ++y)
        for (int x = 0; x < tw; ++x) hist[lsrc[(size_t)(j * th + y) * ew + i * tw + x]]++;
      if (clip > 0) {
        int clipped = 0;
        for (int b = 0; b < 256; ++b)
          if (hist[b] > clip) {
            clipped += hist[b] - clip;
            hist[b] = clip;
          }
        const int batch = clipped / 256;
        int residual = clipped - batch * 256;
        for (int b = 0; b < 256; ++b) hist[b] += batch;
        if (residual != 0) {
          const int step = std::max(256 / residual, 1);
          for (int b = 0; b < 256 && residual > 0; b += step, residual--) hist[b]++;
        }
      }
      int sum = 0;
      uint8_t *L = &lut[(size_t)(j * tx + i) * 256];
      for (int b = 0; b < 256; ++b) {
        sum += hist[b];
        const int v = (int)std::nearbyint((float)sum * lut_scale);
        L[b] = (uint8_t)std::min(std::max(v, 0), 255);
      }
    }
  const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
  for (int y = 0; y < h; ++y) {
    const float tyf = y * inv_th - 0.5f;
    int ty1 = (int)std::floor(tyf), ty2 = ty1 + 1;
    const float ya = tyf - ty1, ya1 = 1.0f - ya;
    ty1 = std::max(ty1, 0);
    ty2 = std::min(ty2, ty - 1);
    for (int x = 0; x < w; ++x) {
      const float txf = x * inv_tw - 0.5f;
      int tx1 = (int)std::floor(txf), tx2 = tx1 + 1;
      const float xa = txf - tx1, xa1 = 1.0f - xa;
      tx1 = std::max(tx1, 0);
      tx2 = std::min(tx2, tx - 1);
      const int v = src[(size_t)y * w + x];
      const float res = (lut[(size_t)(ty1 * tx + tx1) * 256 + v] * xa1 + lut[(size_t)(ty1 * tx + tx2) * 256 + v] * xa) * ya1 +
                        (lut[(size_t)(ty2 * tx + tx1) * 256 + v] * xa1 + lut[(size_t)(ty2 * tx + tx2) * 256 + v] * xa) * ya;
      const int r = (int)std::nearbyint(res);
      dst[(size_t)y * w + x] = (uint8_t)std::min(std::max(r, 0), 255);
    }
  }
}


// Pyramid handle
void *orc_pyramid_build(const uint8_t *img, int w, int h, int stride, int win, int max_level) {
  Pyramid *P = new Pyramid();
  build_pyramid(img, w, h, stride, win, max_level, true, *P);
  return P;
}
void orc_pyramid_free(void *p) { delete (Pyramid *)p; }
// cv::pyrDown with an explicit destination size (REF call site: UpdaterCamera.cpp:91,93, Size(cols / 2.0, rows / 2.0))
void orc_pyr_down(const uint8_t *src, int w, int h, uint8_t *dst, int dw, int dh) { pyr_down(src, w, h, dst, dw, dh); }
int orc_pyramid_levels(void *p) { return ((Pyramid *)p)->levels; }
void orc_pyramid_level(void *p, int l, int *w, int *h, uint8_t *img_out, int16_t *der_out) {
  Pyramid *P = (Pyramid *)p;
  *w = P->w[l];
  *h = P->h[l];
  if (img_out) memcpy(img_out, P->img[l].data(), P->img[l].size());
  if (der_out) memcpy(der_out, P->der[l].data(), P->der[l].size() * 2);
}

// calcOpticalFlowPyrLK(prev, cur, pts0, pts1 (initial flow, in/out), status) — nthreads like cv::parallel_for_
void orc_lk_track(void *prev, void *cur, int n, const float *pts0, float *pts1, uint8_t *status, int win, int max_iters,
                  float eps, int nthreads, long long *total_iters) {
  const Pyramid &I = *(Pyramid *)prev, &J = *(Pyramid *)cur;
  std::vector<int> it(n, 0);
  auto work = [&](int a, int b) {
    for (int i = a; i < b; ++i) lk_point(I, J, win, max_iters, eps, pts0[2 * i], pts0[2 * i + 1], pts1[2 * i], pts1[2 * i + 1], status[i], &it[i]);
  };
  if (nthreads <= 1) {
    work(0, n);
  } else {
    std::vector<std::thread> th;
    int per = (n + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; ++t) {
      int a = t * per, b = std::min(n, a + per);
      if (a < b) th.emplace_back(work, a, b);
    }
    for (auto &x : th) x.join();
  }
  if (total_iters) {
    long long s = 0;
    for (int v : it) s += v;
    *total_iters = s;
  }
}

void orc_undistort(const double *K8, int n, const float *uv, float *xy) {
  for (int i = 0; i < n; ++i) undistort_radtan(K8, uv[2 * i], uv[2 * i + 1], xy[2 * i], xy[2 * i + 1]);
}

int orc_ransac_fundamental(const float *m1, const float *m2, int n, double thr, double conf, int max_iters, uint32_t seed,
                           uint8_t *mask, int *iters_used) {
  return ransac_fundamental(m1, m2, n, thr, conf, max_iters, seed, mask, iters_used);
}

int orc_run7point(const float *m1, const float *m2, const int *idx, double *F) { return run7point(m1, m2, idx, F); }

// TrackKLT::perform_matching REF: TrackKLT.cpp:829-886.  pts1 holds the initial guess (== pts0 in
// the reference's monocular path) and receives the tracked positions; n1/n0 receive the
// normalised coordinates; mask_out = klt status & ransac inlier.  Returns 0, or 1 when n < 10
// (all-zero mask, REF :848-852).
int orc_perform_matching(void *prev, void *cur, int n, const float *pts0, float *pts1, const double *K8, int win,
                         int max_iters, float eps, double ransac_thr_px, double conf, int ransac_iters, uint32_t seed,
                         uint8_t *mask_out, float *n0, float *n1, int nthreads) {
  memset(mask_out, 0, n);
  if (n < 10) return 1;
  std::vector<uint8_t> st(n), rs(n);
  orc_lk_track(prev, cur, n, pts0, pts1, st.data(), win, max_iters, eps, nthreads, nullptr);
  orc_undistort(K8, n, pts0, n0);
  orc_undistort(K8, n, pts1, n1);
  double fmax = std::max(K8[0], K8[1]);
  ransac_fundamental(n0, n1, n, ransac_thr_px / fmax, conf, ransac_iters, seed, rs.data(), nullptr);
  for (int i = 0; i < n; ++i) mask_out[i] = (st[i] && rs[i]) ? 1 : 0;
  return 0;
}

}  // extern "C"
