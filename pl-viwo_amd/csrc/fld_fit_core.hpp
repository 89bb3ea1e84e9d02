// fld_fit_core.hpp — FastLineDetector's segment growing for ONE edge chain, shared by fld_fit_kernel (one lane per chain,
// device-walk mode) and the host path of plv_detect_lines (default: the chain walk and the segment growth are one dependent
// scalar sequence per chain; a host core retires it ~25x faster than a single GPU lane, DESIGN.md "Line detector").
//   REF (OpenCV contract): ximgproc/src/fast_line_detector.cpp lineDetection / extractSegments /
//   additionalOperationsOnSegment; cv::fitLine(DIST_L2).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#define PLV_HD __host__ __device__

namespace plv {

PLV_HD inline int plv_f2i_rn(float x) {   // Point2i(Point2f): saturate_cast = round to nearest even
#ifdef __HIP_DEVICE_COMPILE__
  return __float2int_rn(x);
#else
  return (int)lrintf(x);
#endif
}
PLV_HD inline int plv_d2i_rn(double x) {
#ifdef __HIP_DEVICE_COMPILE__
  return __double2int_rn(x);
#else
  return (int)lrint(x);
#endif
}
PLV_HD inline float4 plv_make_float4(float a, float b, float c, float d) {
  float4 r;
  r.x = a, r.y = b, r.z = c, r.w = d;
  return r;
}

struct L3 {
  double a, b, c;
};
PLV_HD inline L3 cr3(double ax, double ay, double az, double bx, double by, double bz) {
  return L3{ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx};
}
PLV_HD inline double dist_pl(double px, double py, L3 &l) {  // normalises l in place (as the reference does)
  // (a line that went through here before has a^2 + b^2 within an ulp of 1: for 1 and for 1 + 2^-52 the correctly rounded square
  // root is exactly 1 and the three divisions change nothing — the common case skips them, ~40 cycles of the ~80 a chain point
  // costs; 1 - 2^-53 has the root 1 - 2^-53 and takes the divisions like any other value)
  const double s2 = l.a * l.a + l.b * l.b;
  if (!(s2 == 1.0 || s2 == 1.0 + 2.220446049250313e-16)) {
    const double wv = sqrt(s2);
    l.a /= wv;
    l.b /= wv;
    l.c /= wv;
  }
  return l.a * px + l.b * py + l.c;
}
struct Fit {
  long long n, sx, sy, sxx, syy, sxy;
  PLV_HD void add(int2 p) {
    ++n;
    sx += p.x;
    sy += p.y;
    sxx += (long long)p.x * p.x;
    syy += (long long)p.y * p.y;
    sxy += (long long)p.x * p.y;
  }
  // cv::fitLine(DIST_L2) = principal axis through the centroid; integer sums are exact in double
  PLV_HD L3 line() const {
    const double wv = (double)(float)n;
    const double x = (double)sx / wv, y = (double)sy / wv, x2 = (double)sxx / wv, y2 = (double)syy / wv, xy = (double)sxy / wv;
    const double dx2 = x2 - x * x, dy2 = y2 - y * y, dxy = xy - x * y;
    const float t = (float)atan2(2 * dxy, dx2 - dy2) / 2;
    const float vx = (float)cos((double)t), vy = (float)sin((double)t), fx = (float)x, fy = (float)y;
    return cr3((double)fx, (double)fy, 1.0, (double)fx + (double)vx, (double)fy + (double)vy, 1.0);
  }
};
PLV_HD inline void incident(const L3 &l, float &px, float &py, int imw, int imh) {
  const L3 lk = cr3((double)px, (double)py, 1.0, l.a, l.b, 0.0);
  L3 xk = cr3(lk.a, lk.b, lk.c, l.a, l.b, l.c);
  const double s = 1.0 / xk.c;
  const float fx = (float)(xk.a * s), fy = (float)(xk.b * s);
  px = fx < 0.0f ? 0.0f : (fx >= (imw - 1.0f) ? (imw - 1.0f) : fx);
  py = fy < 0.0f ? 0.0f : (fy >= (imh - 1.0f) ? (imh - 1.0f) : fy);
}
PLV_HD inline void inboard(int &x, int &y, int w, int h) {
  x = x <= 5 ? 5 : (x >= w - 5 ? w - 6 : x);
  y = y <= 5 ? 5 : (y >= h - 5 ? h - 6 : y);
}


// Segments of one chain P[0..total): writes them to out (capacity total / length_threshold + 1) and returns their number.
PLV_HD inline int fit_chain(const uint8_t *src, int w, int h, int length_threshold, float distance_threshold, const int2 *P, int total,
                            float4 *out) {
  int nseg = 0;
  for (int i = 0; i + length_threshold < total; ++i) {
    int2 ps = P[i], pe = P[i + length_threshold];
    L3 l = cr3(ps.x, ps.y, 1.0, pe.x, pe.y, 1.0);
    bool is_line = true;
    // Is every point between the two within distance_threshold of the chord?  The reference normalises the chord (a square root and
    // three divisions) and compares |a' x + b' y + c'| with the threshold; most starts of a chain fail this test after a few points
    // (round 6: 27 ns per chain point, two thirds of the detector's host time).  The chord through two pixels has integer
    // coefficients, so v = a x + b y + c is exact and |v| / sqrt(a^2 + b^2) is the distance the reference rounds: when v^2 lies
    // clear of threshold^2 (a^2 + b^2) — by 1e-9, the reference's own rounding is below 1e-12 — the verdict is taken from the
    // integers; a value inside that band (a tie to nine digits) is decided by the reference's arithmetic.  Same verdicts, bit for bit.
    {
      const double s2 = l.a * l.a + l.b * l.b, thr = (double)distance_threshold;
      const double lim = thr * thr * s2, lim_hi = lim * (1.0 + 1e-9), lim_lo = lim * (1.0 - 1e-9);
      L3 ln = l;
      bool normalised = false;
      for (int j = 1; j < length_threshold; ++j) {
        const int2 pt = P[i + j];
        const double v = l.a * pt.x + l.b * pt.y + l.c, v2 = v * v;
        if (v2 < lim_lo) continue;
        if (v2 > lim_hi) {
          is_line = false;
          break;
        }
        if (!normalised) {  // (the reference's own sequence: dist_pl normalises the chord in place at its first call)
          const double wv = sqrt(s2);
          ln.a /= wv, ln.b /= wv, ln.c /= wv;
          normalised = true;
        }
        if (fabs(ln.a * pt.x + ln.b * pt.y + ln.c) > distance_threshold) {
          is_line = false;
          break;
        }
      }
    }
    if (!is_line) continue;
    // (the sums of the fit are exact integers: adding the window's points once the test has passed gives the values the reference
    //  accumulates while it tests)
    Fit fs{0, 0, 0, 0, 0, 0};
    for (int j = 0; j <= length_threshold; ++j) fs.add(P[i + j]);
    l = fs.line();
    {
      float fx = (float)ps.x, fy = (float)ps.y;
      incident(l, fx, fy, w, h);
      ps.x = plv_f2i_rn(fx);  // Point2i(Point2f): saturate_cast = round to nearest even
      ps.y = plv_f2i_rn(fy);
    }
    int j;
    for (j = length_threshold + 1; i + j < total; ++j) {
      const int2 pt = P[i + j];
      double dist = dist_pl(pt.x, pt.y, l);
      if (fabs(dist) > distance_threshold) {
        l = fs.line();
        dist = dist_pl(pt.x, pt.y, l);
        if (fabs(dist) > distance_threshold) {
          j--;
          break;
        }
      }
      pe = pt;
      fs.add(pt);
    }
    l = fs.line();
    float e1x = (float)ps.x, e1y = (float)ps.y, e2x = (float)pe.x, e2y = (float)pe.y;
    incident(l, e1x, e1y, w, h);
    incident(l, e2x, e2y, w, h);
    i = i + j;
    // lineDetection's filters, then additionalOperationsOnSegment (orientation by side brightness)
    const float length = sqrtf((e1x - e2x) * (e1x - e2x) + (e1y - e2y) * (e1y - e2y));
    if (length < length_threshold) continue;
    if ((e1x <= 5.0f && e2x <= 5.0f) || (e1y <= 5.0f && e2y <= 5.0f) || (e1x >= w - 5.0f && e2x >= w - 5.0f) ||
        (e1y >= h - 5.0f && e2y >= h - 5.0f))
      continue;
    if (!(e1x == 0.0f && e2x == 0.0f && e1y == 0.0f && e2y == 0.0f)) {
      const double ang = (double)atan2f(e2y - e1y, e2x - e1x);
      const double dx = (double)e2x - (double)e1x, dy = (double)e2y - (double)e1y;
      const double ca = cos(90.0 * 3.14159265358979323846 / 180.0 + ang), sa = sin(90.0 * 3.14159265358979323846 / 180.0 + ang);
      int iR = 0, iL = 0;
      for (int q = 0; q < 10; ++q) {
        float qx, qy;
        if (q == 0) {
          qx = e1x;
          qy = e1y;
        } else if (q == 9) {
          qx = e2x;
          qy = e2y;
        } else {
          qx = e1x + ((float)dx / 9.0f * (float)q);
          qy = e1y + ((float)dy / 9.0f * (float)q);
        }
        int rx = plv_d2i_rn((double)qx + ca), ry = plv_d2i_rn((double)qy + sa);
        int lx = plv_d2i_rn((double)qx - ca), ly = plv_d2i_rn((double)qy - sa);
        inboard(rx, ry, w, h);
        inboard(lx, ly, w, h);
        iR += src[(size_t)ry * w + rx];
        iL += src[(size_t)ly * w + lx];
      }
      if (iR > iL) {
        float t = e1x;
        e1x = e2x;
        e2x = t;
        t = e1y;
        e1y = e2y;
        e2y = t;
      }
    }
    out[nseg++] = plv_make_float4(e1x, e1y, e2x, e2y);
  }
  return nseg;
}


}  // namespace plv
