// ORACLE — TEST INFRASTRUCTURE ONLY (see dense.h).  PARITY UNPINNED.
//
// CPU restatement of the detection half of the point front-end:
//   cv::FAST(roi, kps, threshold, true)   REF call site: open_vins/ov_core/src/track/Grider_GRID.h:125
//   Grider_GRID::perform_griding          REF: Grider_GRID.h:74-180
//   cv::cornerSubPix(5x5, (-1,-1), 20|1e-3) REF call site: Grider_GRID.h:163-174
//   TrackKLT::perform_detection_monocular REF: open_vins/ov_core/src/track/TrackKLT.cpp:395-528
// OpenCV is un-vendored: FAST-9/16 (Rosten), its corner score (largest threshold that keeps the
// corner), 3x3 strict non-max suppression and the Foerstner sub-pixel iteration follow the
// published algorithms / SURVEY Appendix A.  std::sort with a response-only comparator leaves
// ties in unspecified order in the reference; here ties are broken by raster order (y, then x).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

const int OFFS[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                         {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// score = largest threshold t for which >= 9 contiguous circle pixels are all brighter than v+t or
// all darker than v-t, minus 1; 0 when the pixel is not a corner at `threshold`.
int fast_score(const uint8_t *img, int stride, int x, int y, int threshold) {
  int d[25];
  const int v = img[y * stride + x];
  for (int k = 0; k < 16; ++k) d[k] = v - img[(y + OFFS[k][1]) * stride + x + OFFS[k][0]];
  for (int k = 16; k < 25; ++k) d[k] = d[k - 16];
  int best_dark = INT32_MIN, best_bright = INT32_MIN;  // max over arcs of min(d) / min(-d)
  for (int k = 0; k < 16; ++k) {
    int mn = d[k], mx = d[k];
    for (int q = 1; q < 9; ++q) {
      mn = std::min(mn, d[k + q]);
      mx = std::max(mx, d[k + q]);
    }
    best_dark = std::max(best_dark, mn);
    best_bright = std::max(best_bright, -mx);
  }
  const int a = std::max(best_dark, best_bright);
  if (a <= threshold) return 0;
  return a - 1;
}

struct Kp {
  float x, y, response;
};

// cv::FAST with NMS on a ROI (x0,y0,w,h) of the image; keypoints in ROI coordinates, raster order
void fast_roi(const uint8_t *img, int stride, int x0, int y0, int w, int h, int threshold, std::vector<Kp> &out) {
  std::vector<int> sc((size_t)w * h, 0);
  const uint8_t *roi = img + (size_t)y0 * stride + x0;
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x) sc[(size_t)y * w + x] = fast_score(roi, stride, x, y, threshold);
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x) {
      const int s = sc[(size_t)y * w + x];
      if (s == 0) continue;
      bool mx = true;
      for (int dy = -1; dy <= 1 && mx; ++dy)
        for (int dx = -1; dx <= 1; ++dx)
          if ((dx || dy) && sc[(size_t)(y + dy) * w + x + dx] >= s) {
            mx = false;
            break;
          }
      if (mx) out.push_back(Kp{(float)x, (float)y, (float)s});
    }
}

inline float px_replicate(const uint8_t *img, int w, int h, int x, int y) {
  x = x < 0 ? 0 : (x >= w ? w - 1 : x);
  y = y < 0 ? 0 : (y >= h ? h - 1 : y);
  return (float)img[(size_t)y * w + x];
}

// cv::cornerSubPix for one point (win 5x5 -> 11x11, no zero zone)
void corner_subpix(const uint8_t *img, int w, int h, const float *mask, int win, int max_iters, double eps, float &px, float &py) {
  const int ww = 2 * win + 1, sw = ww + 2;
  std::vector<float> sub((size_t)sw * sw);
  const float cTx = px, cTy = py;
  float cIx = px, cIy = py;
  const double eps2 = eps * eps;
  int iter = 0;
  double err = 0;
  do {
    // getRectSubPix: bilinear sample of a (ww+2)^2 patch centred on cI, replicated border
    const float cx = cIx - (sw - 1) * 0.5f, cy = cIy - (sw - 1) * 0.5f;
    const int ix = (int)floorf(cx), iy = (int)floorf(cy);
    const float a = cx - ix, b = cy - iy;
    const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
    for (int i = 0; i < sw; ++i)
      for (int j = 0; j < sw; ++j)
        sub[(size_t)i * sw + j] = px_replicate(img, w, h, ix + j, iy + i) * a11 + px_replicate(img, w, h, ix + j + 1, iy + i) * a12 +
                                  px_replicate(img, w, h, ix + j, iy + i + 1) * a21 + px_replicate(img, w, h, ix + j + 1, iy + i + 1) * a22;
    double A = 0, B = 0, Cc = 0, bb1 = 0, bb2 = 0;
    for (int i = 0; i < ww; ++i) {
      const double pyy = i - win;
      for (int j = 0; j < ww; ++j) {
        const float m = mask[i * ww + j];
        const float *s = &sub[(size_t)(i + 1) * sw + j + 1];
        const float tgx = s[1] - s[-1], tgy = s[sw] - s[-sw];
        const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
        const double pxx = j - win;
        A += gxx;
        B += gxy;
        Cc += gyy;
        bb1 += gxx * pxx + gxy * pyy;
        bb2 += gxy * pxx + gyy * pyy;
      }
    }
    const double det = A * Cc - B * B;
    if (std::fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
    const double scale = 1.0 / det;
    const float nx = (float)(cIx + Cc * scale * bb1 - B * scale * bb2);
    const float ny = (float)(cIy - B * scale * bb1 + A * scale * bb2);
    err = (double)(nx - cIx) * (nx - cIx) + (double)(ny - cIy) * (ny - cIy);
    cIx = nx;
    cIy = ny;
    if (cIx < 0 || cIx >= w || cIy < 0 || cIy >= h) break;
  } while (++iter < max_iters && err > eps2);
  if (std::fabs(cIx - cTx) > win || std::fabs(cIy - cTy) > win) {
    cIx = cTx;
    cIy = cTy;
  }
  px = cIx;
  py = cIy;
}

void subpix_mask(int win, std::vector<float> &mask) {
  const int ww = 2 * win + 1;
  mask.resize((size_t)ww * ww);
  for (int i = 0; i < ww; ++i) {
    float y = (float)(i - win) / win;
    float vy = std::exp(-y * y);
    for (int j = 0; j < ww; ++j) {
      float x = (float)(j - win) / win;
      mask[i * ww + j] = (float)(vy * std::exp(-x * x));
    }
  }
}

}  // namespace

extern "C" {

int orc_fast_score(const uint8_t *img, int stride, int x, int y, int threshold) { return fast_score(img, stride, x, y, threshold); }

// cv::FAST(img(roi), ..., threshold, true): returns count, fills xy (ROI coords) / response
int orc_fast_roi(const uint8_t *img, int stride, int x0, int y0, int w, int h, int threshold, int cap, float *xy, float *resp) {
  std::vector<Kp> k;
  fast_roi(img, stride, x0, y0, w, h, threshold, k);
  int n = std::min((int)k.size(), cap);
  for (int i = 0; i < n; ++i) {
    xy[2 * i] = k[i].x;
    xy[2 * i + 1] = k[i].y;
    resp[i] = k[i].response;
  }
  return (int)k.size();
}

void orc_corner_subpix(const uint8_t *img, int w, int h, int n, float *xy, int win, int max_iters, double eps) {
  std::vector<float> mask;
  subpix_mask(win, mask);
  for (int i = 0; i < n; ++i) corner_subpix(img, w, h, mask.data(), win, max_iters, eps, xy[2 * i], xy[2 * i + 1]);
}

// TrackKLT::perform_detection_monocular (REF: TrackKLT.cpp:395-528) on the level-0 image `img`
// (w x h packed) and `mask` (w x h, 255 = masked).  pts/ids hold n_in tracked points on entry and
// are compacted/extended in place (capacity cap); *currid is the tracker's id counter.
// Returns the new count.
int orc_perform_detection(const uint8_t *img, const uint8_t *mask, int w, int h, int num_features, int grid_x, int grid_y,
                          int min_px_dist, int threshold, float *pts, uint64_t *ids, int n_in, int cap, uint64_t *currid) {
  const int cw = (int)((float)w / (float)min_px_dist), ch = (int)((float)h / (float)min_px_dist);
  std::vector<uint8_t> close((size_t)cw * ch, 0), grid((size_t)grid_x * grid_y, 0);
  const float size_x = (float)w / (float)grid_x, size_y = (float)h / (float)grid_y;
  std::vector<uint8_t> mask_upd(mask, mask + (size_t)w * h);
  int n = 0;
  for (int i = 0; i < n_in; ++i) {
    const float fx = pts[2 * i], fy = pts[2 * i + 1];
    const int x = (int)fx, y = (int)fy;
    const int edge = 10;
    if (x < edge || x >= w - edge || y < edge || y >= h - edge) continue;
    const int xc = (int)(fx / (float)min_px_dist), yc = (int)(fy / (float)min_px_dist);
    if (xc < 0 || xc >= cw || yc < 0 || yc >= ch) continue;
    const int xg = (int)std::floor(fx / size_x), yg = (int)std::floor(fy / size_y);
    if (xg < 0 || xg >= grid_x || yg < 0 || yg >= grid_y) continue;
    if (close[(size_t)yc * cw + xc] > 127) continue;
    if (mask[(size_t)y * w + x] > 127) continue;
    close[(size_t)yc * cw + xc] = 255;
    if (grid[(size_t)yg * grid_x + xg] < 255) grid[(size_t)yg * grid_x + xg] += 1;
    if (x - min_px_dist >= 0 && x + min_px_dist < w && y - min_px_dist >= 0 && y + min_px_dist < h)
      for (int yy = y - min_px_dist; yy <= y + min_px_dist; ++yy) memset(&mask_upd[(size_t)yy * w + x - min_px_dist], 255, 2 * min_px_dist + 1);
    pts[2 * n] = fx;
    pts[2 * n + 1] = fy;
    ids[n] = ids[i];
    ++n;
  }
  const double min_feat_percent = 0.50;
  const int needed = num_features - n;
  if (needed < std::min(20, (int)(min_feat_percent * num_features))) return n;
  // mask resized NEAREST to the grid: src = floor(dst * scale)
  const int nfg_req = std::max(1, (int)(min_feat_percent * ((int)((double)num_features / (double)(grid_x * grid_y)) + 1)));
  std::vector<std::pair<int, int>> valid;
  for (int x = 0; x < grid_x; ++x)
    for (int y = 0; y < grid_y; ++y) {
      int sx = std::min((int)std::floor(x * (double)w / grid_x), w - 1), sy = std::min((int)std::floor(y * (double)h / grid_y), h - 1);
      if ((int)grid[(size_t)y * grid_x + x] < nfg_req && (int)mask[(size_t)sy * w + sx] != 255) valid.emplace_back(x, y);
    }
  // Grider_GRID::perform_griding
  std::vector<Kp> ext;
  if (!valid.empty()) {
    int gx = grid_x, gy = grid_y;
    if (num_features < gx * gy) {
      double ratio = (double)gx / (double)gy;
      gy = (int)std::ceil(std::sqrt(num_features / ratio));
      gx = (int)std::ceil(gy * ratio);
    }
    const int nfg = (int)((double)num_features / (double)(gx * gy)) + 1;
    const int sxp = w / gx, syp = h / gy;
    for (auto &g : valid) {
      const int x = g.first * sxp, y = g.second * syp;
      if (x + sxp > w || y + syp > h) continue;
      std::vector<Kp> k;
      fast_roi(img, w, x, y, sxp, syp, threshold, k);
      std::stable_sort(k.begin(), k.end(), [](const Kp &a, const Kp &b) { return a.response > b.response; });
      for (size_t i = 0; i < (size_t)nfg && i < k.size(); ++i) {
        Kp p = k[i];
        p.x += (float)x;
        p.y += (float)y;
        if ((int)p.x < 0 || (int)p.x > w || (int)p.y < 0 || (int)p.y > h) continue;
        if (mask_upd[(size_t)(int)p.y * w + (int)p.x] > 127) continue;
        ext.push_back(p);
      }
    }
    if (!ext.empty()) {
      std::vector<float> mk;
      subpix_mask(5, mk);
      for (auto &p : ext) corner_subpix(img, w, h, mk.data(), 5, 20, 0.001, p.x, p.y);
    }
  }
  for (auto &p : ext) {
    const int xg = (int)(p.x / (float)min_px_dist), yg = (int)(p.y / (float)min_px_dist);
    if (xg < 0 || xg >= cw || yg < 0 || yg >= ch) continue;
    if (close[(size_t)yg * cw + xg] > 127) continue;
    if (n >= cap) break;
    close[(size_t)yg * cw + xg] = 255;
    pts[2 * n] = p.x;
    pts[2 * n + 1] = p.y;
    ids[n] = ++*currid;
    ++n;
  }
  return n;
}

}  // extern "C"
