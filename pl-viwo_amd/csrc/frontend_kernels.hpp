// frontend_kernels.hpp — device-side descriptors and launcher declarations of frontend_kernels.hip.
#pragma once
#include "plv_ctx.hpp"

namespace plv {

#define PLV_MAX_LEVELS 8

// One image pyramid in a single device allocation (levels packed, row stride = level width).
struct PyrDesc {
  uint8_t *base;
  int levels;
  int w[PLV_MAX_LEVELS], h[PLV_MAX_LEVELS];
  unsigned off[PLV_MAX_LEVELS];
};

struct CamK {
  double v[8];
};

int launch_equalize(plv_ctx *ctx, const uint8_t *d_src, uint8_t *d_dst, int npix, unsigned *d_hist);
int launch_clahe(plv_ctx *ctx, const uint8_t *d_src, uint8_t *d_dst, int w, int h, double clip_limit, int tiles, uint8_t *d_lut);
int launch_pyramid(plv_ctx *ctx, const PyrDesc &p, int first_level = 0, unsigned *clear_hist = nullptr);
// h_src != null: the raw image is still in pinned host memory; the histogram kernel reads it from there and writes d_raw (no copy command)
int launch_equalize_pyramid(plv_ctx *ctx, const uint8_t *d_raw, const PyrDesc &p, unsigned *d_hist, const uint8_t *h_src = nullptr);
int launch_pyrdown(plv_ctx *ctx, const uint8_t *d_src, int sw, int sh, uint8_t *d_dst, int dw, int dh);
int launch_lk(plv_ctx *ctx, const PyrDesc &prev, const PyrDesc &cur, int n, const float *d_pts0, float *d_pts1,
              uint8_t *d_status, int *d_iters, int win, int max_iters, float eps,
              const CamK *K = nullptr, float *d_n0 = nullptr, float *d_n1 = nullptr,
              const float *pts1_init = nullptr /* initial guesses when they are not in d_pts1 (e.g. pinned host memory) */);
int launch_undistort(plv_ctx *ctx, const CamK &K, int n, const float *d_uv, float *d_xy);
int launch_undistort2(plv_ctx *ctx, const CamK &K, int n, const float *d_uv0, const float *d_uv1, float *d_xy0,
                      float *d_xy1);
int launch_ransac(plv_ctx *ctx, const float *d_m1, const float *d_m2, int n, double thr, double conf, int max_iters,
                  unsigned seed, int *d_counts, const uint8_t *d_klt, uint8_t *d_mask, int *d_info, double *d_models = nullptr,
                  // optional: the selection kernel copies mir_bytes from mir_src to mir_dst (pinned host) and its mask to mir_mask;
                  // *mirrored tells whether a kernel that does it was launched (not for n < 7)
                  const void *mir_src = nullptr, void *mir_dst = nullptr, size_t mir_bytes = 0, uint8_t *mir_mask = nullptr,
                  bool *mirrored = nullptr,
                  // optional (with mir_dst): pinned word the selection kernel stores done_val to behind everything it wrote
                  unsigned *done_word = nullptr, unsigned done_val = 0);

}  // namespace plv
