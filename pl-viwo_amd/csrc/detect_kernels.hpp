// detect_kernels.hpp — parameters of fast_cells_kernel / subpix_kernel.
#pragma once
#include "plv_ctx.hpp"

namespace plv {

struct DetectParams {
  const uint8_t *img;   // level-0 (equalised) image, packed
  int W, H;
  const uint8_t *mask;  // optional W x H, > 127 = masked
  const int *cells;     // [n_cells][2] grid coordinates of the cells to extract from
  int cell_w, cell_h;
  int threshold, nfg, cand_cap;
  const int *boxes;     // [n_boxes][2] integer positions of the tracked points whose +-min_px_dist box is masked
  int n_boxes, min_px_dist;
  float *out_xy;        // [n_cells * nfg][2]
  float *out_resp;      // [n_cells * nfg]
  uint8_t *out_valid;   // [n_cells * nfg]
};

int launch_fast_cells(plv_ctx *ctx, const DetectParams &P, int n_cells, unsigned long long *d_cand, int *d_cand_n);
int launch_subpix(plv_ctx *ctx, const uint8_t *d_img, int W, int H, int n, const uint8_t *d_valid, float *d_xy,
                  const float *d_mask, int win, int max_iters, double eps);

}  // namespace plv
