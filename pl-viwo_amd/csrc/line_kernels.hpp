// line_kernels.hpp — device buffers and launcher of the line detector (line_kernels.hip).
#pragma once
#include <cstdint>

#include "plv_ctx.hpp"

namespace plv {

struct FldChain {
  int start, len, slot;  // points [start, start+len) of the chain buffer; first output slot
};
struct FldParams {
  int length_threshold;      // 20   REF: TrackLSD.h:269
  float distance_threshold;  // sqrt(2)   :270
  int canny_low, canny_high; // 50, 50   :271-272 (aperture 3 is what canny_kernel implements)
};
struct FldBuffers {
  uint8_t *half, *map, *work;  // half-resolution image, Canny map (0 weak / 1 none / 2 edge), walk scratch
  int2 *pts;
  FldChain *chains;
  int chain_cap;
  int *counts;     // [0] chains, [1] segment slots, [2] points
  float4 *segs;    // segment slots, chain c writes segs[chain.slot ...]
  int *seg_count;  // [chain_cap]
  int *lab_work = nullptr;      // [w*h] union-find forest of the component labelling (null: no labels for this detection)
  int *lab_cnt = nullptr;       // [w*h] pixels per component at its root, then -(part) there
  int *lab_roots = nullptr;     // [line_label_roots_bytes()] the roots' list and its counter
  uint8_t *lab_out = nullptr;   // [w*h] parts, host-visible (0 not an edge, else 1 .. line_label_parts())
  // (nullable, host-visible) the parts' pixels in raster order, per run of 256 pixels: see ccl_flatten_kernel
  uint8_t *blk_sorted = nullptr;        // [256 * ceil(w*h / 256)]
  unsigned short *blk_bins = nullptr;   // [ceil(w*h / 256)][line_label_parts() + 1]
};

// d_hist: d_img is the RAW image and d_hist its histogram (the kernel equalises on the fly: canny_kernel); null: d_img is the equalised image
int launch_line_edges(plv_ctx *ctx, const uint8_t *d_img, int W, int H, const FldParams &fp, FldBuffers &b, hipStream_t st = nullptr /* default: the ctx stream */,
                      const unsigned *d_hist = nullptr);
int launch_line_labels(plv_ctx *ctx, int w, int h, FldBuffers &b, hipStream_t st, int length_threshold);
int line_label_parts();         // parts launch_line_labels writes (the largest components first: 1 .. 8, then 8 hashed ones)
size_t line_label_roots_bytes();
int launch_line_walk(plv_ctx *ctx, int w, int h, const FldParams &fp, FldBuffers &b);
int launch_line_fit(plv_ctx *ctx, int w, int h, const FldParams &fp, FldBuffers &b);

// frontend_api.hip: equalised level-0 image of the current (which = 0) or previous (1) frame
const uint8_t *plv_front_level0(plv_ctx *ctx, int which, int *w, int *h);
int plv_front_fed_count(plv_ctx *ctx);  // images fed so far: identifies the frame a cached detection belongs to

}  // namespace plv
