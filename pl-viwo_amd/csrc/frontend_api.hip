// frontend_api.hip — point front-end entry points of include/plviwo.h (product code, no CPU path).
#include <algorithm>
#include <cmath>
#include <vector>

#include "detect_kernels.hpp"
#include "frontend_kernels.hpp"
#include "line_kernels.hpp"

using namespace plv;

namespace {

struct DetJob {  // one top-up detection between its host stages (plv_perform_detection / _ahead)
  std::vector<uint8_t> close;        // occupancy grid of the kept points (min_px_dist cells)
  std::vector<int> boxes, cells;
  std::vector<float> pts;            // kept points
  std::vector<uint64_t> ids;
  int cw = 0, ch = 0, nfg = 0, sxp = 0, syp = 0, n_cells = 0, n_slots = 0;
  char *h_out = nullptr;             // pinned: [xy n_slots x 2 f32][resp n_slots f32][valid n_slots u8]
  // what an ahead-of-time job was started from
  bool active = false, has_mask = false;
  int fed = 0, n_in = 0;
  std::vector<float> in_pts;
  std::vector<uint64_t> in_ids;
  std::vector<uint8_t> in_mask;
};

struct FrontState {
  int W = 0, H = 0;
  PyrDesc pyr[2];          // ping-pong pyramids
  DevBuf pyr_mem[2];
  int cur = 0;             // index of the current pyramid; last = 1 - cur
  int fed = 0;             // number of images fed so far (last is valid when fed >= 2)
  int pending_n = -1;      // plv_perform_matching_launch .. _wait
  bool pending_ran = false;
  DevBuf raw;              // incoming raw image (packed)
  DevBuf slots[8];
  DevBuf hist, clahe_lut;
  DevBuf ds_src, ds_dst;   // full-resolution staging of plv_downsample / plv_feed_image_downsampled
  // per-call point buffers
  DevBuf pts0, pts1, n0, n1, status, iters, mask, counts, info, io, models;
  DevBuf det_in, det_out, det_mask, subpix_tab, det_cand, det_cand_n;  // detection staging
  PinBuf det_pin;
  PinBuf img_pin[6];       // host images on their way to the device (plv_feed_image_enqueue): the caller's buffer is free at return
  int img_pin_next = 0;
  DetJob det_pending;
  hipStream_t det_stream = nullptr;
  hipEvent_t det_done = nullptr;
  unsigned long long match_done_stamp = 0;  // plv_ctx::gather_stamp when match_done was recorded
  unsigned match_word_seq = 0;              // nonzero: the flow's last kernel stores this number to plv_ctx::done_word(0)
  hipEvent_t match_done = nullptr;  // behind the result copy of plv_perform_matching_launch: the wait does not cover what is enqueued after it
};

#define TRY(expr)                  \
  do {                             \
    int _rc = (expr);              \
    if (_rc != PLV_OK) return _rc; \
  } while (0)

int level_geometry(int W, int H, int win, int max_level, PyrDesc &p) {
  p.levels = 0;
  unsigned off = 0;
  int w = W, h = H;
  for (int l = 0; l <= max_level && l < PLV_MAX_LEVELS; ++l) {
    if (l > 0) {
      int nw = (w + 1) / 2, nh = (h + 1) / 2;
      if (nw <= win || nh <= win) break;  // cv::buildOpticalFlowPyramid stops here
      w = nw;
      h = nh;
    }
    p.w[l] = w;
    p.h[l] = h;
    p.off[l] = off;
    off += (unsigned)(((size_t)w * h + 255) & ~(size_t)255);
    p.levels = l + 1;
  }
  return (int)off;
}

}  // namespace
extern "C" int plv_set_lk_window(plv_ctx *ctx, int win) {
  if (!ctx || win < 3 || win > 21 || (win & 1) == 0) {
    set_last_error("plv_set_lk_window: odd window sizes 3 .. 21");
    return PLV_E_BADARG;
  }
  if (ctx->fe_state) {
    FrontState *s = (FrontState *)ctx->fe_state;
    if (s->pyr_mem[0].p) {
      PyrDesc want{};
      (void)level_geometry(s->W, s->H, win, ctx->cfg.pyr_levels, want);
      if (want.levels != s->pyr[0].levels) {
        set_last_error("plv_set_lk_window: a %d x %d window gives %d pyramid levels, the pyramids of this context have %d", win, win, want.levels, s->pyr[0].levels);
        return PLV_E_BADARG;
      }
    }
  }
  ctx->cfg.win_size = win;
  return PLV_OK;
}
namespace {
FrontState *fe(plv_ctx *ctx) {
  if (!ctx->fe_state) {
    auto *s = new FrontState();
    s->W = ctx->cfg.width;
    s->H = ctx->cfg.height;
    ctx->fe_state = s;
  }
  return (FrontState *)ctx->fe_state;
}

int ensure_pyramids(plv_ctx *ctx, FrontState *s) {
  if (s->pyr_mem[0].p) return PLV_OK;
  if (s->W < 16 || s->H < 16) {
    set_last_error("front-end: image %dx%d too small", s->W, s->H);
    return PLV_E_BADARG;
  }
  for (int i = 0; i < 2; ++i) {
    int bytes = level_geometry(s->W, s->H, ctx->cfg.win_size, ctx->cfg.pyr_levels, s->pyr[i]);
    TRY(s->pyr_mem[i].reserve((size_t)bytes));
    s->pyr[i].base = s->pyr_mem[i].as<uint8_t>();
  }
  if (!s->hist.p) {  // 256 bins + arrival counter; equalize_kernel leaves it zero for the next frame
    TRY(s->hist.reserve(257 * sizeof(unsigned)));
    PLV_HIP_CHECK(hipMemsetAsync(s->hist.p, 0, 257 * sizeof(unsigned), ctx->stream));
  }
  TRY(s->raw.reserve((size_t)s->W * s->H));
  TRY(s->img_pin[0].reserve((size_t)s->W * s->H));  // (pinned blocks of plv_feed_image_enqueue: not allocated inside a frame)
  TRY(s->img_pin[1].reserve((size_t)s->W * s->H));
  return PLV_OK;
}

int sync(plv_ctx *ctx) {
  const unsigned long long stamp = ctx->gather_stamp;
  PLV_HIP_CHECK(plv::stream_sync(ctx->stream));
  ctx->cov_host_synced = stamp;
  ctx->prof.collect();
  return PLV_OK;
}

// equalize + pyramid of the packed device image d_img into the next "current" pyramid
// h_src != null: d_img is still empty and the image sits in pinned host memory at h_src (the first kernel that reads it brings it in)
int feed_device(plv_ctx *ctx, FrontState *s, const uint8_t *d_img, const uint8_t *h_src = nullptr) {
  plv::HostPhase ph("feed image (enqueue hist + pyramid)");
  const int next = s->fed == 0 ? s->cur : 1 - s->cur;
  PyrDesc &p = s->pyr[next];
  const int npix = s->W * s->H;
  switch (ctx->cfg.histogram_method) {
    case PLV_HIST_HISTOGRAM:  // (the equalisation rides on the first pyramid launch)
      TRY(launch_equalize_pyramid(ctx, d_img, p, s->hist.as<unsigned>(), h_src));
      s->cur = next;
      s->fed++;
      return PLV_OK;
    case PLV_HIST_NONE:
      if (h_src) PLV_HIP_CHECK(plv::memcpy_async((void *)d_img, h_src, (size_t)npix, hipMemcpyHostToDevice, ctx->stream));
      PLV_HIP_CHECK(plv::memcpy_async(p.base + p.off[0], d_img, (size_t)npix, hipMemcpyDeviceToDevice, ctx->stream));
      break;
    case PLV_HIST_CLAHE:  // REF: TrackKLT.cpp:60-64 — clip 10.0, 8x8 tiles
      if (h_src) PLV_HIP_CHECK(plv::memcpy_async((void *)d_img, h_src, (size_t)npix, hipMemcpyHostToDevice, ctx->stream));
      TRY(s->clahe_lut.reserve(64 * 256));
      TRY(launch_clahe(ctx, d_img, p.base + p.off[0], s->W, s->H, 10.0, 8, s->clahe_lut.as<uint8_t>()));
      break;
    default:
      set_last_error("front-end: unknown histogram method %d", ctx->cfg.histogram_method);
      return PLV_E_BADARG;
  }
  TRY(launch_pyramid(ctx, p));
  s->cur = next;
  s->fed++;
  return PLV_OK;
}

int upload_image(plv_ctx *ctx, FrontState *s, void *dst, const uint8_t *img, int stride) {
  if (!img || stride < s->W) {
    set_last_error("front-end: null image or stride %d < width %d", stride, s->W);
    return PLV_E_BADARG;
  }
  PLV_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)s->W, img, (size_t)stride, (size_t)s->W, (size_t)s->H, hipMemcpyHostToDevice,
                                 ctx->stream));
  return PLV_OK;
}

}  // namespace

extern "C" {

void plv_frontend_destroy(plv_ctx *ctx) {
  auto *s = (FrontState *)ctx->fe_state;
  if (!s) return;
  if (s->det_pending.active) (void)hipEventSynchronize(s->det_done);
  if (s->det_done) (void)hipEventDestroy(s->det_done);
  if (s->match_done) (void)hipEventDestroy(s->match_done);
  if (s->det_stream) (void)hipStreamDestroy(s->det_stream);
  for (auto &b : s->img_pin) b.release();
  DevBuf *bufs[] = {&s->pyr_mem[0], &s->pyr_mem[1], &s->raw, &s->hist, &s->clahe_lut, &s->ds_src, &s->ds_dst, &s->pts0, &s->pts1, &s->n0, &s->n1,
                    &s->status, &s->iters, &s->mask, &s->counts, &s->info, &s->io, &s->models, &s->det_in, &s->det_out,
                    &s->det_mask, &s->subpix_tab, &s->det_cand, &s->det_cand_n};
  for (auto *b : bufs) b->release();
  for (auto &b : s->slots) b.release();
  s->det_pin.release();
  delete s;
  ctx->fe_state = nullptr;
}

// plv_feed_image without the wait at its end (the tracker's path: plv_tracker_feed goes on to enqueue the flow and waits there).
// The image is copied into one of two pinned blocks of the library first, so the caller's buffer is free when the call returns
// although the transfer and the kernels are only enqueued.
int plv_feed_image_enqueue(plv_ctx *ctx, const uint8_t *img, int stride) {
  if (!ctx) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  FrontState *s = fe(ctx);
  TRY(ensure_pyramids(ctx, s));
  if (!img || stride < s->W) {
    set_last_error("front-end: null image or stride %d < width %d", stride, s->W);
    return PLV_E_BADARG;
  }
  const size_t bytes = (size_t)s->W * s->H;
  // an image the caller wrote straight into one of the library's pinned blocks (plv_image_buffer): nothing to copy on the host
  const uint8_t *h_src = nullptr;
  if (stride == s->W)
    for (auto &b : s->img_pin)
      if (b.p && img >= b.as<uint8_t>() && img + bytes <= b.as<uint8_t>() + b.cap) h_src = img;
  if (!h_src) {
    plv::HostPhase ph("feed image: host copy into the pinned block");
    PinBuf &pin = s->img_pin[s->img_pin_next];
    s->img_pin_next ^= 1;
    TRY(pin.reserve(bytes));
    if (stride == s->W) {
      memcpy(pin.p, img, bytes);
    } else {
      for (int y = 0; y < s->H; ++y) memcpy(pin.as<uint8_t>() + (size_t)y * s->W, img + (size_t)y * stride, s->W);
    }
    h_src = pin.as<uint8_t>();  // (a block is reused two images later: the wait for that image's flow lies in between)
  }
  // no copy command: the histogram kernel reads the pinned block over PCIe (16 bytes per lane) and leaves the image in s->raw
  return feed_device(ctx, s, s->raw.as<uint8_t>(), h_src);
}

// A pinned host block of the library for the caller to write the next image into (a camera driver's DMA target, cv_bridge's copy
// target: `cv::Mat(h, w, CV_8UC1, ptr)`): plv_tracker_feed / plv_camera_frame recognise a pointer into it and skip their own host
// copy.  index 0 .. 3; packed rows (stride = width).  The block must not be overwritten before the call it was handed to has returned.
int plv_image_buffer(plv_ctx *ctx, int index, uint8_t **ptr, int *stride) {
  if (!ctx || !ptr || index < 0 || index >= 4) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  FrontState *s = fe(ctx);
  TRY(ensure_pyramids(ctx, s));
  TRY(s->img_pin[2 + index].reserve((size_t)s->W * s->H));  // (blocks 0 and 1 are the library's own)
  *ptr = s->img_pin[2 + index].as<uint8_t>();
  if (stride) *stride = s->W;
  return PLV_OK;
}

int plv_feed_image(plv_ctx *ctx, const uint8_t *img, int stride) {
  if (!ctx) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  FrontState *s = fe(ctx);
  TRY(ensure_pyramids(ctx, s));
  TRY(upload_image(ctx, s, s->raw.p, img, stride));
  TRY(feed_device(ctx, s, s->raw.as<uint8_t>()));
  return sync(ctx);
}

// cv::pyrDown(img, out, Size(cols / 2.0, rows / 2.0)) on the device: the Size_<int> constructor truncates, so an
// odd dimension halves downwards (the default would be (n + 1) / 2).  REF: UpdaterCamera.cpp:85-98
static int downsample_to(plv_ctx *ctx, FrontState *s, const uint8_t *src, int stride, int sw, int sh, uint8_t *d_dst) {
  if (!src || sw < 2 || sh < 2 || stride < sw) {
    set_last_error("downsample: bad source %dx%d stride %d", sw, sh, stride);
    return PLV_E_BADARG;
  }
  TRY(s->ds_src.reserve((size_t)sw * sh));
  PLV_HIP_CHECK(hipMemcpy2DAsync(s->ds_src.p, (size_t)sw, src, (size_t)stride, (size_t)sw, (size_t)sh, hipMemcpyHostToDevice,
                                 ctx->stream));
  return launch_pyrdown(ctx, s->ds_src.as<uint8_t>(), sw, sh, d_dst, sw / 2, sh / 2);
}

int plv_downsample(plv_ctx *ctx, const uint8_t *src, int stride, int src_w, int src_h, uint8_t *dst, int dstride) {
  if (!ctx || !dst || dstride < src_w / 2) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  FrontState *s = fe(ctx);
  const int dw = src_w / 2, dh = src_h / 2;
  TRY(s->ds_dst.reserve((size_t)dw * dh));
  TRY(downsample_to(ctx, s, src, stride, src_w, src_h, s->ds_dst.as<uint8_t>()));
  PLV_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)dstride, s->ds_dst.p, (size_t)dw, (size_t)dw, (size_t)dh, hipMemcpyDeviceToHost,
                                 ctx->stream));
  return sync(ctx);
}

int plv_feed_image_downsampled(plv_ctx *ctx, const uint8_t *img, int stride, int src_w, int src_h) {
  if (!ctx) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  FrontState *s = fe(ctx);
  TRY(ensure_pyramids(ctx, s));
  if (src_w / 2 != s->W || src_h / 2 != s->H) {
    set_last_error("feed_image_downsampled: %dx%d halves to %dx%d, the context tracks %dx%d", src_w, src_h, src_w / 2, src_h / 2,
                   s->W, s->H);
    return PLV_E_BADARG;
  }
  TRY(downsample_to(ctx, s, img, stride, src_w, src_h, s->raw.as<uint8_t>()));
  TRY(feed_device(ctx, s, s->raw.as<uint8_t>()));
  return sync(ctx);
}

int plv_image_stage(plv_ctx *ctx, int slot, const uint8_t *img, int stride) {
  if (!ctx || slot < 0 || slot >= 8) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  FrontState *s = fe(ctx);
  TRY(ensure_pyramids(ctx, s));
  TRY(s->slots[slot].reserve((size_t)s->W * s->H));
  TRY(upload_image(ctx, s, s->slots[slot].p, img, stride));
  return sync(ctx);
}

int plv_feed_staged(plv_ctx *ctx, int slot) {
  if (!ctx || slot < 0 || slot >= 8) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  FrontState *s = fe(ctx);
  if (!s->slots[slot].p) {
    set_last_error("plv_feed_staged: slot %d is empty", slot);
    return PLV_E_BADARG;
  }
  // stream-ordered, no host sync: the next call on this ctx that returns data synchronises
  return feed_device(ctx, s, s->slots[slot].as<uint8_t>());
}

}  // extern "C"
namespace plv {
int plv_front_fed_count(plv_ctx *ctx) { return (ctx && ctx->fe_state) ? fe(ctx)->fed : 0; }
// where the flow launched last (plv_perform_matching_launch, not waited for yet) leaves its results on the device: tracked positions
// and their normalised coordinates [n][2], the inlier mask [n] (LK status and RANSAC)
int plv_front_match_device(plv_ctx *ctx, const float **d_p1, const float **d_n1, const uint8_t **d_mask, int *n) {
  if (!ctx || !ctx->fe_state) return PLV_E_BADARG;
  FrontState *s = fe(ctx);
  if (s->pending_n < 10 || !s->pending_ran || !s->io.p) return PLV_E_BADARG;
  const size_t nn = (size_t)s->pending_n;
  const char *dp_ = s->io.as<char>();
  *d_p1 = (const float *)(dp_ + nn * 8), *d_n1 = (const float *)(dp_ + nn * 24), *d_mask = (const uint8_t *)(dp_ + nn * 36), *n = s->pending_n;
  return PLV_OK;
}
const uint8_t *plv_front_level0(plv_ctx *ctx, int which, int *w, int *h) {
  if (!ctx || !ctx->fe_state) return nullptr;
  FrontState *s = fe(ctx);
  if (s->fed < (which == PLV_PYR_LAST ? 2 : 1)) return nullptr;
  const PyrDesc &p = s->pyr[which == PLV_PYR_LAST ? 1 - s->cur : s->cur];
  if (w) *w = p.w[0];
  if (h) *h = p.h[0];
  return p.base + p.off[0];
}
}  // namespace plv
extern "C" {

int plv_pyramid_levels(plv_ctx *ctx, int which) {
  if (!ctx || !ctx->fe_state) return 0;
  FrontState *s = fe(ctx);
  if (s->fed < (which == PLV_PYR_LAST ? 2 : 1)) return 0;
  return s->pyr[which == PLV_PYR_LAST ? 1 - s->cur : s->cur].levels;
}

int plv_pyramid_download(plv_ctx *ctx, int which, int level, int *w, int *h, uint8_t *out) {
  if (!ctx || !ctx->fe_state) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  FrontState *s = fe(ctx);
  if (s->fed < (which == PLV_PYR_LAST ? 2 : 1)) return PLV_E_BADARG;
  const PyrDesc &p = s->pyr[which == PLV_PYR_LAST ? 1 - s->cur : s->cur];
  if (level < 0 || level >= p.levels) return PLV_E_BADARG;
  if (w) *w = p.w[level];
  if (h) *h = p.h[level];
  if (out) {
    PLV_HIP_CHECK(plv::memcpy_async(out, p.base + p.off[level], (size_t)p.w[level] * p.h[level], hipMemcpyDeviceToHost,
                                 ctx->stream));
    return sync(ctx);
  }
  return PLV_OK;
}

static int need_two(plv_ctx *ctx, FrontState *s, const char *who) {
  if (s->fed < 2) {
    set_last_error("%s: needs two fed images (last and current pyramid)", who);
    return PLV_E_BADARG;
  }
  (void)ctx;
  return PLV_OK;
}

static int reserve_points(FrontState *s, int n, int ransac_iters) {
  size_t nn = (size_t)std::max(n, 1);
  TRY(s->pts0.reserve(nn * 8));
  TRY(s->pts1.reserve(nn * 8));
  TRY(s->n0.reserve(nn * 8));
  TRY(s->n1.reserve(nn * 8));
  TRY(s->status.reserve(nn));
  TRY(s->mask.reserve(nn));
  TRY(s->iters.reserve(nn * 4));
  TRY(s->counts.reserve((size_t)std::max(ransac_iters, 1) * 3 * 4));
  TRY(s->models.reserve((size_t)std::max(ransac_iters, 1) * 28 * 8));
  TRY(s->info.reserve(16));
  return PLV_OK;
}

int plv_lk_track(plv_ctx *ctx, int n, const float *pts0, float *pts1, uint8_t *status, int *iters) {
  if (!ctx || n < 0 || (n > 0 && (!pts0 || !pts1 || !status))) return PLV_E_BADARG;
  if (n == 0) return PLV_OK;
  (void)hipSetDevice(ctx->device);
  FrontState *s = fe(ctx);
  TRY(need_two(ctx, s, "plv_lk_track"));
  TRY(reserve_points(s, n, 1));
  PLV_HIP_CHECK(plv::memcpy_async(s->pts0.p, pts0, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(s->pts1.p, pts1, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
  TRY(launch_lk(ctx, s->pyr[1 - s->cur], s->pyr[s->cur], n, s->pts0.as<float>(), s->pts1.as<float>(),
                s->status.as<uint8_t>(), s->iters.as<int>(), ctx->cfg.win_size, ctx->cfg.lk_max_iters, ctx->cfg.lk_eps));
  PLV_HIP_CHECK(plv::memcpy_async(pts1, s->pts1.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(status, s->status.p, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  if (iters) PLV_HIP_CHECK(plv::memcpy_async(iters, s->iters.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  return sync(ctx);
}

static CamK cam_of(plv_ctx *ctx) {
  CamK K;
  for (int i = 0; i < 8; ++i) K.v[i] = ctx->cfg.intrinsics[i];
  return K;
}

int plv_undistort(plv_ctx *ctx, int n, const float *uv, float *xy) {
  if (!ctx || n < 0 || (n > 0 && (!uv || !xy))) return PLV_E_BADARG;
  if (n == 0) return PLV_OK;
  (void)hipSetDevice(ctx->device);
  FrontState *s = fe(ctx);
  TRY(reserve_points(s, n, 1));
  PLV_HIP_CHECK(plv::memcpy_async(s->pts0.p, uv, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
  TRY(launch_undistort(ctx, cam_of(ctx), n, s->pts0.as<float>(), s->n0.as<float>()));
  PLV_HIP_CHECK(plv::memcpy_async(xy, s->n0.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
  return sync(ctx);
}

int plv_ransac_fundamental(plv_ctx *ctx, int n, const float *m1, const float *m2, double thr, uint32_t seed,
                           uint8_t *mask, int *n_inliers, int *iters_used) {
  if (!ctx || n < 0 || (n > 0 && (!m1 || !m2 || !mask))) return PLV_E_BADARG;
  if (n_inliers) *n_inliers = 0;
  if (iters_used) *iters_used = 0;
  if (n == 0) return PLV_OK;
  (void)hipSetDevice(ctx->device);
  FrontState *s = fe(ctx);
  const int mi = std::max(1, ctx->cfg.ransac_max_iters);
  TRY(reserve_points(s, n, mi));
  PLV_HIP_CHECK(plv::memcpy_async(s->n0.p, m1, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(s->n1.p, m2, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
  TRY(launch_ransac(ctx, s->n0.as<float>(), s->n1.as<float>(), n, thr, ctx->cfg.ransac_conf, mi, seed, s->counts.as<int>(),
                    nullptr, s->mask.as<uint8_t>(), s->info.as<int>(), s->models.as<double>()));
  int info[2] = {0, 0};
  PLV_HIP_CHECK(plv::memcpy_async(mask, s->mask.p, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  PLV_HIP_CHECK(plv::memcpy_async(info, s->info.p, sizeof(info), hipMemcpyDeviceToHost, ctx->stream));
  TRY(sync(ctx));
  if (n_inliers) *n_inliers = info[0];
  if (iters_used) *iters_used = info[1];
  return PLV_OK;
}

int plv_perform_matching_launch(plv_ctx *ctx, int n, const float *pts0, const float *pts1_init) {
  if (!ctx || n < 0 || (n > 0 && (!pts0 || !pts1_init))) return PLV_E_BADARG;
  FrontState *s0 = fe(ctx);
  s0->pending_n = -1;
  if (n < 10) {  // REF: TrackKLT.cpp:848-852 (nothing to run: the wait returns an all-zero mask)
    s0->pending_n = n;
    s0->pending_ran = false;
    return PLV_OK;
  }
  (void)hipSetDevice(ctx->device);
  FrontState *s = s0;
  TRY(need_two(ctx, s, "plv_perform_matching"));
  const int mi = std::max(1, ctx->cfg.ransac_max_iters);
  TRY(reserve_points(s, n, mi));
  // one device block  [pts0 | pts1 | n0 | n1 | iters | mask | status]  -> one H2D (first two
  // fields) and one D2H (pts1 .. mask) per call
  const size_t nn = (size_t)n;
  const size_t o_p0 = 0, o_p1 = nn * 8, o_n0 = nn * 16, o_n1 = nn * 24, o_it = nn * 32, o_mk = nn * 36, o_st = nn * 37,
               total = nn * 38;
  TRY(s->io.reserve(total));
  TRY(ctx->h_pin_flow.reserve(total));
  char *hp = ctx->h_pin_flow.as<char>();
  char *dp_ = s->io.as<char>();
  // No copy commands around the two kernels: lk_kernel reads the points and the initial guesses straight from the pinned buffer
  // (16 B per point, once) and the last kernel of the call copies the results back into it (ransac_select_kernel).
  memcpy(hp + o_p0, pts0, nn * 8);
  memcpy(hp + o_p1, pts1_init, nn * 8);
  float *d_p1 = (float *)(dp_ + o_p1), *d_n0 = (float *)(dp_ + o_n0), *d_n1 = (float *)(dp_ + o_n1);
  int *d_it = (int *)(dp_ + o_it);
  uint8_t *d_mk = (uint8_t *)(dp_ + o_mk), *d_st = (uint8_t *)(dp_ + o_st);
  const CamK camk = cam_of(ctx);
  TRY(launch_lk(ctx, s->pyr[1 - s->cur], s->pyr[s->cur], n, (const float *)(hp + o_p0), d_p1, d_st, d_it, ctx->cfg.win_size,
                ctx->cfg.lk_max_iters, ctx->cfg.lk_eps, &camk, d_n0, d_n1,
                (const float *)(hp + o_p1)));  // (+ the undistortion of both point sets on the same launch)
  const double fmax = std::max(ctx->cfg.intrinsics[0], ctx->cfg.intrinsics[1]);
  bool mirrored = false;
  TRY(ctx->h_done.reserve(256));
  const unsigned seq = ++ctx->match_seq;
  TRY(launch_ransac(ctx, d_n0, d_n1, n, ctx->cfg.ransac_thr_px / fmax, ctx->cfg.ransac_conf, mi, 0u, s->counts.as<int>(), d_st,
                    d_mk, s->info.as<int>(), s->models.as<double>(), dp_ + o_p1, hp + o_p1, o_mk - o_p1, (uint8_t *)(hp + o_mk), &mirrored,
                    (unsigned *)ctx->done_word(0), seq));
  s->match_word_seq = mirrored ? seq : 0;
  if (!mirrored) PLV_HIP_CHECK(plv::memcpy_async(hp + o_p1, dp_ + o_p1, o_st - o_p1, hipMemcpyDeviceToHost, ctx->stream));
  if (!s->match_done) PLV_HIP_CHECK(hipEventCreateWithFlags(&s->match_done, hipEventDisableTiming));
  PLV_HIP_CHECK(hipEventRecord(s->match_done, ctx->stream));
  s->match_done_stamp = ctx->gather_stamp;
  s->pending_n = n;
  s->pending_ran = true;
  return PLV_OK;
}

int plv_perform_matching_wait(plv_ctx *ctx, float *pts1, uint8_t *mask_out, float *n0, float *n1, long long *lk_iters) {
  if (!ctx) return PLV_E_BADARG;
  FrontState *s = fe(ctx);
  const int n = s->pending_n;
  if (n < 0) {
    set_last_error("plv_perform_matching_wait: nothing was launched");
    return PLV_E_BADARG;
  }
  s->pending_n = -1;
  if (lk_iters) *lk_iters = 0;
  if (n == 0) return PLV_OK;
  if (!pts1 || !mask_out) return PLV_E_BADARG;
  if (!s->pending_ran) {
    memset(mask_out, 0, (size_t)n);
    return PLV_OK;
  }
  (void)hipSetDevice(ctx->device);
  if (ctx->prof.on)
    TRY(sync(ctx));  // (the per-kernel timer reads every event recorded so far)
  else
  {
    // not the whole stream: the caller may have enqueued more behind the flow.  The flow's last kernel says when its results are in
    // pinned memory; everything enqueued on the stream before it has finished by then as well (in-order stream).
    if (s->match_word_seq && plv::knob(plv::PLV_KNOB_DONE_WORDS))
      PLV_HIP_CHECK(plv::wait_done_word(ctx->done_word(0), s->match_word_seq, s->match_done));
    else
      PLV_HIP_CHECK(plv::event_sync(s->match_done));
    if (s->match_done_stamp > ctx->cov_host_synced) ctx->cov_host_synced = s->match_done_stamp;
  }
  const size_t nn = (size_t)n;
  const size_t o_p1 = nn * 8, o_n0 = nn * 16, o_n1 = nn * 24, o_it = nn * 32, o_mk = nn * 36;
  const char *hp = ctx->h_pin_flow.as<char>();
  memcpy(pts1, hp + o_p1, nn * 8);
  if (n0) memcpy(n0, hp + o_n0, nn * 8);
  if (n1) memcpy(n1, hp + o_n1, nn * 8);
  memcpy(mask_out, hp + o_mk, nn);
  long long t = 0;
  const int *it = (const int *)(hp + o_it);
  for (int i = 0; i < n; ++i) t += it[i];
  plv::counters().lk_iters += (unsigned long long)t;
  if (lk_iters) *lk_iters = t;
  return PLV_OK;
}

int plv_perform_matching(plv_ctx *ctx, int n, const float *pts0, float *pts1, uint8_t *mask_out, float *n0, float *n1,
                         long long *lk_iters) {
  if (!ctx || n < 0 || (n > 0 && (!pts0 || !pts1 || !mask_out))) return PLV_E_BADARG;
  if (lk_iters) *lk_iters = 0;
  if (n == 0) return PLV_OK;
  TRY(plv_perform_matching_launch(ctx, n, pts0, pts1));
  return plv_perform_matching_wait(ctx, pts1, mask_out, n0, n1, lk_iters);
}


// plv_perform_detection replaces TrackKLT::perform_detection_monocular (REF: open_vins/ov_core/src/
// track/TrackKLT.cpp:395-528).  The occupancy-grid bookkeeping is host logic exactly as in the
// reference (a few hundred integer operations); the per-cell FAST + top-k (Grider_GRID.h:108-151)
// and the sub-pixel refinement (:163-174) run on the device on level 0 of the chosen pyramid.
// Three stages: det_pre (host: grids, cells that need features), det_launch (device, on any stream), det_post (host: reject
// near existing points, ids in extraction order).  plv_perform_detection runs them back to back; plv_perform_detection_ahead runs
// the first two on a side stream as soon as a frame's points are known, so that the top-up the NEXT frame starts with (it works on
// the then-last image with these very points, TrackKLT.cpp:127-131) is waiting for it instead of sitting on its critical path.
namespace {
int det_pre(plv_ctx *ctx, FrontState *s, const uint8_t *mask, const float *pts_in, const uint64_t *ids_in, int n_in, DetJob &J) {
  const int w = s->W, h = s->H;
  const plv_config &c = ctx->cfg;
  const int min_px = c.min_px_dist, grid_x = c.grid_x, grid_y = c.grid_y, num_features = c.num_features;
  if (min_px < 1 || grid_x < 1 || grid_y < 1) return PLV_E_BADARG;
  // ---- REF :401-464: occupancy grids, drop edge / masked / too-close points, remember the painted boxes
  J.cw = (int)((float)w / (float)min_px), J.ch = (int)((float)h / (float)min_px);
  J.close.assign((size_t)J.cw * J.ch, 0);
  std::vector<uint8_t> grid((size_t)grid_x * grid_y, 0);
  const float size_x = (float)w / (float)grid_x, size_y = (float)h / (float)grid_y;
  J.boxes.clear();
  J.pts.clear();
  J.ids.clear();
  for (int i = 0; i < n_in; ++i) {
    const float fx = pts_in[2 * i], fy = pts_in[2 * i + 1];
    const int x = (int)fx, y = (int)fy;
    const int edge = 10;
    if (x < edge || x >= w - edge || y < edge || y >= h - edge) continue;
    const int xc = (int)(fx / (float)min_px), yc = (int)(fy / (float)min_px);
    if (xc < 0 || xc >= J.cw || yc < 0 || yc >= J.ch) continue;
    const int xg = (int)std::floor(fx / size_x), yg = (int)std::floor(fy / size_y);
    if (xg < 0 || xg >= grid_x || yg < 0 || yg >= grid_y) continue;
    if (J.close[(size_t)yc * J.cw + xc] > 127) continue;
    if (mask && mask[(size_t)y * w + x] > 127) continue;
    J.close[(size_t)yc * J.cw + xc] = 255;
    if (grid[(size_t)yg * grid_x + xg] < 255) grid[(size_t)yg * grid_x + xg] += 1;
    if (x - min_px >= 0 && x + min_px < w && y - min_px >= 0 && y + min_px < h) {
      J.boxes.push_back(x);
      J.boxes.push_back(y);
    }
    J.pts.push_back(fx);
    J.pts.push_back(fy);
    J.ids.push_back(ids_in[i]);
  }
  J.n_cells = 0;
  J.n_slots = 0;
  // ---- REF :466-471
  const double min_feat_percent = 0.50;
  const int n = (int)J.ids.size();
  if (num_features - n < std::min(20, (int)(min_feat_percent * num_features))) return PLV_OK;
  // ---- REF :478-492 cells that still need features and are not fully masked
  const int nfg_req = std::max(1, (int)(min_feat_percent * ((int)((double)num_features / (double)(grid_x * grid_y)) + 1)));
  // REF Grider_GRID.h:88-98 grid actually used for extraction
  int gx = grid_x, gy = grid_y;
  if (num_features < gx * gy) {
    double ratio = (double)gx / (double)gy;
    gy = (int)std::ceil(std::sqrt(num_features / ratio));
    gx = (int)std::ceil(gy * ratio);
  }
  J.nfg = (int)((double)num_features / (double)(gx * gy)) + 1;
  J.sxp = w / gx, J.syp = h / gy;
  J.cells.clear();
  for (int x = 0; x < grid_x; ++x)
    for (int y = 0; y < grid_y; ++y) {
      const int sx = std::min((int)std::floor(x * (double)w / grid_x), w - 1), sy = std::min((int)std::floor(y * (double)h / grid_y), h - 1);
      const bool masked = mask && mask[(size_t)sy * w + sx] == 255;
      if ((int)grid[(size_t)y * grid_x + x] < nfg_req && !masked) {
        if (x * J.sxp + J.sxp > w || y * J.syp + J.syp > h) continue;  // REF Grider_GRID.h:117-118
        J.cells.push_back(x);
        J.cells.push_back(y);
      }
    }
  const int n_cells = (int)J.cells.size() / 2;
  if (n_cells == 0 || J.sxp < 7 || J.syp < 7) return PLV_OK;
  J.n_cells = n_cells;
  J.n_slots = n_cells * J.nfg;
  return PLV_OK;
}

// FAST per cell + sub-pixel refinement of every kept slot + the copy of the slots to pinned memory, all on `stream`
int det_launch(plv_ctx *ctx, FrontState *s, const PyrDesc &pyr, const uint8_t *mask, DetJob &J, hipStream_t stream) {
  const int w = s->W, h = s->H;
  const plv_config &c = ctx->cfg;
  const int n_cells = J.n_cells, n_slots = J.n_slots;
  const size_t o_cells = 0, o_boxes = ((size_t)n_cells * 8 + 15) & ~(size_t)15, in_total = o_boxes + ((J.boxes.size() * 4 + 15) & ~(size_t)15);
  TRY(s->det_pin.reserve(in_total + (size_t)n_slots * 13 + 128));
  TRY(s->det_in.reserve(in_total + 16));
  const size_t o_xy = 0, o_resp = (size_t)n_slots * 8, o_valid = (size_t)n_slots * 12, out_total = (size_t)n_slots * 13;
  TRY(s->det_out.reserve(out_total + 16));
  char *hp = s->det_pin.as<char>();
  J.h_out = hp + ((in_total + 63) & ~(size_t)63);  // results land behind the inputs (both may be in flight at once)
  memcpy(hp + o_cells, J.cells.data(), J.cells.size() * 4);
  if (!J.boxes.empty()) memcpy(hp + o_boxes, J.boxes.data(), J.boxes.size() * 4);
  PLV_HIP_CHECK(plv::memcpy_async(s->det_in.p, hp, in_total, hipMemcpyHostToDevice, stream));
  const uint8_t *d_mask = nullptr;
  if (mask) {
    TRY(s->det_mask.reserve((size_t)w * h));
    PLV_HIP_CHECK(plv::memcpy_async(s->det_mask.p, mask, (size_t)w * h, hipMemcpyHostToDevice, stream));
    d_mask = s->det_mask.as<uint8_t>();
  }
  if (!s->subpix_tab.p) {  // cv::cornerSubPix window weights exp(-(x/5)^2) exp(-(y/5)^2)
    float tab[121];
    for (int i = 0; i < 11; ++i) {
      float y = (float)(i - 5) / 5;
      float vy = std::exp(-y * y);
      for (int j = 0; j < 11; ++j) {
        float x = (float)(j - 5) / 5;
        tab[i * 11 + j] = (float)(vy * std::exp(-x * x));
      }
    }
    TRY(s->subpix_tab.reserve(sizeof(tab)));
    PLV_HIP_CHECK(hipMemcpy(s->subpix_tab.p, tab, sizeof(tab), hipMemcpyHostToDevice));
  }
  DetectParams P{};
  P.img = pyr.base + pyr.off[0];
  P.W = w;
  P.H = h;
  P.mask = d_mask;
  P.cells = (const int *)(s->det_in.as<char>() + o_cells);
  P.cell_w = J.sxp;
  P.cell_h = J.syp;
  P.threshold = c.fast_threshold;
  P.nfg = J.nfg;
  P.cand_cap = 4096;
  P.boxes = (const int *)(s->det_in.as<char>() + o_boxes);
  P.n_boxes = (int)J.boxes.size() / 2;
  P.min_px_dist = c.min_px_dist;
  P.out_xy = (float *)(s->det_out.as<char>() + o_xy);
  P.out_resp = (float *)(s->det_out.as<char>() + o_resp);
  P.out_valid = (uint8_t *)(s->det_out.as<char>() + o_valid);
  {  // per-cell candidate lists of fast_tiles_kernel; the counters start at zero and fast_topk_kernel leaves them at zero
    const size_t all_cells = (size_t)c.grid_x * c.grid_y;
    if (s->det_cand_n.cap < all_cells * 4) {
      TRY(s->det_cand_n.reserve(all_cells * 4));
      PLV_HIP_CHECK(hipMemsetAsync(s->det_cand_n.p, 0, s->det_cand_n.cap, stream));
    }
    TRY(s->det_cand.reserve(all_cells * (size_t)P.cand_cap * 8));
  }
  hipStream_t keep = ctx->stream;
  ctx->stream = stream;  // (the launchers take the stream from the ctx)
  int rc = launch_fast_cells(ctx, P, n_cells, s->det_cand.as<unsigned long long>(), s->det_cand_n.as<int>());
  if (rc == PLV_OK) rc = launch_subpix(ctx, P.img, w, h, n_slots, P.out_valid, P.out_xy, s->subpix_tab.as<float>(), 5, 20, 0.001);
  ctx->stream = keep;
  TRY(rc);
  PLV_HIP_CHECK(plv::memcpy_async(J.h_out, s->det_out.p, out_total, hipMemcpyDeviceToHost, stream));
  return PLV_OK;
}

// ---- REF :497-527 reject near existing points, assign ids in extraction order
void det_post(plv_ctx *ctx, const DetJob &J, float *pts, uint64_t *ids, int cap, uint64_t *currid, int *n_out) {
  int n = (int)J.ids.size();
  std::copy(J.pts.begin(), J.pts.end(), pts);
  std::copy(J.ids.begin(), J.ids.end(), ids);
  if (J.n_slots > 0) {
    const int min_px = ctx->cfg.min_px_dist;
    std::vector<uint8_t> close = J.close;
    const float *oxy = (const float *)(J.h_out);
    const uint8_t *oval = (const uint8_t *)(J.h_out + (size_t)J.n_slots * 12);
    for (int sl = 0; sl < J.n_slots && n < cap; ++sl) {
      if (!oval[sl]) continue;
      const float px = oxy[2 * sl], py = oxy[2 * sl + 1];
      const int xg = (int)(px / (float)min_px), yg = (int)(py / (float)min_px);
      if (xg < 0 || xg >= J.cw || yg < 0 || yg >= J.ch) continue;
      if (close[(size_t)yg * J.cw + xg] > 127) continue;
      close[(size_t)yg * J.cw + xg] = 255;
      pts[2 * n] = px;
      pts[2 * n + 1] = py;
      ids[n] = ++*currid;
      ++n;
    }
  }
  *n_out = n;
}

// a detection that was started ahead of time and has not been collected: wait for it so that its buffers can be reused
void det_drop_pending(FrontState *s);
}  // namespace
// plv_ctx_synchronize: the side stream's work belongs to the frame that started it — wait for it too (the job stays collectable)
extern "C" int plv_front_quiesce(plv_ctx *ctx) {
  if (!ctx || !ctx->fe_state) return PLV_OK;
  FrontState *s = (FrontState *)ctx->fe_state;
  if (s->det_pending.active && s->det_done) PLV_HIP_CHECK(plv::event_sync(s->det_done));
  return PLV_OK;
}
namespace {
void det_drop_pending(FrontState *s) {
  if (s->det_pending.active) {
    (void)plv::event_sync(s->det_done);
    s->det_pending.active = false;
  }
}
}  // namespace

int plv_perform_detection(plv_ctx *ctx, int which, const uint8_t *mask, float *pts, uint64_t *ids, int n_in, int cap,
                          uint64_t *currid, int *n_out) {
  if (!ctx || !pts || !ids || !currid || !n_out || n_in < 0 || cap < n_in) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  FrontState *s = fe(ctx);
  if (s->fed < (which == PLV_PYR_LAST ? 2 : 1)) {
    set_last_error("plv_perform_detection: no %s pyramid yet", which == PLV_PYR_LAST ? "last" : "current");
    return PLV_E_BADARG;
  }
  // ---- a detection started ahead of time on what is now the last image, with these very points (and mask)?
  DetJob &A = s->det_pending;
  if (A.active) {
    const bool same = which == PLV_PYR_LAST && A.fed + 1 == s->fed && A.n_in == n_in && A.has_mask == (mask != nullptr) &&
                      (n_in == 0 || (!memcmp(A.in_pts.data(), pts, (size_t)n_in * 8) && !memcmp(A.in_ids.data(), ids, (size_t)n_in * 8))) &&
                      (!mask || !memcmp(A.in_mask.data(), mask, (size_t)s->W * s->H));
    (void)plv::event_sync(s->det_done);
    A.active = false;
    if (same) {
      if ((int)A.ids.size() > cap) return PLV_E_CAPACITY;
      det_post(ctx, A, pts, ids, cap, currid, n_out);
      return PLV_OK;
    }
  }
  const PyrDesc &pyr = s->pyr[which == PLV_PYR_LAST ? 1 - s->cur : s->cur];
  DetJob J;
  TRY(det_pre(ctx, s, mask, pts, ids, n_in, J));
  if (J.n_slots > 0) {
    TRY(det_launch(ctx, s, pyr, mask, J, ctx->stream));
    TRY(sync(ctx));
  }
  det_post(ctx, J, pts, ids, cap, currid, n_out);
  return PLV_OK;
}

// The top-up detection of the NEXT frame, started now: on the current image (the next frame's last image) with the points this frame
// ended with.  Runs on a side stream next to whatever the caller enqueues on the ctx stream (the updates); plv_perform_detection of
// the next frame finds it finished (the per-kernel profiler follows it onto the side stream: Profiler::collect).
// on_ctx_stream: enqueue behind what is on the ctx stream instead (the caller has submitted an update whose wait ends at its own
// last kernel: the detection then runs in the device's idle time between that update and the next submission, not next to it).
int plv_perform_detection_ahead(plv_ctx *ctx, const uint8_t *mask, const float *pts, const uint64_t *ids, int n_in, int on_ctx_stream) {
  if (!ctx || n_in < 0 || (n_in > 0 && (!pts || !ids))) return PLV_E_BADARG;
  (void)hipSetDevice(ctx->device);
  FrontState *s = fe(ctx);
  if (s->fed < 1) return PLV_OK;
  det_drop_pending(s);
  DetJob &A = s->det_pending;
  TRY(det_pre(ctx, s, mask, pts, ids, n_in, A));
  A.fed = s->fed;
  A.n_in = n_in;
  A.has_mask = mask != nullptr;
  A.in_pts.assign(pts, pts + 2 * (size_t)n_in);
  A.in_ids.assign(ids, ids + n_in);
  if (mask)
    A.in_mask.assign(mask, mask + (size_t)s->W * s->H);
  else
    A.in_mask.clear();
  if (!s->det_stream) {
    PLV_HIP_CHECK(hipStreamCreateWithFlags(&s->det_stream, hipStreamNonBlocking));
    PLV_HIP_CHECK(hipEventCreateWithFlags(&s->det_done, hipEventDisableTiming));
  }
  hipStream_t st = on_ctx_stream ? ctx->stream : s->det_stream;
  if (A.n_slots > 0) TRY(det_launch(ctx, s, s->pyr[s->cur], mask ? A.in_mask.data() : nullptr, A, st));  // (the job's own copy of the mask)
  PLV_HIP_CHECK(hipEventRecord(s->det_done, st));
  A.active = true;
  return PLV_OK;
}

}  // extern "C"
