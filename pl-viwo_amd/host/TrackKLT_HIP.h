// TrackKLT_HIP.h — ov_core::TrackBase implementation on libplviwo_hip.so.
// REF: open_vins/ov_core/src/track/TrackBase.h:72-196 (interface), TrackKLT.cpp:34-200 (feed_new_camera / feed_monocular),
//      :395-528 (perform_detection_monocular), :829-886 (perform_matching), feat/FeatureDatabase.cpp:60-120 (update_feature)
//
// The whole frame logic runs behind plv_tracker_feed (first-frame detection, top-up on the last image, pyramidal LK with the
// previous positions as the initial flow, undistortion, 7-point RANSAC, bounds / mask filter, id hand-over).  The adapter keeps
// TrackBase's observable state current: pts_last / ids_last (get_last_obs / get_last_ids, read by TrackLSD and the display code),
// img_last / img_mask_last, and the ov_core::FeatureDatabase (update_feature with the frame's observations), so code that walks
// `get_feature_database()` keeps working.  UpdaterCameraHIP.h uses the library's own track store instead and does not need it.
//
// Two constructors.  The first has TrackKLT's own signature (TrackKLT.h:54-58), so the reference's call site
//   new TrackKLT(state->cam_intrinsic_model, op->n_pts, 0, op->use_stereo, op->histogram, op->fast, op->grid_x, op->grid_y, op->min_px_dist)
// (UpdaterCamera.cpp:41,63) compiles with the class name changed and nothing else: the tracker creates and owns its plv_ctx
// (image size and intrinsics from the CamBase of its camera, everything else from the arguments / the reference's constants) and
// TrackLSD_HIP / UpdaterCameraHIP / StateHelperHIP find it through context().  The second takes a ctx the caller made (PlvContext.h).
#pragma once
#include <cstdlib>
#include <memory>
#include <mutex>
#include <vector>

#include "plviwo.h"
#include "track/TrackBase.h"
#include "utils/print.h"

namespace ov_core {

class TrackKLT_HIP : public TrackBase {
public:
  // REF: TrackKLT.h:54-58 — the reference's signature; the context is created here and destroyed with the tracker
  explicit TrackKLT_HIP(std::unordered_map<size_t, std::shared_ptr<CamBase>> cameras, int numfeats, int numaruco, bool stereo,
                        HistogramMethod histmethod, int fast_threshold, int gridx, int gridy, int minpxdist)
      : TrackBase(cameras, numfeats, numaruco, stereo, histmethod), ctx(nullptr), mirror_db(true) {
    if (stereo || cameras.empty()) {  // monocular path only (stereo is out of scope)
      PRINT_ERROR(RED "[TrackKLT_HIP]: one monocular camera per tracker\n" RESET);
      std::exit(EXIT_FAILURE);
    }
    const std::shared_ptr<CamBase> &cam = cameras.begin()->second;
    plv_config cfg;
    plv_config_default(&cfg, cam->w(), cam->h());  // CamBase.h:190-193
    cfg.num_features = numfeats;
    cfg.fast_threshold = fast_threshold;
    cfg.grid_x = gridx, cfg.grid_y = gridy;
    cfg.min_px_dist = minpxdist;
    cfg.histogram_method = (int)histmethod;  // PLV_HIST_* follow TrackBase::HistogramMethod (TrackBase.h:78)
    const Eigen::MatrixXd K = cam->get_value();  // fx fy cx cy k1 k2 p1 p2 (CamBase.h:56-82,181)
    for (int i = 0; i < 8; ++i) cfg.intrinsics[i] = K(i, 0);
    if (const char *dev = std::getenv("PLV_DEVICE")) cfg.device = std::atoi(dev);
    plv_ctx *c = nullptr;
    if (plv_ctx_create(&cfg, &c) != PLV_OK) {
      PRINT_ERROR(RED "[TrackKLT_HIP]: %s\n" RESET, plv_last_error());
      std::exit(EXIT_FAILURE);  // no CPU fallback: the library needs a gfx950 device
    }
    owned = std::shared_ptr<plv_ctx>(c, plv_ctx_destroy);
    ctx = c;
  }

  // a context the caller created (PlvContext.h: sized from the estimator's options as well)
  TrackKLT_HIP(std::unordered_map<size_t, std::shared_ptr<CamBase>> cameras, int numfeats, int numaruco, bool stereo,
               HistogramMethod histmethod, plv_ctx *ctx_, bool mirror_database = true)
      : TrackBase(cameras, numfeats, numaruco, stereo, histmethod), ctx(ctx_), mirror_db(mirror_database) {}

  // the library keeps its own track store (plv_db_*); UpdaterCameraHIP turns the ov_core::FeatureDatabase mirror off
  void mirror_database(bool on) { mirror_db = on; }

  void feed_new_camera(const CameraData &message) override {
    if (message.sensor_ids.size() != 1) {  // monocular path only (stereo is out of scope)
      PRINT_ERROR(RED "[TrackKLT_HIP]: one camera per context\n" RESET);
      std::exit(EXIT_FAILURE);
    }
    const size_t cam_id = message.sensor_ids.at(0);
    std::lock_guard<std::mutex> lck(mtx_feeds.at(cam_id));
    const cv::Mat &img = message.images.at(0), &mask = message.masks.at(0);
    if (img.type() != CV_8UC1 || !mask.isContinuous() ||
        plv_tracker_feed(ctx, message.timestamp, img.data, (int)img.step, mask.empty() ? nullptr : mask.data) != PLV_OK) {
      PRINT_ERROR(RED "[TrackKLT_HIP]: %s\n" RESET, plv_last_error());  // TrackKLT.cpp:37-43 exits on bad sizes as well
      std::exit(EXIT_FAILURE);
    }
    int n = 0;
    plv_tracker_last(ctx, nullptr, nullptr, 1 << 30, &n);
    std::vector<float> xy(2 * (size_t)n + 2);
    std::vector<uint64_t> ids((size_t)n + 1);
    plv_tracker_last(ctx, xy.data(), ids.data(), n, &n);
    std::vector<cv::KeyPoint> kps(n);
    std::vector<size_t> kid(n);
    for (int i = 0; i < n; ++i) {
      kps[i].pt = cv::Point2f(xy[2 * i], xy[2 * i + 1]);
      kid[i] = (size_t)ids[i];
    }
    if (mirror_db && have_last) {  // TrackKLT.cpp:176-179: every surviving point is an observation of this frame
      for (int i = 0; i < n; ++i) {
        const cv::Point2f npt = camera_calib.at(cam_id)->undistort_cv(kps[i].pt);
        database->update_feature(kid[i], message.timestamp, cam_id, kps[i].pt.x, kps[i].pt.y, npt.x, npt.y);
      }
    }
    have_last = true;
    std::lock_guard<std::mutex> lckv(mtx_last_vars);  // TrackKLT.cpp:182-189
    img_last[cam_id] = img;
    img_mask_last[cam_id] = mask;
    pts_last[cam_id] = kps;
    ids_last[cam_id] = kid;
  }

  plv_ctx *context() const { return ctx; }

protected:
  std::shared_ptr<plv_ctx> owned;  // (first constructor)
  plv_ctx *ctx;
  bool mirror_db, have_last = false;
};

}  // namespace ov_core
