// TrackLSD_HIP.h — viw::TrackLSD's interface on libplviwo_hip.so.
// REF: PL-VIWO/src/update/cam/TrackLSD.h:74-100, TrackLSD.cpp:39-192 (feed_new_camera / feed_monocular), :194-235 (detector),
//      :744-814 (AssignPointToLines), :368-407 (LineMatch), :318-366 (LineClassification),
//      linefeat/LineHelper.cpp:1026-1088 (Vanishing_Points), linefeat/LineFeatureDatabase.cpp:40-76 (update_feature)
//
// The image is the one the point tracker of the same context was fed last (UpdaterCamera.cpp:105-109 calls the point tracker
// first; the reference's second equalizeHist, TrackLSD.cpp:83, reproduces the same image).  The line track store lives in the
// library (plv_line_db_*); `export_tracks` hands it out in the layout of plv_line_tracks for code that wants LineFeature objects.
//
// The first constructor has TrackLSD's own signature (TrackLSD.h:85-86): the reference's call site
//   new TrackLSD(state->cam_intrinsic_model, op->use_stereo, op->histogram, trackFEATS)
// (UpdaterCamera.cpp:44) compiles with the class name changed; the context is the one of the camera's point tracker
// (trackFEATS holds TrackKLT_HIP objects), which the line tracker shares as the reference's shares the point tracker's outputs
// (TrackLSD.cpp:100-101,127-129: get_last_obs / get_last_ids).
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <unordered_map>
#include <vector>

#include <Eigen/Eigen>

#include "TrackKLT_HIP.h"
#include "plviwo.h"
#include "utils/print.h"
#include "utils/sensor_data.h"

namespace viw {

class TrackLSD_HIP {
public:
  // REF: TrackLSD.h:85-86 — the reference's signature
  TrackLSD_HIP(std::unordered_map<size_t, std::shared_ptr<ov_core::CamBase>> cameras, bool stereo, ov_core::TrackBase::HistogramMethod histmethod,
               std::map<int, std::shared_ptr<ov_core::TrackBase>> _trackFEATS)
      : ctx(nullptr) {
    (void)cameras, (void)histmethod;  // (intrinsics and equalisation are the point tracker's: same context, same image)
    std::shared_ptr<ov_core::TrackKLT_HIP> klt;
    for (auto &kv : _trackFEATS)
      if ((klt = std::dynamic_pointer_cast<ov_core::TrackKLT_HIP>(kv.second))) break;
    if (stereo || !klt) {
      PRINT_ERROR(RED "[TrackLSD_HIP]: needs the monocular TrackKLT_HIP of its camera in trackFEATS\n" RESET);
      std::exit(EXIT_FAILURE);
    }
    ctx = klt->context();
    plv_line_prefetch_mode(ctx, 1);
  }

  explicit TrackLSD_HIP(plv_ctx *ctx_, bool prefetch = true) : ctx(ctx_) {
    plv_line_prefetch_mode(ctx, prefetch ? 1 : 0);  // the detector's host stage runs while the device tracks the points
  }

  // LineHelper::Vanishing_Points(state): R_ItoC row-major, K8 = (fx fy cx cy k1 k2 p1 p2)
  static std::vector<Eigen::Vector2d> vanishing_points(const Eigen::Matrix3d &R_ItoC, const Eigen::Matrix<double, 8, 1> &K8) {
    Eigen::Matrix<double, 3, 3, Eigen::RowMajor> R = R_ItoC;
    double v[6];
    plv_vanishing_points(R.data(), K8.data(), v);
    return {Eigen::Vector2d(v[0], v[1]), Eigen::Vector2d(v[2], v[3]), Eigen::Vector2d(v[4], v[5])};
  }

  void feed_new_camera(const ov_core::CameraData &message, std::vector<Eigen::Vector2d> &vps, bool async = false) {
    const double v[6] = {vps.at(0).x(), vps.at(0).y(), vps.at(1).x(), vps.at(1).y(), vps.at(2).x(), vps.at(2).y()};
    const int rc = async ? plv_line_tracker_feed_async(ctx, message.timestamp, v) : plv_line_tracker_feed(ctx, message.timestamp, v);
    if (rc != PLV_OK) {
      PRINT_ERROR(RED "[TrackLSD_HIP]: %s\n" RESET, plv_last_error());
      std::exit(EXIT_FAILURE);
    }
  }
  void wait() { plv_line_tracker_feed_wait(ctx); }

  // TrackLSD::lines_last / ids_last of the newest frame
  void get_last(std::vector<Eigen::Vector4f> &lines, std::vector<size_t> &ids) {
    int n = 0;
    plv_line_tracker_last(ctx, nullptr, nullptr, 1 << 30, &n);
    std::vector<float> l(4 * (size_t)n + 4);
    std::vector<uint64_t> id((size_t)n + 1);
    plv_line_tracker_last(ctx, l.data(), id.data(), n, &n);
    lines.resize(n);
    ids.resize(n);
    for (int i = 0; i < n; ++i) {
      lines[i] = Eigen::Vector4f(l[4 * i], l[4 * i + 1], l[4 * i + 2], l[4 * i + 3]);
      ids[i] = (size_t)id[i];
    }
  }

  plv_ctx *context() const { return ctx; }

private:
  plv_ctx *ctx;
};

}  // namespace viw
