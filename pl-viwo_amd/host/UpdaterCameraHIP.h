// UpdaterCameraHIP.h — UpdaterCamera::feed_measurement + try_update (MSCKF points and lines) on the library's track stores.
// REF: PL-VIWO/src/update/cam/UpdaterCamera.cpp:77-116 (feed_measurement), :139-195 (try_update), :197-294 (msckf_update),
//      :371-464 (lines_update); CamHelper.cpp:613-738 (get_features / cleanup_features); linefeat/LineHelper.cpp:19-72,522-553
//
// Covariance mode (b) of StateHelperHIP.h: the caller uploads state->cov after propagation / cloning on the host, or runs those
// through plv_propagate / plv_cov_clone / plv_cov_marginalize as well, and downloads it when host code needs it.
// In-state landmarks (max_slam > 0) keep the reference's per-landmark loops over plv_camera_update_list.
#pragma once
#include <deque>
#include <memory>
#include <vector>

#include <Eigen/Eigen>

#include "StateHelperHIP.h"
#include "TrackKLT_HIP.h"
#include "TrackLSD_HIP.h"
#include "plviwo.h"
#include "state/State.h"

namespace viw {

class UpdaterCameraHIP {
public:
  UpdaterCameraHIP(std::shared_ptr<State> state_, plv_ctx *ctx_, std::shared_ptr<ov_core::TrackKLT_HIP> klt, std::shared_ptr<TrackLSD_HIP> lsd,
                   int max_obs_ = 24)
      : state(state_), ctx(ctx_), trackFEATS(klt), trackLSDS(lsd), max_obs(max_obs_) {}

  // the window as the kernels read it: clone times ascending, rotations row-major, first estimates, covariance ids
  struct View {
    std::vector<double> t, R, p, Rf, pf;
    std::vector<int> ids;
    plv_state_view v;
  };
  void build_view(View &w) const {
    const int cam_id = 0;
    w.t.clear(), w.R.clear(), w.p.clear(), w.Rf.clear(), w.pf.clear(), w.ids.clear();
    auto put = [](std::vector<double> &dst, const Eigen::Matrix3d &M) {
      Eigen::Matrix<double, 3, 3, Eigen::RowMajor> r = M;
      dst.insert(dst.end(), r.data(), r.data() + 9);
    };
    for (const auto &c : state->clones) {  // std::map: ascending time
      w.t.push_back(c.first);
      put(w.R, c.second->Rot());
      put(w.Rf, c.second->Rot_fej());
      for (int i = 0; i < 3; ++i) w.p.push_back(c.second->pos()(i)), w.pf.push_back(c.second->pos_fej()(i));
      w.ids.push_back(c.second->id());
    }
    plv_state_view &v = w.v;
    v = plv_state_view{};
    v.n_clones = (int)w.t.size();
    v.clone_time = w.t.data(), v.clone_R = w.R.data(), v.clone_p = w.p.data();
    v.clone_R_fej = w.Rf.data(), v.clone_p_fej = w.pf.data(), v.clone_state_id = w.ids.data();
    Eigen::Map<Eigen::Matrix<double, 3, 3, Eigen::RowMajor>>(v.R_ItoC) = state->cam_extrinsic.at(cam_id)->Rot();
    Eigen::Map<Eigen::Vector3d>(v.p_IinC) = state->cam_extrinsic.at(cam_id)->pos();
    Eigen::Map<Eigen::Matrix<double, 8, 1>>(v.intrinsics) = state->cam_intrinsic.at(cam_id)->value();
    v.cam_dt = state->cam_dt.at(cam_id)->value()(0);
    const auto &oc = state->op->cam;
    v.extrinsic_state_id = oc->do_calib_ext ? state->cam_extrinsic.at(cam_id)->id() : -1;
    v.intrinsic_state_id = oc->do_calib_int ? state->cam_intrinsic.at(cam_id)->id() : -1;
    v.dt_state_id = oc->do_calib_dt ? state->cam_dt.at(cam_id)->id() : -1;
    v.intr_order = state->op->intr_order;
    v.dt_exp = state->op->dt_exp;
    v.sigma_pix = oc->sigma_pix;
    v.use_pol_cov = state->op->use_pol_cov ? 1 : 0;
    v.intr_ori_cov = state->op->use_pol_cov ? state->intr_ori_cov(state->op->clone_freq, state->op->intr_order) : 0.0;
    v.intr_pos_cov = state->op->use_pol_cov ? state->intr_pos_cov(state->op->clone_freq, state->op->intr_order) : 0.0;
    v.feat_rep = (int)oc->feat_rep;
  }

  // UpdaterCamera::feed_measurement
  void feed_measurement(const ov_core::CameraData &camdata) {
    t_hist.push_back(camdata.timestamp);
    if (t_hist.size() > 100) t_hist.pop_front();
    trackFEATS->feed_new_camera(camdata);
    if (trackLSDS) {
      auto vps = TrackLSD_HIP::vanishing_points(state->cam_extrinsic.at(0)->Rot(), state->cam_intrinsic.at(0)->value());
      trackLSDS->feed_new_camera(camdata, vps, /*async*/ true);  // joined by plv_camera_update_lines below
    }
  }

  // UpdaterCamera::try_update: point update, dx applied, line update, dx applied.  Returns false when EKFUpdate rejected one.
  bool try_update() {
    if (!state->have_polynomial() || t_hist.size() < 2) return true;
    const auto &oc = state->op->cam;
    plv_update_options o{};
    o.max_msckf = oc->max_msckf, o.max_obs = max_obs, o.chi2_mult = oc->chi2_mult;
    o.tri.min_dist = oc->featinit_options.min_dist, o.tri.max_dist = oc->featinit_options.max_dist;
    o.tri.max_cond_number = oc->featinit_options.max_cond_number, o.tri.max_baseline = oc->featinit_options.max_baseline;
    o.tri.refine_features = oc->featinit_options.refine_features ? 1 : 0;
    o.t_prev_frame = t_hist.at(t_hist.size() - 2), o.state_time = state->time;
    o.window_full = state->clone_window() > state->op->window_size ? 1 : 0;
    o.init_min_meas = 10;
    const int n = (int)state->cov.rows();
    Eigen::VectorXd dx = Eigen::VectorXd::Zero(n);
    plv_update_result r;
    View w;
    bool ok = true;
    build_view(w);
    if (plv_camera_update_points(ctx, &w.v, &o, dx.data(), &r, nullptr, nullptr, nullptr) != PLV_OK) return false;
    if (r.status == PLV_OK && r.n_accepted > 0) StateHelperHIP::apply(ctx, state, dx);
    ok = ok && r.status == PLV_OK;
    if (trackLSDS) {
      build_view(w);  // the clones moved
      if (plv_camera_update_lines(ctx, &w.v, &o, dx.data(), &r, nullptr, nullptr, nullptr, 0x7fffffff) != PLV_OK) return false;
      if (r.status == PLV_OK && r.n_accepted > 0) StateHelperHIP::apply(ctx, state, dx);
      ok = ok && r.status == PLV_OK;
    }
    return ok;
  }

private:
  std::shared_ptr<State> state;
  plv_ctx *ctx;
  std::shared_ptr<ov_core::TrackKLT_HIP> trackFEATS;
  std::shared_ptr<TrackLSD_HIP> trackLSDS;
  int max_obs;
  std::deque<double> t_hist;
};

}  // namespace viw
