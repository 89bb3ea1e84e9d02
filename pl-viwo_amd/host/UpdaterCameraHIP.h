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
  // REF: UpdaterCamera.cpp:26 — the reference's signature `UpdaterCamera(shared_ptr<State> state)`: builds the camera's trackers with the
  // reference's own constructor arguments (:41,44; TrackKLT_HIP creates and owns the plv_ctx)
  explicit UpdaterCameraHIP(std::shared_ptr<State> state_) : state(state_), ctx(nullptr), max_obs(24) {
    const std::shared_ptr<OptionsCamera> op = state->op->cam;
    trackFEATS = std::make_shared<ov_core::TrackKLT_HIP>(state->cam_intrinsic_model, op->n_pts, 0, op->use_stereo, op->histogram, op->fast, op->grid_x,
                                                         op->grid_y, op->min_px_dist);
    trackFEATS->mirror_database(false);  // (this class reads the library's track store)
    ctx = trackFEATS->context();
    if (op->use_lines) trackLSDS = std::make_shared<TrackLSD_HIP>(ctx);
  }
  UpdaterCameraHIP(std::shared_ptr<State> state_, plv_ctx *ctx_, std::shared_ptr<ov_core::TrackKLT_HIP> klt, std::shared_ptr<TrackLSD_HIP> lsd,
                   int max_obs_ = 24)
      : state(state_), ctx(ctx_), trackFEATS(klt), trackLSDS(lsd), max_obs(max_obs_) {}

  // A cv::Mat header over one of the library's page-locked image blocks (plv_image_buffer, index 0..3): where the ROS callback puts
  // its copy of the message (`cv_bridge::toCvShare(msg)->image.copyTo(up_cam->image_block(k))` instead of `.clone()`,
  // REF: ROSSubscriber's camera callbacks) or a camera driver its frame.  frame() / feed_measurement() given such an image read it from
  // where it lies: no host copy inside the call, the pixels cross PCIe in the frame's first kernel.  Alternate the index frame by frame.
  cv::Mat image_block(int index) {
    uint8_t *ptr = nullptr;
    int stride = 0;
    if (plv_image_buffer(ctx, index, &ptr, &stride) != PLV_OK) return cv::Mat();
    const std::vector<int> &wh = state->op->cam->wh.at(0);
    return cv::Mat(wh.at(1), wh.at(0), CV_8UC1, ptr, (size_t)stride);
  }

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
    o.tri.min_dist = oc->featinit_options->min_dist, o.tri.max_dist = oc->featinit_options->max_dist;
    o.tri.max_cond_number = oc->featinit_options->max_cond_number, o.tri.max_baseline = oc->featinit_options->max_baseline;
    o.tri.refine_features = oc->featinit_options->refine_features ? 1 : 0;
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

  // ---- feed_measurement + try_update as ONE library call (plv_camera_frame; no in-state landmarks).  The mean of the state lives
  // in flat arrays that also back the plv_state_view: plv_state_boxplus moves them in place (the arithmetic of Type::update), the view
  // the line half reads is current without being rebuilt, and write_back() puts the values into the reference's Type objects.
  struct Mean {
    View w;                         // clone R / p arrays = `out` / `val` of the clone variables
    std::vector<double> q;          // clone quaternions [n][4] (JPL), then the IMU's and the extrinsics'
    std::vector<double> vec;        // IMU p v bg ba (12), extrinsic p (3), intrinsics (8), dt (1)
    std::vector<plv_state_var> vars;
  };
  void build_mean(Mean &m) const {
    build_view(m.w);
    const int n = m.w.v.n_clones, cam_id = 0;
    m.q.assign(4 * (size_t)(n + 2), 0.0);
    m.vec.assign(24, 0.0);
    m.vars.clear();
    int i = 0;
    for (const auto &c : state->clones) {
      Eigen::Map<Eigen::Vector4d>(&m.q[4 * i]) = c.second->quat();
      m.vars.push_back({PLV_VAR_QUAT, c.second->id(), 4, &m.q[4 * i], &m.w.R[9 * i], nullptr});
      m.vars.push_back({PLV_VAR_VEC, c.second->id() + 3, 3, &m.w.p[3 * i], nullptr, nullptr});
      ++i;
    }
    auto imu = state->imu;
    Eigen::Map<Eigen::Vector4d>(&m.q[4 * n]) = imu->quat();
    Eigen::Map<Eigen::Matrix<double, 12, 1>>(&m.vec[0]) = imu->value().block(4, 0, 12, 1);
    m.vars.push_back({PLV_VAR_QUAT, imu->id(), 4, &m.q[4 * n], nullptr, nullptr});
    m.vars.push_back({PLV_VAR_VEC, imu->id() + 3, 12, &m.vec[0], nullptr, nullptr});
    plv_state_view &v = m.w.v;
    if (v.extrinsic_state_id >= 0) {
      Eigen::Map<Eigen::Vector4d>(&m.q[4 * (n + 1)]) = state->cam_extrinsic.at(cam_id)->quat();
      Eigen::Map<Eigen::Vector3d>(&m.vec[12]) = state->cam_extrinsic.at(cam_id)->pos();
      m.vars.push_back({PLV_VAR_QUAT, v.extrinsic_state_id, 4, &m.q[4 * (n + 1)], nullptr, v.R_ItoC});
      m.vars.push_back({PLV_VAR_VEC, v.extrinsic_state_id + 3, 3, &m.vec[12], nullptr, v.p_IinC});
    }
    if (v.intrinsic_state_id >= 0) {
      Eigen::Map<Eigen::Matrix<double, 8, 1>>(&m.vec[15]) = state->cam_intrinsic.at(cam_id)->value();
      m.vars.push_back({PLV_VAR_VEC, v.intrinsic_state_id, 8, &m.vec[15], nullptr, v.intrinsics});
    }
    if (v.dt_state_id >= 0) {
      m.vec[23] = v.cam_dt;
      m.vars.push_back({PLV_VAR_VEC, v.dt_state_id, 1, &m.vec[23], nullptr, &v.cam_dt});
    }
    // (wheel calibration variables, when estimated, are listed the same way)
  }
  void write_back(const Mean &m) {  // Type::set_value for every variable the updates moved (StateHelper.cpp:156-160)
    const int n = m.w.v.n_clones, cam_id = 0;
    int i = 0;
    for (auto &c : state->clones) {
      Eigen::Matrix<double, 7, 1> x;
      x << Eigen::Map<const Eigen::Vector4d>(&m.q[4 * i]), Eigen::Map<const Eigen::Vector3d>(&m.w.p[3 * i]);
      c.second->set_value(x);
      ++i;
    }
    Eigen::Matrix<double, 16, 1> xi;
    xi << Eigen::Map<const Eigen::Vector4d>(&m.q[4 * n]), Eigen::Map<const Eigen::Matrix<double, 12, 1>>(&m.vec[0]);
    state->imu->set_value(xi);
    const plv_state_view &v = m.w.v;
    if (v.extrinsic_state_id >= 0) {
      Eigen::Matrix<double, 7, 1> x;
      x << Eigen::Map<const Eigen::Vector4d>(&m.q[4 * (n + 1)]), Eigen::Map<const Eigen::Vector3d>(&m.vec[12]);
      state->cam_extrinsic.at(cam_id)->set_value(x);
    }
    if (v.intrinsic_state_id >= 0) state->cam_intrinsic.at(cam_id)->set_value(Eigen::Map<const Eigen::Matrix<double, 8, 1>>(&m.vec[15]));
    if (v.dt_state_id >= 0) state->cam_dt.at(cam_id)->set_value(Eigen::Matrix<double, 1, 1>(m.vec[23]));
  }
  // returns false when EKFUpdate rejected one of the two updates (the state is then as the other one left it)
  bool frame(const ov_core::CameraData &camdata) {
    t_hist.push_back(camdata.timestamp);
    if (t_hist.size() > 100) t_hist.pop_front();
    const auto &oc = state->op->cam;
    plv_update_options o{};
    o.max_msckf = oc->max_msckf, o.max_obs = max_obs, o.chi2_mult = oc->chi2_mult;
    o.tri.min_dist = oc->featinit_options->min_dist, o.tri.max_dist = oc->featinit_options->max_dist;
    o.tri.max_cond_number = oc->featinit_options->max_cond_number, o.tri.max_baseline = oc->featinit_options->max_baseline;
    o.tri.refine_features = oc->featinit_options->refine_features ? 1 : 0;
    o.state_time = state->time;
    o.window_full = state->clone_window() > state->op->window_size ? 1 : 0;
    o.init_min_meas = 10;
    const bool update = state->have_polynomial() && t_hist.size() >= 2;
    if (update) o.t_prev_frame = t_hist.at(t_hist.size() - 2);
    const int n = (int)state->cov.rows();
    Eigen::VectorXd dx_p = Eigen::VectorXd::Zero(n), dx_l = Eigen::VectorXd::Zero(n);
    plv_update_result rp{}, rl{};
    Mean m;
    build_mean(m);
    plv_try_update up{&o, trackLSDS ? &o : nullptr, (int)m.vars.size(), m.vars.data(), dx_p.data(), dx_l.data(), &rp, &rl,
                      nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0x7fffffff, 0};
    const cv::Mat &img = camdata.images.at(0), &mask = camdata.masks.at(0);
    plv_camera_frame_io io{camdata.timestamp, -1, img.data, (int)img.step, mask.empty() ? nullptr : mask.data, trackLSDS ? 1 : 0,
                           update ? &up : nullptr, 0};
    if (plv_camera_frame(ctx, &m.w.v, &io) != PLV_OK) return false;
    if (update) write_back(m);
    return rp.status == PLV_OK && rl.status == PLV_OK;
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
