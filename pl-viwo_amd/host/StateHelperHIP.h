// StateHelperHIP.h — the dense-algebra half of viw::StateHelper on libplviwo_hip.so.
// REF: PL-VIWO/src/state/StateHelper.h:59-221; StateHelper.cpp:94-173 (EKFUpdate), :602-614 (measurement_compress_inplace),
//      :175-201 (augment_clone), :214-303 (marginalize_old_clone / marginalize), :20-92 (EKFPropagation)
//
// Two ways to hold the covariance:
//  (a) host-resident (state->cov is the truth): EKFUpdate passes it down and gets it back, every call is self-contained;
//  (b) device-resident (the per-frame path): `to_device` once after a host-side change of state->cov, the update / clone /
//      marginalise calls work on the resident copy with P = NULL, `to_host` before host code reads state->cov again.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include <Eigen/Eigen>

#include "plviwo.h"
#include "state/State.h"
#include "types/Type.h"

namespace viw {

class StateHelperHIP {
public:
  using VEC_TYPE = std::vector<std::shared_ptr<ov_type::Type>>;

  static std::vector<int> col_to_state(const VEC_TYPE &H_order) {  // flat form of H_order (id(), size())
    std::vector<int> c;
    for (const auto &v : H_order)
      for (int d = 0; d < v->size(); ++d) c.push_back(v->id() + d);
    return c;
  }

  // x <- x [+] dx for every variable, then the bookkeeping of StateHelper.cpp:156-171
  static void apply(plv_ctx *ctx, std::shared_ptr<State> state, const Eigen::VectorXd &dx) {
    for (auto &var : state->variables) var->update(dx.block(var->id(), 0, var->size(), 1));
    if (state->op->cam->enabled && state->op->cam->do_calib_int) {
      for (auto &calib : state->cam_intrinsic) state->cam_intrinsic_model.at(calib.first)->set_value(calib.second->value());
      plv_set_camera_intrinsics(ctx, state->cam_intrinsic.at(0)->value().data());  // the tracker's undistortion follows
    }
    if (!state->op->use_imu_res) state->build_polynomial_data(false);  // StateHelper.cpp:171 (KAIST sets use_imu_res: the polynomial is not rebuilt)
  }

  // StateHelper::EKFUpdate.  The camera path passes R = I (UpdaterCamera.cpp:290), the GPS path a diagonal, the wheel path a DENSE 6 x 6 /
  // 3 x 3 preintegration covariance (UpdaterWheel.cpp:130-134).  plv_ekf_update takes a diagonal; a dense R is applied by whitening
  // with its Cholesky factor, R = L L^T: H <- L^-1 H, res <- L^-1 res, R <- I — the same K res and the same K M^T (what
  // plv_wheel_update does inside its kernel).  resident = covariance mode (b).
  static bool EKFUpdate(plv_ctx *ctx, std::shared_ptr<State> state, const VEC_TYPE &H_order, const Eigen::MatrixXd &H,
                        const Eigen::VectorXd &res, const Eigen::MatrixXd &R, bool resident = false) {
    const std::vector<int> cols = col_to_state(H_order);
    const int n = (int)state->cov.rows();
    Eigen::VectorXd dx = Eigen::VectorXd::Zero(n);
    Eigen::MatrixXd Hw = H;
    Eigen::VectorXd rw = res, Rdiag;
    const double *rd = nullptr;
    if (R.isDiagonal()) {
      Rdiag = R.diagonal();
      rd = Rdiag.data();
    } else {
      Eigen::LLT<Eigen::MatrixXd> llt(R);
      if (llt.info() != Eigen::Success) return false;  // (a noise matrix that is not positive definite)
      Hw = llt.matrixL().solve(H);
      rw = llt.matrixL().solve(res);
    }
    const int rc = plv_ekf_update(ctx, resident ? nullptr : state->cov.data(), n, (int)state->cov.outerStride(), Hw.data(), (int)Hw.rows(),
                                  (int)Hw.cols(), (int)Hw.outerStride(), cols.data(), rw.data(), rd, dx.data());
    if (rc != PLV_OK) return false;  // PLV_E_NOT_PSD: nothing was modified (StateHelper.cpp:143-152)
    apply(ctx, state, dx);
    return true;
  }

  // StateHelper::measurement_compress_inplace
  static void measurement_compress_inplace(plv_ctx *ctx, Eigen::MatrixXd &H_x, Eigen::VectorXd &res) {
    if (H_x.rows() <= H_x.cols()) return;  // StateHelper.cpp:604-606
    int m_out = 0;
    plv_compress(ctx, H_x.data(), (int)H_x.rows(), (int)H_x.cols(), (int)H_x.outerStride(), res.data(), &m_out);
    H_x.conservativeResize(m_out, H_x.cols());
    res.conservativeResize(m_out);
  }

  // covariance hand-over for mode (b)
  static bool to_device(plv_ctx *ctx, const std::shared_ptr<State> &state) {
    return plv_cov_upload(ctx, state->cov.data(), (int)state->cov.rows(), (int)state->cov.outerStride()) == PLV_OK;
  }
  static bool to_host(plv_ctx *ctx, std::shared_ptr<State> &state) {
    return plv_cov_download(ctx, state->cov.data(), (int)state->cov.rows(), (int)state->cov.outerStride()) == PLV_OK;
  }
  // the covariance halves of augment_clone / marginalize in mode (b); the Type objects and their ids stay with the caller's code
  static bool clone_cov(plv_ctx *ctx, int n, int src_id, int size) { return plv_cov_clone(ctx, n, src_id, size) == PLV_OK; }
  static bool marginalize_cov(plv_ctx *ctx, int id, int size) { return plv_cov_marginalize(ctx, id, size) == PLV_OK; }
};

}  // namespace viw
