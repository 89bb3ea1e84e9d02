// PlvContext.h — one plv_ctx per camera, configured from the reference's options.
// REF: PL-VIWO/src/update/cam/UpdaterCamera.cpp:30-75 (tracker construction), PL-VIWO/src/options/OptionsCamera.h
#pragma once
#include <cstdlib>
#include <memory>

#include <Eigen/Eigen>

#include "options/OptionsCamera.h"
#include "plviwo.h"
#include "state/State.h"
#include "utils/print.h"

namespace viw {

struct PlvContext {
  plv_ctx *ctx = nullptr;
  PlvContext(const std::shared_ptr<OptionsCamera> &op, const std::shared_ptr<State> &state, int cam_id, int device = 0) {
    plv_config cfg;
    plv_config_default(&cfg, op->wh.at(cam_id).at(0), op->wh.at(cam_id).at(1));  // OptionsCamera.h:45 (width, height)
    cfg.device = device;
    cfg.num_features = op->n_pts;
    cfg.fast_threshold = op->fast;
    cfg.grid_x = op->grid_x;
    cfg.grid_y = op->grid_y;
    cfg.min_px_dist = op->min_px_dist;
    cfg.histogram_method = (int)op->histogram;
    cfg.sigma_pix = op->sigma_pix;
    cfg.chi2_mult = op->chi2_mult;
    Eigen::Map<Eigen::Matrix<double, 8, 1>>(cfg.intrinsics) = state->cam_intrinsic.at(cam_id)->value();
    cfg.max_state_dim = 15 + 45 + 6 * (int)(state->op->window_size * 30 + 3) + 3 * op->max_slam;
    if (plv_ctx_create(&cfg, &ctx) != PLV_OK) {
      PRINT4(RED "[plv] %s\n" RESET, plv_last_error());
      std::exit(EXIT_FAILURE);  // no CPU fallback: the library needs a gfx950 device
    }
  }
  ~PlvContext() {
    if (ctx) plv_ctx_destroy(ctx);
  }
  PlvContext(const PlvContext &) = delete;
  PlvContext &operator=(const PlvContext &) = delete;
};

}  // namespace viw
