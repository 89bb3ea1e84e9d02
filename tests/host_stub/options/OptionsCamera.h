// Stand-ins for viw::OptionsCamera / OptionsEstimator and ov_core::FeatureInitializerOptions: the fields the adapters read
// (REF: PL-VIWO/src/options/OptionsCamera.h:28-120, OptionsEstimator.h:32-117, open_vins/ov_core/src/feat/FeatureInitializerOptions.h:36-69)
#pragma once
#include <map>
#include <memory>
#include <vector>
#include "track/TrackBase.h"
#include "types/Type.h"
namespace ov_core {
struct FeatureInitializerOptions {
  bool triangulate_1d, refine_features;
  int max_runs;
  double init_lamda, max_lamda, min_dx, min_dcost, lam_mult, min_dist, max_dist, max_baseline, max_cond_number;
};
}  // namespace ov_core
namespace viw {
struct OptionsCamera {
  bool enabled;
  int max_n;
  std::map<size_t, std::vector<int>> wh;                                       // :45
  bool do_calib_ext, do_calib_int, do_calib_dt, downsample;
  int n_pts, fast, grid_x, grid_y, min_px_dist;
  ov_core::TrackBase::HistogramMethod histogram;                               // :92
  std::shared_ptr<ov_core::FeatureInitializerOptions> featinit_options;       // :98
  int max_slam, max_msckf;
  bool use_stereo, use_lines;
  ov_type::LandmarkRepresentation::Representation feat_rep;                    // :111
  double chi2_mult, sigma_pix;
};
struct OptionsEstimator {
  std::shared_ptr<OptionsCamera> cam;                                          // :32
  double window_size;
  int clone_freq;
  double dt_exp;
  int intr_order;
  bool use_imu_res, use_pol_cov;
};
}  // namespace viw
