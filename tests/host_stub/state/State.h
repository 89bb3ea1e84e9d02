// Stand-in for viw::State: the members the adapters read (REF: PL-VIWO/src/state/State.h:41-287)
#pragma once
#include <map>
#include <memory>
#include <unordered_map>
#include <vector>
#include <Eigen/Eigen>
#include "cam/CamBase.h"
#include "options/OptionsCamera.h"
#include "types/Type.h"
namespace viw {
class State {
public:
  explicit State(std::shared_ptr<OptionsEstimator> op);
  double intr_ori_cov(int hz, int order);                                      // :118
  double intr_pos_cov(int hz, int order);                                      // :119
  bool have_polynomial();                                                      // :124
  void build_polynomial_data(bool fej);                                        // :130
  double clone_window();                                                       // :148
  std::shared_ptr<OptionsEstimator> op;                                        // :163
  double time;                                                                 // :168
  std::shared_ptr<ov_type::IMU> imu;
  std::map<double, std::shared_ptr<ov_type::PoseJPL>> clones;                  // :177
  std::unordered_map<size_t, std::shared_ptr<ov_type::Vec>> cam_dt;            // :189
  std::unordered_map<size_t, std::shared_ptr<ov_type::PoseJPL>> cam_extrinsic; // :192
  std::unordered_map<size_t, std::shared_ptr<ov_type::Vec>> cam_intrinsic;     // :195
  std::unordered_map<size_t, std::shared_ptr<ov_core::CamBase>> cam_intrinsic_model;  // :198
  Eigen::MatrixXd cov;                                                         // :226
  std::vector<std::shared_ptr<ov_type::Type>> variables;                       // :287
};
}  // namespace viw
