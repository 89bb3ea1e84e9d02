// Stand-in for ov_core::CamBase (REF: open_vins/ov_core/src/cam/CamBase.h:49-116)
#pragma once
#include <Eigen/Eigen>
#include "opencv2/core.hpp"
namespace ov_core {
class CamBase {
public:
  virtual ~CamBase();
  virtual void set_value(const Eigen::MatrixXd &calib);
  cv::Point2f undistort_cv(const cv::Point2f &uv_dist);
  Eigen::MatrixXd get_value();                                                                         // :181
  int w();                                                                                             // :190
  int h();                                                                                             // :193
};
}  // namespace ov_core
