// Stand-in for ov_core::CameraData (REF: open_vins/ov_core/src/utils/sensor_data.h:55-79)
#pragma once
#include <vector>
#include "opencv2/core.hpp"
namespace ov_core {
struct CameraData {
  double timestamp;
  std::vector<int> sensor_ids;
  std::vector<cv::Mat> images;
  std::vector<cv::Mat> masks;
};
}  // namespace ov_core
