// Stand-in for ov_core::TrackBase (REF: open_vins/ov_core/src/track/TrackBase.h:72-196)
#pragma once
#include <atomic>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "cam/CamBase.h"
#include "feat/FeatureDatabase.h"
#include "utils/sensor_data.h"
namespace ov_core {
class TrackBase {
public:
  enum HistogramMethod { NONE, HISTOGRAM, CLAHE };                                                     // :78
  TrackBase(std::unordered_map<size_t, std::shared_ptr<CamBase>> cameras, int numfeats, int numaruco, bool stereo,
            HistogramMethod histmethod);                                                               // :88-89
  virtual ~TrackBase();
  virtual void feed_new_camera(const CameraData &message) = 0;                                         // :97
  std::shared_ptr<FeatureDatabase> get_feature_database();                                             // :123
protected:
  std::unordered_map<size_t, std::shared_ptr<CamBase>> camera_calib;                                   // :156
  std::shared_ptr<FeatureDatabase> database;                                                           // :159
  HistogramMethod histogram_method;                                                                    // :171
  std::vector<std::mutex> mtx_feeds;                                                                   // :174
  std::mutex mtx_last_vars;                                                                            // :177
  std::unordered_map<size_t, cv::Mat> img_last, img_mask_last;                                         // :180-183
  std::unordered_map<size_t, std::vector<cv::KeyPoint>> pts_last;                                      // :186
  std::unordered_map<size_t, std::vector<size_t>> ids_last;                                            // :189
  std::atomic<size_t> currid;                                                                          // :192
};
}  // namespace ov_core
