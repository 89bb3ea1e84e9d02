// Stand-in for ov_core::FeatureDatabase (REF: open_vins/ov_core/src/feat/FeatureDatabase.h:91)
#pragma once
#include <cstddef>
namespace ov_core {
class FeatureDatabase {
public:
  void update_feature(size_t id, double timestamp, size_t cam_id, float u, float v, float u_n, float v_n);
};
}  // namespace ov_core
