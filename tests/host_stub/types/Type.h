// Stand-ins for ov_type::Type, Vec, PoseJPL, IMU, LandmarkRepresentation (REF: open_vins/ov_core/src/types/Type.h, Vec.h,
// PoseJPL.h:74-160, IMU.h:78-160, LandmarkRepresentation.h)
#pragma once
#include <Eigen/Eigen>
namespace ov_type {
class Type {
public:
  virtual ~Type();
  int id();
  int size();
  virtual void update(const Eigen::VectorXd &dx) = 0;
  virtual const Eigen::MatrixXd &value() const;
  virtual void set_value(const Eigen::MatrixXd &new_value);
};
class Vec : public Type {
public:
  void update(const Eigen::VectorXd &dx) override;
};
class PoseJPL : public Type {
public:
  void update(const Eigen::VectorXd &dx) override;
  Eigen::Matrix<double, 3, 3> Rot() const;
  Eigen::Matrix<double, 3, 3> Rot_fej() const;
  Eigen::Matrix<double, 4, 1> quat() const;
  Eigen::Matrix<double, 3, 1> pos() const;
  Eigen::Matrix<double, 3, 1> pos_fej() const;
};
class IMU : public Type {
public:
  void update(const Eigen::VectorXd &dx) override;
  Eigen::Matrix<double, 4, 1> quat() const;
};
struct LandmarkRepresentation {
  enum Representation { GLOBAL_3D, GLOBAL_FULL_INVERSE_DEPTH, ANCHORED_3D, ANCHORED_FULL_INVERSE_DEPTH, ANCHORED_MSCKF_INVERSE_DEPTH, ANCHORED_INVERSE_DEPTH_SINGLE, UNKNOWN };
};
}  // namespace ov_type
