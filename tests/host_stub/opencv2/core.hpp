// Stand-in for the OpenCV types the adapters touch (cv::Mat, cv::Point2f, cv::KeyPoint), declarations only.
#pragma once
#include <cstddef>
#define CV_8UC1 0
namespace cv {
struct Point2f {
  float x, y;
  Point2f();
  Point2f(float, float);
};
struct KeyPoint {
  Point2f pt;
  float size, angle, response;
};
struct MatStep {
  operator size_t() const;
};
class Mat {
public:
  Mat();
  Mat(int rows, int cols, int type, void *data, size_t step = 0);   // (header over caller-owned memory)
  void copyTo(Mat &dst) const;
  int type() const;
  bool isContinuous() const;
  bool empty() const;
  unsigned char *data;
  MatStep step;
  int rows, cols;
};
}  // namespace cv
