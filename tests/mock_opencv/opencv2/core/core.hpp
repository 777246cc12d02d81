// MOCK -- NOT OpenCV.  A few declarations with the spelling of OpenCV's core types, just enough for
// tests/test_host.py::test_shim_compiles_with_opencv_spelling to SYNTAX-CHECK the `HIP_CASCADE_HAVE_OPENCV` branch of
// ugproj-detector_amd/hip_cascade/cascade_classifier.hpp and face_detector.hpp (the branch a maintainer of the
// reference gets, whose tracker includes <opencv2/opencv.hpp>).  Nothing here computes anything; no product code
// includes it; it is not a stand-in for building the reference.
#pragma once
#include <cstddef>
#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16
namespace cv {
struct Rect { int x = 0, y = 0, width = 0, height = 0; Rect() = default; Rect(int a, int b, int c, int d) : x(a), y(b), width(c), height(d) {} };
struct Size { int width = 0, height = 0; Size() = default; Size(int w, int h) : width(w), height(h) {} };
struct Scalar { double val[4] = {0, 0, 0, 0}; Scalar() = default; Scalar(double a, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; } double operator[](int i) const { return val[i]; } };
struct Mat {
  unsigned char* data = nullptr;
  int rows = 0, cols = 0;
  size_t step = 0;
  int type_ = CV_8UC1;
  int type() const { return type_; }
  int depth() const { return type_ & 7; }
  int channels() const { return (type_ >> 3) + 1; }
};
}  // namespace cv
