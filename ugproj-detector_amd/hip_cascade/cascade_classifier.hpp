// cascade_classifier.hpp -- C++ host-side mirror of the reference's operator interface for
// the hot path: a class with the cv::CascadeClassifier spelling the reference compiles
// against (OpenCV 2.4 generation), backed by the C ABI in
// include/hip_cascade.h.
//
//   reference src/file_io.h:38            cv::CascadeClassifier cascade_;
//   reference src/file_io.cc:11           cascade_.load(path)                  -> load()
//   reference src/tracker/detector.h:17   cv::CascadeClassifier& cascade;
//   reference src/tracker/detector.cc:25  cascade.detectMultiScale(gray, faces, scale, 2,
//                                           0|CASCADE_SCALE_IMAGE, Size(30,30)) -> detectMultiScale()
//
// With OpenCV headers present (the reference's own build), cv::Mat / cv::Rect / cv::Size are
// used directly, so swapping `cv::CascadeClassifier` for `hip_cascade::CascadeClassifier` in
// those two declarations is the whole integration.  Without OpenCV (this
// repository's build and tests) layout-compatible PODs stand in.
//
// Error behaviour mirrors OpenCV: load() returns false on failure; detectMultiScale() throws
// on scaleFactor <= 1 or a non-8-bit image (cv::Exception upstream, which the
// reference never catches: src/tracker/main.cc:14-41) and on any device failure -- there is
// no CPU fallback to hide behind.
#pragma once

#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hip_cascade.h"

#if defined(HIP_CASCADE_WITH_OPENCV) || (defined(__has_include) && __has_include(<opencv2/core/core.hpp>))
#include <opencv2/core/core.hpp>
#define HIP_CASCADE_HAVE_OPENCV 1
#endif

namespace hip_cascade {

#ifdef HIP_CASCADE_HAVE_OPENCV
using Rect = cv::Rect;
using Size = cv::Size;
using Mat = cv::Mat;
inline const unsigned char* mat_data(const Mat& m) { return m.data; }
inline int mat_rows(const Mat& m) { return m.rows; }
inline int mat_cols(const Mat& m) { return m.cols; }
inline size_t mat_step(const Mat& m) { return m.step; }
inline bool mat_is_u8c1(const Mat& m) { return m.type() == CV_8UC1; }
inline bool mat_is_u8c3(const Mat& m) { return m.type() == CV_8UC3; }
inline int mat_u8_channels(const Mat& m) { return m.depth() == CV_8U ? m.channels() : 0; }
using Scalar = cv::Scalar;
#else
struct Rect { int x = 0, y = 0, width = 0, height = 0; };
struct Size {
  int width = 0, height = 0;
  Size() = default;
  Size(int w, int h) : width(w), height(h) {}
};
// minimal stand-in for an 8-bit cv::Mat view (no ownership); channels = 1 (gray) or 3 (BGR)
struct Mat {
  const unsigned char* data = nullptr;
  int rows = 0, cols = 0, nchannels = 1;
  size_t step = 0;
  Mat() = default;
  Mat(int r, int c, const unsigned char* d, size_t s = 0, int ch = 1)
      : data(d), rows(r), cols(c), nchannels(ch), step(s ? s : (size_t)c * ch) {}
};
struct Scalar {
  double val[4] = {0, 0, 0, 0};
  Scalar() = default;
  Scalar(double a, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
  double operator[](int i) const { return val[i]; }
};
inline const unsigned char* mat_data(const Mat& m) { return m.data; }
inline int mat_rows(const Mat& m) { return m.rows; }
inline int mat_cols(const Mat& m) { return m.cols; }
inline size_t mat_step(const Mat& m) { return m.step; }
inline bool mat_is_u8c1(const Mat& m) { return m.nchannels == 1; }
inline bool mat_is_u8c3(const Mat& m) { return m.nchannels == 3; }
inline int mat_u8_channels(const Mat& m) { return m.nchannels; }
#endif
static_assert(sizeof(Rect) == sizeof(hc_rect), "Rect must be layout-compatible with hc_rect");

enum { CASCADE_SCALE_IMAGE = 2 };   // accepted and ignored, as OpenCV >= 4 does for this path

class CascadeClassifier {
 public:
  CascadeClassifier() = default;
  explicit CascadeClassifier(const std::string& filename) { load(filename); }
  CascadeClassifier(const CascadeClassifier&) = delete;
  CascadeClassifier& operator=(const CascadeClassifier&) = delete;
  ~CascadeClassifier() { hc_cascade_free(h_); }

  bool load(const std::string& filename) {
    hc_cascade_free(h_);
    h_ = nullptr;
    return hc_cascade_load(filename.c_str(), &h_) == HC_OK;
  }
  bool empty() const { return h_ == nullptr; }

  // Explicit switch to the OpenCV 2.4.x evaluation the reference was written against (see
  // hc_detect_multiscale_cv24 in include/hip_cascade.h).  Off by default: the backend follows OpenCV >= 4.  With it on,
  // detectMultiScale needs the CASCADE_SCALE_IMAGE flag (the reference passes it, src/tracker/detector.cc:30); a colour
  // image is converted with cvtColor(.., BGR2GRAY) first, as 2.4's detectMultiScale does.
  void setLegacySemantics(bool on) { legacy_ = on; }
  bool legacySemantics() const { return legacy_; }

  void detectMultiScale(const Mat& image, std::vector<Rect>& objects, double scaleFactor = 1.1, int minNeighbors = 3,
                        int flags = 0, Size minSize = Size(), Size maxSize = Size()) {
    objects.clear();                                    // upstream clears the output first
    if (empty()) return;                                // upstream: empty classifier -> no detections
    // upstream: CV_Assert(scaleFactor > 1 && image.depth() == CV_8U); a colour image is converted with
    // cvtColor(image, gray, COLOR_BGR2GRAY) first -- the backend does the same on device
    const int ch = mat_u8_channels(image);
    if (!(scaleFactor > 1.0) || (ch != 1 && ch != 3 && ch != 4) || !mat_data(image))
      throw std::invalid_argument("hip_cascade::detectMultiScale: scaleFactor > 1 and an 8-bit image with 1, 3 or 4 channels required");
    const unsigned char* data = mat_data(image);
    size_t step = mat_step(image);
    int nch = ch;
    std::vector<unsigned char> gray;
    if (legacy_ && ch != 1) {
      // 2.4: cvtColor(image, gray, CV_BGR2GRAY), then the legacy evaluation of the gray plane
      gray.resize((size_t)mat_rows(image) * mat_cols(image));
      if (hc_convert_bgr2gray(data, mat_rows(image), mat_cols(image), ch, step, gray.data(), (size_t)mat_cols(image)) != HC_OK)
        throw std::runtime_error(std::string("hip_cascade::detectMultiScale: ") + hc_last_error());
      data = gray.data();
      step = (size_t)mat_cols(image);
      nch = 1;
    }
    objects.resize(1024);
    int n = 0;
    int rc = legacy_ ? hc_detect_multiscale_cv24(h_, data, mat_rows(image), mat_cols(image), step, scaleFactor, minNeighbors, flags,
                                                 minSize.width, minSize.height, maxSize.width, maxSize.height,
                                                 reinterpret_cast<hc_rect*>(objects.data()), (int)objects.size(), &n)
                     : hc_detect_multiscale_image(h_, data, mat_rows(image), mat_cols(image), nch, step, scaleFactor, minNeighbors,
                                                  flags, minSize.width, minSize.height, maxSize.width, maxSize.height,
                                                  reinterpret_cast<hc_rect*>(objects.data()), (int)objects.size(), &n);
    if (rc == HC_ERR_CAPACITY) {
      // more rectangles than the first buffer holds: the detection is done, the library kept them for this thread
      // (*out_count = how many).  Should they be gone all the same (HC_ERR_STATE), the call simply runs again with a
      // buffer of the size it asked for -- a std::vector grows as needed, the caller never sizes it.
      objects.resize(n);
      rc = hc_detect_multiscale_reread(h_, reinterpret_cast<hc_rect*>(objects.data()), n, &n);
      for (int attempt = 0; (rc == HC_ERR_STATE || rc == HC_ERR_CAPACITY) && attempt < 3; attempt++) {
        objects.resize((size_t)std::max(n, (int)objects.size()) + 256);
        rc = legacy_ ? hc_detect_multiscale_cv24(h_, data, mat_rows(image), mat_cols(image), step, scaleFactor, minNeighbors, flags,
                                                minSize.width, minSize.height, maxSize.width, maxSize.height,
                                                reinterpret_cast<hc_rect*>(objects.data()), (int)objects.size(), &n)
                    : hc_detect_multiscale_image(h_, data, mat_rows(image), mat_cols(image), nch, step, scaleFactor, minNeighbors,
                                                 flags, minSize.width, minSize.height, maxSize.width, maxSize.height,
                                                 reinterpret_cast<hc_rect*>(objects.data()), (int)objects.size(), &n);
      }
    }
    if (rc != HC_OK) {
      objects.clear();
      throw std::runtime_error(std::string("hip_cascade::detectMultiScale: ") + hc_last_error());
    }
    objects.resize(n);
  }

  hc_cascade* handle() const { return h_; }   // for hip_cascade::FaceDetector

 private:
  hc_cascade* h_ = nullptr;
  bool legacy_ = false;
};

}  // namespace hip_cascade
