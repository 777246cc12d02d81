// face_detector.hpp -- C++ mirror of the reference's ugproj::FaceDetector (src/tracker/detector.h:10-19,
// src/tracker/detector.cc:10-43) on top of the C ABI: the whole detectFaces body -- BGR->YCrCb +
// inRange skin mask, BGR->GRAY, equalizeHist, gray &= mask, the cascade call, and the skin-ratio
// post-filter -- runs on the MI355X (hc_detect_faces, include/hip_cascade.h).
//
//   reference                                              here
//   FaceDetector(cv::CascadeClassifier&, const             FaceDetector(hip_cascade::CascadeClassifier&,
//       Configuration::DetectionSection&)                      const DetectionSection&)
//   void detectFaces(const cv::Mat& frame,                 void detectFaces(const Mat& frame,
//       std::vector<cv::Rect>& rects)   [appends]              std::vector<Rect>& rects)   [appends]
#pragma once

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "cascade_classifier.hpp"

namespace hip_cascade {

// the fields of Configuration::DetectionSection (reference src/structure.h:43-52) this path consumes;
// defaults are the reference's (src/structure.cc:75-85)
struct DetectionSection {
  Scalar skin_lower = Scalar(0, 133, 77);
  Scalar skin_upper = Scalar(255, 173, 127);
  double scale = 1.1;
};

class FaceDetector {
 public:
  FaceDetector(CascadeClassifier& cascade, const DetectionSection& cfg) : cascade(cascade), cfg_(cfg) {}

  void detectFaces(const Mat& frame, std::vector<Rect>& rects) {
    if (cascade.empty()) return;
    if (!mat_is_u8c3(frame) || !mat_data(frame))
      throw std::invalid_argument("hip_cascade::FaceDetector::detectFaces: 8-bit 3-channel BGR frame required");
    int lower[3], upper[3];
    // cv::inRange on 8-bit data converts its scalar bounds to integers by ROUNDING (half to even), not by truncation: a
    // configured 133.7 is 134 (identical for the reference's integer defaults, src/structure.cc:75-82)
    for (int k = 0; k < 3; k++) { lower[k] = (int)std::nearbyint(cfg_.skin_lower[k]); upper[k] = (int)std::nearbyint(cfg_.skin_upper[k]); }
    std::vector<hc_rect> buf(1024);
    int n = 0;
    int rc = hc_detect_faces(cascade.handle(), mat_data(frame), mat_rows(frame), mat_cols(frame), mat_step(frame), lower, upper,
                             cfg_.scale, buf.data(), (int)buf.size(), &n);
    if (rc == HC_ERR_CAPACITY) {                  // the library kept the rectangles for this thread: no second detection
      buf.resize(n);
      rc = hc_detect_multiscale_reread(cascade.handle(), buf.data(), n, &n);
      for (int attempt = 0; (rc == HC_ERR_STATE || rc == HC_ERR_CAPACITY) && attempt < 3; attempt++) {   // gone all the same: run again, larger
        buf.resize((size_t)std::max(n, (int)buf.size()) + 256);
        rc = hc_detect_faces(cascade.handle(), mat_data(frame), mat_rows(frame), mat_cols(frame), mat_step(frame), lower, upper,
                             cfg_.scale, buf.data(), (int)buf.size(), &n);
      }
    }
    if (rc != HC_OK) throw std::runtime_error(std::string("hip_cascade::FaceDetector::detectFaces: ") + hc_last_error());
    for (int i = 0; i < n; i++) {
      Rect r;
      r.x = buf[i].x; r.y = buf[i].y; r.width = buf[i].width; r.height = buf[i].height;
      rects.push_back(r);                         // the reference appends (detector.cc:41)
    }
  }

 private:
  CascadeClassifier& cascade;
  const DetectionSection& cfg_;
};

}  // namespace hip_cascade
