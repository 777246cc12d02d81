// shim_demo.cpp -- exercises hip_cascade::CascadeClassifier exactly the way the reference's
// tracker uses cv::CascadeClassifier: load() once (src/file_io.cc:11), then
// detectMultiScale(gray, faces, scale, 2, 0|CASCADE_SCALE_IMAGE, Size(30,30)) per frame
// (src/tracker/detector.cc:25-31).  Input: a raw 8-bit gray frame file.
//   shim_demo <cascade.xml|.hcc> <frame.u8> <rows> <cols> [scale=1.1] [legacy|modern] [channels=1] [minNeighbors=2]
// A sixth argument "legacy" switches the classifier to the OpenCV 2.4.x evaluation (setLegacySemantics); channels 3 / 4
// reads a packed BGR / BGRA frame, which detectMultiScale converts itself as upstream does.
// Output: one "x y w h" line per detected face; exit code 0 on success, 1 on load failure
// (the reference's convention), 2 on a detection error.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <cstring>

#include "cascade_classifier.hpp"
#include "face_detector.hpp"

// shim_demo --faces <cascade> <frame.bgr> <rows> <cols> [scale]: the whole FaceDetector::detectFaces
// (reference src/tracker/detector.cc:10-43) on a raw packed BGR frame, default skin bounds.
static int faces_main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s --faces cascade frame.bgr rows cols [scale]\n", argv[0]); return 64; }
  hip_cascade::CascadeClassifier cascade;
  if (!cascade.load(argv[2])) { fprintf(stderr, "cannot load cascade %s\n", argv[2]); return 1; }
  const int rows = atoi(argv[4]), cols = atoi(argv[5]);
  hip_cascade::DetectionSection cfg;
  if (argc > 6) cfg.scale = atof(argv[6]);
  std::vector<unsigned char> buf((size_t)rows * cols * 3);
  FILE* f = fopen(argv[3], "rb");
  if (!f || fread(buf.data(), 1, buf.size(), f) != buf.size()) { fprintf(stderr, "cannot read frame\n"); return 1; }
  fclose(f);
  hip_cascade::Mat frame(rows, cols, buf.data(), 0, 3);
  hip_cascade::FaceDetector detector(cascade, cfg);
  std::vector<hip_cascade::Rect> rects;
  try {
    detector.detectFaces(frame, rects);
  } catch (const std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return 2;
  }
  for (const auto& r : rects) printf("%d %d %d %d\n", r.x, r.y, r.width, r.height);
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "--faces")) return faces_main(argc, argv);
  if (argc < 5) { fprintf(stderr, "usage: %s cascade frame.u8 rows cols [scale]\n", argv[0]); return 64; }
  hip_cascade::CascadeClassifier cascade;
  if (!cascade.load(argv[1])) { fprintf(stderr, "cannot load cascade %s\n", argv[1]); return 1; }
  const int rows = atoi(argv[3]), cols = atoi(argv[4]);
  const double scale = argc > 5 ? atof(argv[5]) : 1.1;
  if (argc > 6 && !strcmp(argv[6], "legacy")) cascade.setLegacySemantics(true);
  const int channels = argc > 7 ? atoi(argv[7]) : 1, min_neighbors = argc > 8 ? atoi(argv[8]) : 2;
  std::vector<unsigned char> buf((size_t)rows * cols * channels);
  FILE* f = fopen(argv[2], "rb");
  if (!f || fread(buf.data(), 1, buf.size(), f) != buf.size()) { fprintf(stderr, "cannot read frame\n"); return 1; }
  fclose(f);
  hip_cascade::Mat gray(rows, cols, buf.data(), 0, channels);
  std::vector<hip_cascade::Rect> facesInGray;
  try {
    cascade.detectMultiScale(gray, facesInGray, scale, min_neighbors, 0 | hip_cascade::CASCADE_SCALE_IMAGE, hip_cascade::Size(30, 30));
  } catch (const std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return 2;
  }
  for (const auto& r : facesInGray) printf("%d %d %d %d\n", r.x, r.y, r.width, r.height);
  return 0;
}
