// shim_demo.cpp -- exercises hip_cascade::CascadeClassifier exactly the way the reference's
// tracker uses cv::CascadeClassifier: load() once (src/file_io.cc:11), then
// detectMultiScale(gray, faces, scale, 2, 0|CASCADE_SCALE_IMAGE, Size(30,30)) per frame
// (src/tracker/detector.cc:25-31).  Input: a raw 8-bit gray frame file.
//   shim_demo <cascade.xml|.hcc> <frame.u8> <rows> <cols> [scale=1.1]
// Output: one "x y w h" line per detected face; exit code 0 on success, 1 on load failure
// (the reference's convention), 2 on a detection error.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cascade_classifier.hpp"

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s cascade frame.u8 rows cols [scale]\n", argv[0]); return 64; }
  hip_cascade::CascadeClassifier cascade;
  if (!cascade.load(argv[1])) { fprintf(stderr, "cannot load cascade %s\n", argv[1]); return 1; }
  const int rows = atoi(argv[3]), cols = atoi(argv[4]);
  const double scale = argc > 5 ? atof(argv[5]) : 1.1;
  std::vector<unsigned char> buf((size_t)rows * cols);
  FILE* f = fopen(argv[2], "rb");
  if (!f || fread(buf.data(), 1, buf.size(), f) != buf.size()) { fprintf(stderr, "cannot read frame\n"); return 1; }
  fclose(f);
  hip_cascade::Mat gray(rows, cols, buf.data());
  std::vector<hip_cascade::Rect> facesInGray;
  try {
    cascade.detectMultiScale(gray, facesInGray, scale, 2, 0 | hip_cascade::CASCADE_SCALE_IMAGE, hip_cascade::Size(30, 30));
  } catch (const std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return 2;
  }
  for (const auto& r : facesInGray) printf("%d %d %d %d\n", r.x, r.y, r.width, r.height);
  return 0;
}
