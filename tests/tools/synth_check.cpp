// synth_check.cpp -- test helper: writes frames of hip_cascade/synth.hpp to a file so that tests/test_host.py can
// compare them, byte for byte, with ugproj-detector_amd/synth.py.
//   synth_check <face64.u8> <kind> <rows> <cols> <first> <count> <out.u8>
#include <cstdio>
#include <cstdlib>

#include "synth.hpp"

int main(int argc, char** argv) {
  if (argc < 8) return 64;
  const std::vector<uint8_t> face = hip_cascade::synth::load_face_patch(argv[1]);
  const int rows = atoi(argv[3]), cols = atoi(argv[4]), first = atoi(argv[5]), count = atoi(argv[6]);
  std::vector<uint8_t> buf((size_t)rows * cols);
  FILE* f = fopen(argv[7], "wb");
  if (!f) return 1;
  for (int i = 0; i < count; i++) {
    hip_cascade::synth::make_frame(argv[2], rows, cols, (uint64_t)(first + i), face, buf.data());
    if (fwrite(buf.data(), 1, buf.size(), f) != buf.size()) return 1;
  }
  fclose(f);
  return 0;
}
