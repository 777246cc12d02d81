// Sanitizer harness (CPU build only; GPU ASan is not available on this pool): exercises the
// product's pure-host code -- cascade loader (HCC1 + XML when reachable), scale planning, resize
// tables, rectangle grouping -- and the CPU oracle end to end, under -fsanitize=address,undefined.
// Built and run by tests/test_sanitize.py.  Exit code 0 and no sanitizer report = pass.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../ugproj-detector_amd/csrc/cascade_model.h"
#include "../../ugproj-detector_amd/csrc/group_rects.h"
#include "../../ugproj-detector_amd/csrc/scale_plan.h"
#include "../../oracle/haar_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
  rng_state += 0x9E3779B97F4A7C15ull;
  uint64_t z = rng_state;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 16);
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s cascade [cascade ...]\n", argv[0]); return 64; }
  for (int a = 1; a < argc; a++) {
    hc::Model m;
    std::string err;
    if (hc::load_model(argv[a], m, err) != 0) { fprintf(stderr, "load %s: %s\n", argv[a], err.c_str()); return 1; }
    const int sizes[][2] = {{720, 1280}, {360, 640}, {1080, 1920}, {97, 131}, {20, 20}, {19, 300}, {2160, 3840}};
    for (const auto& s : sizes) {
      hc::Plan p;
      if (hc::build_plan(m, s[0], s[1], 1.1, 30, 30, 0, 0, 16, 16, p) != 0) return 2;
      size_t words = 0;
      for (const auto& L : p.levels) words += (size_t)L.int_pitch * (L.h + 1) + 32;
      if ((int64_t)words != p.int_words) return 3;
    }
    // oracle end to end on a small noisy frame with a bright square
    orc_cascade* oc = nullptr;
    const std::string path = argv[a];
    const int rc = path.size() > 4 && path.substr(path.size() - 4) == ".hcc" ? orc_cascade_load_hcc(argv[a], &oc)
                                                                            : orc_cascade_load_xml(argv[a], &oc);
    if (rc) return 4;
    std::vector<uint8_t> img(120 * 160);
    for (auto& v : img) v = (uint8_t)(rnd() & 255);
    std::vector<orc_rect> out(1024), raw(1 << 16);
    int n = 0, nr = 0;
    std::vector<int8_t> taps(1 << 20);
    if (orc_detect(oc, img.data(), 120, 160, 160, 1.1, 2, 0, 0, 0, 0, out.data(), 1024, &n, raw.data(), 1 << 16, &nr,
                   taps.data(), taps.size()))
      return 5;
    std::vector<uint8_t> bgr(120 * 160 * 3);
    for (auto& v : bgr) v = (uint8_t)(rnd() & 255);
    const int lo[3] = {0, 133, 77}, hi[3] = {255, 173, 127};
    if (orc_detect_faces(oc, bgr.data(), 120, 160, 480, lo, hi, 1.1, out.data(), 1024, &n, nullptr, nullptr)) return 6;
    orc_cascade_free(oc);
  }
  // malformed HCC1 containers must be rejected by the loader (never indexed out of bounds later):
  // truncations at every 997th byte, a stage with zero / negative / oversized tree counts, a window <= 2
  {
    std::string good;
    if (FILE* f = fopen(argv[1], "rb")) {
      char tmp[1 << 16];
      size_t n;
      while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) good.append(tmp, n);
      fclose(f);
    }
    if (good.size() > 64 && good.compare(0, 4, "HCC1") == 0) {
      const std::string ppath = std::string(argv[0]) + ".poison.hcc";
      auto try_load = [&](const std::string& bytes) {
        FILE* f = fopen(ppath.c_str(), "wb");
        if (!f) return -100;
        fwrite(bytes.data(), 1, bytes.size(), f);
        fclose(f);
        hc::Model m;
        std::string err;
        return hc::load_model(ppath, m, err);
      };
      if (try_load(good) != 0) return 7;
      for (size_t cut = 0; cut < good.size(); cut += 997)
        if (try_load(good.substr(0, cut)) == 0) return 8;
      auto poke = [&](size_t ofs, int32_t v) { std::string b = good; memcpy(&b[ofs], &v, 4); return b; };
      // header: magic, win_w, win_h, nstages, ntrees, nnodes, nleaves, flags; stage record: f64 thr, i32 ntrees, parent, next
      const size_t stage0_ntrees = 4 + 28 + 8, stage1_ntrees = stage0_ntrees + 20;
      int32_t n0, n1;
      memcpy(&n0, &good[stage0_ntrees], 4);
      memcpy(&n1, &good[stage1_ntrees], 4);
      if (try_load(poke(stage0_ntrees, 0)) == 0) return 9;                       // counts no longer add up
      { std::string b = poke(stage0_ntrees, 0); int32_t v = n0 + n1; memcpy(&b[stage1_ntrees], &v, 4);
        if (try_load(b) == 0) return 10; }                                        // sums match, stage 0 is empty
      { std::string b = poke(stage0_ntrees, -3); int32_t v = n0 + n1 + 3; memcpy(&b[stage1_ntrees], &v, 4);
        if (try_load(b) == 0) return 11; }                                        // negative tree count
      if (try_load(poke(4, 2)) == 0) return 12;                                  // win_w = 2: normrect would be empty
      if (try_load(poke(8, 200)) == 0) return 13;                                // window taller than the 8-bit rect fields
      remove(ppath.c_str());
    }
  }
  for (int it = 0; it < 200; it++) {
    std::vector<hc_rect> r;
    const int n = 1 + rnd() % 150;
    for (int i = 0; i < n; i++) {
      const int s = 30 + (rnd() % 4) * 7;
      r.push_back(hc_rect{(int)(rnd() % 300), (int)(rnd() % 200), s, s});
    }
    hc::group_rectangles(r, (int)(rnd() % 4), 0.2);
  }
  std::vector<uint32_t> t;
  for (int dst = 1; dst < 400; dst += 7) hc::build_resize_table(1 + rnd() % 2000, dst, t);
  puts("sanitize OK");
  return 0;
}
