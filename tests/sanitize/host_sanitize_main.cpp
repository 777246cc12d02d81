// Sanitizer harness (CPU build only; GPU ASan is not available on this pool): exercises the
// product's pure-host code -- cascade loader (HCC1 + XML when reachable), scale planning, resize
// tables, rectangle grouping -- and the CPU oracle end to end, under -fsanitize=address,undefined.
// Built and run by tests/test_sanitize.py.  Exit code 0 and no sanitizer report = pass.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../ugproj-detector_amd/csrc/cascade_model.h"
#include "../../ugproj-detector_amd/csrc/group_rects.h"
#include "../../ugproj-detector_amd/csrc/scale_plan.h"
#include "../../oracle/haar_oracle.h"
#include "../../ugproj-detector_amd/hip_cascade/box_records.hpp"

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
  // shared prep pass (hc_context_configure_shared): a plan laid out over the union of two cascades' ladders scans
  // exactly the levels, grids and stripes of the cascade's own plan; the other levels have empty grids
  if (argc >= 4) {
    hc::Model a, b;
    std::string err;
    if (hc::load_model(argv[2], a, err) || hc::load_model(argv[3], b, err)) return 30;      // alt2 (win 20), default (win 24)
    const int geo[][2] = {{1080, 1920}, {720, 1280}, {97, 131}, {23, 40}};
    for (const auto& g : geo) {
      std::vector<float> la = hc::scale_ladder(a, g[0], g[1], 1.1, 30, 30, 0, 0), lb = hc::scale_ladder(b, g[0], g[1], 1.1, 30, 30, 0, 0);
      std::vector<float> uni = la;
      for (float s : lb) if (std::find(uni.begin(), uni.end(), s) == uni.end()) uni.push_back(s);
      std::sort(uni.begin(), uni.end());
      const hc::Model* ms[2] = {&a, &b};
      for (int k = 0; k < 2; k++) {
        hc::Plan own, sh;
        if (hc::build_plan(*ms[k], g[0], g[1], 1.1, 30, 30, 0, 0, 16, 16, own)) return 31;
        if (hc::build_plan(*ms[k], g[0], g[1], 1.1, 30, 30, 0, 0, 16, 16, sh, &uni)) return 32;
        if (sh.levels.size() != uni.size() || sh.algorithmic_bytes != own.algorithmic_bytes || sh.nstripes != own.nstripes) return 33;
        size_t j = 0, bands_own = own.band_items[0].size() + own.band_items[1].size();
        for (size_t i = 0; i < sh.levels.size(); i++) {
          const hc::DevLevel& L = sh.levels[i];
          if (!sh.scanned[i]) { if (L.nx || L.ny || L.ny_all) return 34; continue; }
          const hc::DevLevel& O = own.levels[j++];
          if (L.scale != O.scale || L.w != O.w || L.h != O.h || L.nx != O.nx || L.ny != O.ny || L.y_end != O.y_end ||
              L.win_w != O.win_w || L.ystep != O.ystep || L.int_pitch != O.int_pitch)
            return 35;
        }
        if (j != own.levels.size() || sh.band_items[0].size() + sh.band_items[1].size() != bands_own) return 36;
      }
      if (g[0] == 1080 && (uni.size() != 39 || la.size() != 37 || lb.size() != 37)) return 37;   // k = 3..41: 5..41 and 3..39
    }
    std::vector<float> bad = {1.5f};
    hc::Plan p;
    if (hc::build_plan(a, 720, 1280, 1.1, 30, 30, 0, 0, 16, 16, p, &bad) == 0) return 38;       // not a superset: refused
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
  // box records of the multi-GPU gather (hip_cascade/box_records.hpp): 3 "ranks" x 5 records, frames with 0, a few,
  // exactly 64 and far more than 64 boxes, padding records, ranks that overflow by different amounts
  {
    const int world = 3, per = 5;
    std::vector<int32_t> gathered;
    std::vector<std::vector<hc_rect>> blocks(world);
    std::map<int32_t, std::vector<hc_rect>> want;
    size_t stride = 0;
    for (int r = 0; r < world; r++) {
      std::vector<int32_t> ids(per), ofs(per + 1, 0), rec;
      std::vector<hc_rect> rects;
      for (int i = 0; i < per; i++) {
        const int fid = r * per + i;
        ids[i] = (r == 2 && i == per - 1) ? -1 : fid;                       // one padding record
        const int n = ids[i] < 0 ? 0 : (fid % 4 == 0 ? 64 + 9 * fid : (fid % 4 == 1 ? 64 : fid % 7));
        for (int k = 0; k < n; k++) rects.push_back(hc_rect{fid, k, 30 + k % 5, 30 + k % 5});
        ofs[i + 1] = (int32_t)rects.size();
        if (ids[i] >= 0) want[fid] = std::vector<hc_rect>(rects.begin() + ofs[i], rects.end());
      }
      hip_cascade::pack_records(ids.data(), per, ofs.data(), rects.data(), rec, blocks[r]);
      gathered.insert(gathered.end(), rec.begin(), rec.end());
      stride = std::max(stride, blocks[r].size());
    }
    const std::vector<int64_t> over = hip_cascade::overflow_counts(gathered.data(), world, per);
    std::vector<hc_rect> ov((size_t)world * stride, hc_rect{-7, -7, -7, -7});
    for (int r = 0; r < world; r++) {
      if ((size_t)over[r] != blocks[r].size()) return 20;
      std::copy(blocks[r].begin(), blocks[r].end(), ov.begin() + (size_t)r * stride);
    }
    if (stride == 0) return 21;
    auto got = hip_cascade::unpack_records(gathered.data(), world, per, ov.data(), stride);
    if (got.size() != want.size()) return 22;
    for (auto& kv : want) {
      const auto& g = got[kv.first];
      if (g.size() != kv.second.size()) return 23;
      for (size_t k = 0; k < g.size(); k++)
        if (g[k].x != kv.second[k].x || g[k].y != kv.second[k].y || g[k].width != kv.second[k].width) return 24;
    }
    bool threw = false;
    try { hip_cascade::unpack_records(gathered.data(), world, per, ov.data(), stride - 1); } catch (const std::runtime_error&) { threw = true; }
    if (!threw) return 25;                                                  // a short block is an error, never a silent loss
  }
  std::vector<uint32_t> t;
  for (int dst = 1; dst < 400; dst += 7) hc::build_resize_table(1 + rnd() % 2000, dst, t);
  puts("sanitize OK");
  return 0;
}
