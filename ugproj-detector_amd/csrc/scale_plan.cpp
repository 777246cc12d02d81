// scale_plan.cpp -- host planning for one detectMultiScale configuration.
//
// Follows OpenCV >= 4 CascadeClassifierImpl::detectMultiScaleNoGrouping +
// FeatureEvaluator::updateScaleData as restated in SURVEY.md A.3/A.6b for the call at
// reference src/tracker/detector.cc:25-31 (scaleFactor = cfg.detection.scale, default 1.1
// at reference src/structure.cc:83-85; minSize 30x30; maxSize unset).

#include "scale_plan.h"

#include <algorithm>
#include <cmath>

#include "../../include/hip_cascade.h"

namespace hc {

namespace {
// round-half-to-even, the behaviour of cvRound on x86-64
inline int rne(double v) { return (int)std::nearbyint(v); }
inline int rne(float v) { return (int)std::nearbyintf(v); }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
}  // namespace

void build_resize_table(int src, int dst, std::vector<uint32_t>& out) {
  // cv::resize(INTER_LINEAR_EXACT) coefficient rule: all in IEEE f64,
  // coefficient as 8-bit fixed point, border zones folded in as c1 = 0.
  const double inv_scale = (double)dst / (double)src;
  const double scale = 1.0 / inv_scale;
  out.reserve(out.size() + dst);
  for (int d = 0; d < dst; d++) {
    const double center = scale * ((double)d + 0.5) - 0.5;
    const double fl = std::floor(center);
    uint32_t ofs = 0, c1 = 0;
    if (fl >= 0.0 && src > 1) {
      if (fl < (double)(src - 1)) {
        ofs = (uint32_t)fl;
        c1 = (uint32_t)rne((center - fl) * 256.0);
      } else {
        ofs = (uint32_t)(src - 1);   // right/bottom replicate zone
      }
    }                                // else: left/top replicate zone -> ofs 0, c1 0
    out.push_back((ofs << 9) | c1);
  }
}

void build_area_table(int ssize, int dsize, std::vector<int>& ofs, std::vector<int>& si, std::vector<float>& alpha) {
  // A destination cell [d*scale, (d+1)*scale) takes a leading partial source cell, whole cells, and a trailing
  // partial cell, each weighted by the covered length over the cell width; parts thinner than 1e-3 are dropped.
  const double scale = (double)ssize / (double)dsize;
  ofs.assign(dsize + 1, 0);
  si.clear();
  alpha.clear();
  for (int d = 0; d < dsize; d++) {
    ofs[d] = (int)si.size();
    const double begin = d * scale, end = begin + scale;
    const double width = std::min(scale, ssize - begin);
    int whole_end = std::min((int)std::floor(end), ssize - 1);
    int whole_begin = std::min((int)std::ceil(begin), whole_end);
    if (whole_begin - begin > 1e-3) { si.push_back(whole_begin - 1); alpha.push_back((float)((whole_begin - begin) / width)); }
    for (int s = whole_begin; s < whole_end; s++) { si.push_back(s); alpha.push_back((float)(1.0 / width)); }
    if (end - whole_end > 1e-3) { si.push_back(whole_end); alpha.push_back((float)(std::min(std::min(end - whole_end, 1.), width) / width)); }
  }
  ofs[dsize] = (int)si.size();
}

std::vector<float> scale_ladder(const Model& m, int rows, int cols, double scale_factor, int min_w, int min_h, int max_w,
                                int max_h) {
  std::vector<float> ladder;
  if (!(scale_factor > 1.0) || rows < m.win_h || cols < m.win_w) return ladder;
  if (max_w == 0 || max_h == 0) { max_w = cols; max_h = rows; }
  // every factor whose window fits the image, then the min/max filter
  std::vector<float> fits;
  {
    double factor = 1.0;
    while (rne(m.win_w * factor) <= cols && rne(m.win_h * factor) <= rows) {
      fits.push_back((float)factor);
      factor *= scale_factor;
    }
  }
  for (float s : fits) {
    const int ww = rne((float)m.win_w * s), wh = rne((float)m.win_h * s);
    if (ww > max_w || wh > max_h) break;
    if (ww >= min_w && wh >= min_h) ladder.push_back(s);
  }
  if (ladder.empty() && !fits.empty()) {
    // nearest precomputed scale (minSize == maxSize single-scale use)
    size_t best = 0;
    double best_d = 0;
    for (size_t i = 0; i < fits.size(); i++) {
      const double dw = min_w - rne((float)m.win_w * fits[i]), dh = min_h - rne((float)m.win_h * fits[i]);
      const double d = dw * dw + dh * dh;
      if (i == 0 || d < best_d) { best = i; best_d = d; }
    }
    ladder.push_back(fits[best]);
  }
  return ladder;
}

int build_plan(const Model& m, int rows, int cols, double scale_factor, int min_w, int min_h,
               int max_w, int max_h, int band_rows_ystep1, int band_rows_ystep2, Plan& p,
               const std::vector<float>* prep_scales) {
  p = Plan{};
  p.band_rows[0] = band_rows_ystep1;
  p.band_rows[1] = band_rows_ystep2;
  p.rows = rows; p.cols = cols; p.scale_factor = scale_factor;
  p.min_w = min_w; p.min_h = min_h;
  if (!(scale_factor > 1.0) || rows <= 0 || cols <= 0) return HC_ERR_ARG;
  if (max_w == 0 || max_h == 0) { max_w = cols; max_h = rows; }
  p.max_w = max_w; p.max_h = max_h;
  p.algorithmic_bytes = (int64_t)rows * cols;
  const std::vector<float> ladder = scale_ladder(m, rows, cols, scale_factor, min_w, min_h, max_w, max_h);
  // Levels whose pyramid image and integrals exist: the cascade's own ladder, or -- when several cascades share one
  // prep pass over the same frames -- the union of their ladders (ascending scale).  Level geometry depends on the
  // scale alone, so the integral layout is the same for every cascade of the group; a level outside this cascade's
  // own ladder has an empty grid and is never scanned by it.
  const std::vector<float>& prep = prep_scales ? *prep_scales : ladder;
  if (prep.empty()) return HC_OK;
  for (float s : ladder)
    if (std::find(prep.begin(), prep.end(), s) == prep.end()) return HC_ERR_ARG;

  // ---- level geometry + workspace layout
  int first_scanned = -1;
  for (float sc : prep) {
    const bool scanned = std::find(ladder.begin(), ladder.end(), sc) != ladder.end();
    DevLevel L{};
    L.scale = sc;
    L.w = rne((float)cols / sc);
    L.h = rne((float)rows / sc);
    L.ystep = sc >= 2.f ? 1 : 2;
    L.szw_w = scanned ? std::max(L.w + 1 - m.win_w, 0) : 0;
    L.szw_h = scanned ? std::max(L.h + 1 - m.win_h, 0) : 0;
    L.win_w = rne((float)m.win_w * sc);
    L.win_h = rne((float)m.win_h * sc);
    // Column 0 of an integral is the zero column, so the stores of the integral kernels start at
    // column 1.  A left pad of 31 words puts column 1 of every row on a 128-byte boundary: a
    // wavefront's 64-dword store then covers exactly two cache lines instead of straddling three.
    L.int_pitch = (int)align_up(L.w + 1 + 31, 32);
    L.pyr_ofs = p.pyr_bytes;
    L.int_ofs = p.int_words + 31;
    L.xtab_ofs = (int)p.xtab.size();
    L.ytab_ofs = (int)p.ytab.size();
    build_resize_table(cols, L.w, p.xtab);
    build_resize_table(rows, L.h, p.ytab);
    p.pyr_bytes += align_up((int64_t)L.w * L.h, 64);
    p.int_words += (int64_t)L.int_pitch * (L.h + 1) + 32;
    if (scanned) {
      p.algorithmic_bytes += 16ll * (L.w + 1) * (L.h + 1);
      if (first_scanned < 0) first_scanned = (int)p.levels.size();
    }
    p.scanned.push_back(scanned ? 1 : 0);
    p.levels.push_back(L);
  }
  // stripe rule: the scanned rows are the union of nstripes stripes whose size is
  // derived with truncating divisions -> trailing rows can drop out
  p.nstripes = first_scanned >= 0 ? (int)std::ceil(p.levels[first_scanned].szw_w / 32.0) : 0;
  for (DevLevel& L : p.levels) {
    const int per = std::max(ceil_div(L.szw_h / L.ystep, std::max(p.nstripes, 1)), 1) * L.ystep;
    L.y_end = std::min(p.nstripes * per, L.szw_h);
    L.nx = ceil_div(L.szw_w, L.ystep);
    L.ny = ceil_div(L.y_end, L.ystep);
    L.ny_all = ceil_div(L.szw_h, L.ystep);
    L.grid_ofs = p.grid_elems;
    p.grid_elems += align_up((int64_t)L.nx * L.ny_all, 64);
  }

  // ---- launch work lists (largest levels first == ladder order)
  for (int li = 0; li < (int)p.levels.size(); li++) {
    const DevLevel& L = p.levels[li];
    for (int y0 = 0; y0 < L.h; y0 += 4)
      for (int x0 = 0; x0 < L.w; x0 += 64) p.resize_items.push_back(WorkItem{li, x0, y0});
    for (int y0 = 0; y0 < L.h; y0 += 4) p.row_items.push_back(WorkItem{li, y0, 0});
    for (int x0 = 0; x0 < L.w; x0 += WG_THREADS) p.col_items.push_back(WorkItem{li, x0, 0});
    const int npos = L.nx * L.ny;
    for (int i0 = 0; i0 < npos; i0 += WG_THREADS) p.var_items.push_back(WorkItem{li, i0, 0});
    const int cls = L.ystep == 1 ? 0 : 1;
    if (L.nx > 0)
      for (int gy0 = 0; gy0 < L.ny; gy0 += p.band_rows[cls]) p.band_items[cls].push_back(WorkItem{li, gy0, 0});
  }
  return HC_OK;
}

}  // namespace hc
