// legacy_cv24.cpp -- host side of the OpenCV-2.4 ("T2") evaluation mode: the semantics the
// reference was literally built against (src/tracker/associator.cc:3 pins OpenCV 2.4) for the call at
// src/tracker/detector.cc:25-31 with its CASCADE_SCALE_IMAGE flag -- cvHaarDetectObjectsForROC, scale-image branch,
// scalar C path, the cascade's stage tree honoured (data/haarcascade_frontalface_alt_tree.xml:1303-1304).
// An explicit switch: hc_detect_multiscale_cv24.  Synchronous, one frame, workspace allocated per call: this mode
// exists for parity with a 2.4 build, not for throughput.  Parity unpinned (no OpenCV reachable).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "group_rects.h"
#include "hc_internal.h"
#include "kernels.h"
#include "legacy_cv24.h"

namespace {

using hc::set_error;

inline int rne(double v) { return (int)std::nearbyint(v); }

struct Level24 { double factor; int w, h, ystep, nx, ny, win_w, win_h; };

// B.1 / B.4: the scale loop of cvHaarDetectObjectsForROC (factor stays f64; the last row and column of valid
// window origins are not scanned)
std::vector<Level24> plan24(const hc::Model& m, int rows, int cols, double sf, int min_w, int min_h, int max_w, int max_h) {
  std::vector<Level24> out;
  if (max_w == 0 || max_h == 0) { max_w = cols; max_h = rows; }
  for (double factor = 1;; factor *= sf) {
    const int ww = rne(m.win_w * factor), wh = rne(m.win_h * factor);
    const int w = rne(cols / factor), h = rne(rows / factor);
    if (w - m.win_w + 1 <= 0 || h - m.win_h + 1 <= 0) break;
    if (ww > max_w || wh > max_h) break;
    if (ww < min_w || wh < min_h) continue;
    const int ystep = factor > 2 ? 1 : 2;
    out.push_back(Level24{factor, w, h, ystep, (w - m.win_w + ystep - 1) / ystep, (h - m.win_h + ystep - 1) / ystep, ww, wh});
  }
  return out;
}

// B.2: coefficient table of one axis of cv::resize(INTER_LINEAR), 8-bit
void linear_table24(int ssize, int dsize, std::vector<int>& ofs, std::vector<short>& a0, std::vector<short>& a1) {
  const double scale = 1. / ((double)dsize / ssize);
  for (int d = 0; d < dsize; d++) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int sidx = (int)std::floor(f);
    f -= (float)sidx;
    if (sidx < 0) { f = 0.f; sidx = 0; }
    if (sidx >= ssize - 1) { f = 0.f; sidx = ssize - 1; }
    ofs.push_back(sidx);
    a0.push_back((short)std::nearbyintf((1.f - f) * 2048.f));
    a1.push_back((short)std::nearbyintf(f * 2048.f));
  }
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess ? 0 : -1; }
  int upload(const std::vector<T>& v) {
    if (alloc(v.size())) return -1;
    return v.empty() || hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
  }
};

int detect24(hc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step, double sf, int min_neighbors, int min_w,
             int min_h, int max_w, int max_h, std::vector<hc_rect>& boxes, int8_t* taps, size_t taps_cap) {
  const hc::Model& m = c->model;
  const std::vector<Level24> lv = plan24(m, rows, cols, sf, min_w, min_h, max_w, max_h);
  boxes.clear();
  if (lv.empty()) return HC_OK;
  // ---- hidden cascade (icvCreateHidHaarClassifierCascade + cvSetImagesForHaarClassifierCascade at scale 1): B.6 - B.8
  const double inv_area = 1. / ((double)(m.win_w - 2) * (m.win_h - 2));
  std::vector<hc::Cv24Node> nodes(m.nodes.size());
  for (size_t i = 0; i < m.nodes.size(); i++) {
    const hc::ModelNode& n = m.nodes[i];
    hc::Cv24Node d{};
    double sum0 = 0, area0 = 0;
    for (int k = 0; k < 3; k++) {
      for (int j = 0; j < 4; j++) d.rect[k][j] = (int8_t)n.rect[k][j];
      if (k > 0 && n.weight[k] == 0.f) { d.w[k] = 0.f; continue; }
      d.w[k] = (float)(n.weight[k] * inv_area);
      if (k == 0) area0 = n.rect[k][2] * n.rect[k][3];
      else sum0 += d.w[k] * n.rect[k][2] * n.rect[k][3];
    }
    d.w[0] = (float)(-sum0 / area0);
    d.thr = n.thr; d.left = n.left; d.right = n.right; d.has3 = n.weight[2] != 0.f;
    nodes[i] = d;
  }
  std::vector<hc::Cv24Tree> trees;
  for (const hc::ModelTree& t : m.trees) trees.push_back(hc::Cv24Tree{t.first_node, t.first_leaf});
  std::vector<hc::Cv24Stage> stages;
  const int nst = (int)m.stages.size();
  for (const hc::ModelStage& s : m.stages) {
    const float bias = 0.0001f;
    const int parent = (s.parent >= 0 && s.parent < nst) ? s.parent : -1, next = (s.next >= 0 && s.next < nst) ? s.next : -1;
    stages.push_back(hc::Cv24Stage{s.first_tree, s.ntrees, (float)s.thr_raw - bias, parent, next, -1});
  }
  for (int i = 0; i < nst; i++)
    if (stages[i].parent >= 0 && stages[stages[i].parent].child < 0) stages[stages[i].parent].child = i;
  // a malformed parent / next graph must not hang the device: every walk below visits each stage at most once if
  // children and next-siblings always have larger indices than where they are reached from
  for (int i = 0; i < nst; i++)
    if ((stages[i].child >= 0 && stages[i].child <= i) || (stages[i].next >= 0 && stages[i].next <= i) || stages[i].parent >= i)
      return set_error(HC_ERR_UNSUPPORTED, "stage tree of the cascade is not in forward order");

  // ---- per-level layout + resize tables
  std::vector<hc::Cv24DevLevel> dl;
  std::vector<hc::DevLevel> il;                      // the shape k_integral_rows / k_integral_cols expect
  std::vector<hc::WorkItem> row_items, col_items;
  std::vector<int> tab_ofs;
  std::vector<short> tab_a0, tab_a1;
  int64_t pyr_bytes = 0, int_words = 0, grid = 0;
  int max_pix = 0, max_pos = 0;
  for (size_t li = 0; li < lv.size(); li++) {
    const Level24& L = lv[li];
    hc::Cv24DevLevel d{};
    d.w = L.w; d.h = L.h; d.ystep = L.ystep; d.nx = L.nx; d.ny = L.ny;
    d.int_pitch = (L.w + 1 + 31 + 31) / 32 * 32;
    d.resize_mode = (L.w == cols && L.h == rows) ? 0 : ((cols == 2 * L.w && rows == 2 * L.h) ? 2 : 1);
    d.xtab = (int)tab_ofs.size();
    linear_table24(cols, L.w, tab_ofs, tab_a0, tab_a1);
    d.ytab = (int)tab_ofs.size();
    linear_table24(rows, L.h, tab_ofs, tab_a0, tab_a1);
    d.pyr_ofs = pyr_bytes; d.int_ofs = int_words + 31; d.grid_ofs = grid;
    pyr_bytes += ((int64_t)L.w * L.h + 63) / 64 * 64;
    int_words += (int64_t)d.int_pitch * (L.h + 1) + 32;
    grid += (int64_t)L.nx * L.ny;
    max_pix = std::max(max_pix, L.w * L.h);
    max_pos = std::max(max_pos, L.nx * L.ny);
    dl.push_back(d);
    hc::DevLevel I{};
    I.w = L.w; I.h = L.h; I.int_pitch = d.int_pitch; I.pyr_ofs = d.pyr_ofs; I.int_ofs = d.int_ofs;
    il.push_back(I);
    for (int y0 = 0; y0 < L.h; y0 += 4) row_items.push_back(hc::WorkItem{(int)li, y0, 0});
    for (int x0 = 0; x0 < L.w; x0 += hc::WG_THREADS) col_items.push_back(hc::WorkItem{(int)li, x0, 0});
  }
  if (taps && (size_t)grid > taps_cap) return set_error(HC_ERR_CAPACITY, "tap buffer too small");

  DevBuf<uint8_t> d_img, d_pyr;
  DevBuf<uint32_t> d_sum, d_sq;
  DevBuf<hc::Cv24DevLevel> d_dl;
  DevBuf<hc::DevLevel> d_il;
  DevBuf<hc::WorkItem> d_rows, d_cols;
  DevBuf<int> d_tofs;
  DevBuf<short> d_ta0, d_ta1;
  DevBuf<hc::Cv24Node> d_nodes;
  DevBuf<hc::Cv24Tree> d_trees;
  DevBuf<hc::Cv24Stage> d_stages;
  DevBuf<float> d_leaves;
  DevBuf<unsigned int> d_count;
  DevBuf<int8_t> d_taps;
  size_t cand_cap = std::max<size_t>(1 << 16, (size_t)grid / 64);
  if (d_img.alloc((size_t)rows * cols) || d_pyr.alloc(pyr_bytes) || d_sum.alloc(int_words) || d_sq.alloc(int_words) ||
      d_dl.upload(dl) || d_il.upload(il) || d_rows.upload(row_items) || d_cols.upload(col_items) || d_tofs.upload(tab_ofs) ||
      d_ta0.upload(tab_a0) || d_ta1.upload(tab_a1) || d_nodes.upload(nodes) || d_trees.upload(trees) || d_stages.upload(stages) ||
      d_leaves.upload(m.leaves) || d_count.alloc(1) || (taps && d_taps.alloc(grid)))
    return set_error(HC_ERR_HIP, "legacy mode: device allocation / upload failed");
  hipStream_t st = nullptr;
  std::vector<unsigned long long> keys;
  bool done = false;
  for (int attempt = 0; attempt < 3 && !done; attempt++) {
    DevBuf<unsigned long long> cand;
    if (cand.alloc(cand_cap)) return set_error(HC_ERR_HIP, "legacy mode: candidate buffer allocation failed");
    hipError_t e = hipMemcpy2DAsync(d_img.p, cols, gray, step, cols, rows, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_sum.p, 0, (size_t)int_words * 4, st);     // row 0 / column 0 of every integral
    if (e == hipSuccess) e = hipMemsetAsync(d_sq.p, 0, (size_t)int_words * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_count.p, 0, 4, st);
    if (e == hipSuccess) e = (hipError_t)hc::launch_cv24_resize(st, (int)dl.size(), max_pix, d_img.p, cols, rows, cols, d_dl.p, d_tofs.p,
                                                                d_ta0.p, d_ta1.p, d_pyr.p);
    if (e == hipSuccess) e = (hipError_t)hc::launch_integral_rows(st, 1, (int)row_items.size(), d_pyr.p, 0, d_il.p, d_rows.p, d_sum.p, d_sq.p, 0);
    if (e == hipSuccess) e = (hipError_t)hc::launch_integral_cols(st, 1, (int)col_items.size(), d_il.p, d_cols.p, d_sum.p, d_sq.p, 0);
    hc::Cv24Args A{};
    A.levels = d_dl.p; A.sum = d_sum.p; A.sq = d_sq.p; A.stages = d_stages.p; A.trees = d_trees.p; A.nodes = d_nodes.p;
    A.leaves = d_leaves.p; A.inv_area = inv_area; A.win_w = m.win_w; A.win_h = m.win_h; A.cand = cand.p; A.cand_count = d_count.p;
    A.cand_cap = (unsigned int)cand_cap; A.taps = taps ? d_taps.p : nullptr;
    if (e == hipSuccess) e = (hipError_t)hc::launch_cv24_cascade(st, (int)dl.size(), max_pos, A);
    unsigned int count = 0;
    if (e == hipSuccess) e = hipMemcpy(&count, d_count.p, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return set_error(HC_ERR_HIP, std::string("legacy mode: ") + hipGetErrorString(e));
    if (count > cand_cap) { cand_cap = (size_t)count + 1024; continue; }   // grow and run again, as the main path does
    keys.resize(count);
    if (count && hipMemcpy(keys.data(), cand.p, (size_t)count * 8, hipMemcpyDeviceToHost) != hipSuccess)
      return set_error(HC_ERR_HIP, "legacy mode: candidate copy failed");
    if (taps && grid && hipMemcpy(taps, d_taps.p, (size_t)grid, hipMemcpyDeviceToHost) != hipSuccess)
      return set_error(HC_ERR_HIP, "legacy mode: tap copy failed");
    done = true;
  }
  if (!done) return set_error(HC_ERR_CAND_OVERFLOW, "legacy mode: raw candidate workspace overflow after growing it twice");
  std::sort(keys.begin(), keys.end());            // level -> y -> x: the single-thread order of upstream's loops
  for (unsigned long long k : keys) {
    const Level24& L = lv[(size_t)(k >> 32)];
    const int gy = (int)((k >> 16) & 0xffff), gx = (int)(k & 0xffff);
    boxes.push_back(hc_rect{rne(gx * L.ystep * L.factor), rne(gy * L.ystep * L.factor), L.win_w, L.win_h});   // B.9, f64
  }
  if (min_neighbors != 0) hc::group_rectangles(boxes, std::max(min_neighbors, 1), 0.2);
  return HC_OK;
}

}  // namespace

extern "C" {

int hc_detect_multiscale_cv24_taps(hc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step_bytes, double scale_factor,
                                   int min_neighbors, int flags, int min_w, int min_h, int max_w, int max_h, hc_rect* out,
                                   int out_capacity, int* out_count, int8_t* taps, size_t taps_capacity) {
  if (!c || !gray || !out_count) return set_error(HC_ERR_ARG, "null argument");
  *out_count = 0;
  if (!(scale_factor > 1.0)) return set_error(HC_ERR_ARG, "scaleFactor must be > 1");
  if (rows <= 0 || cols <= 0 || step_bytes < (size_t)cols) return set_error(HC_ERR_ARG, "bad image geometry");
  if (!(flags & 2))   // CV_HAAR_SCALE_IMAGE: without it 2.4 scales the FEATURES instead of the image (not restated)
    return set_error(HC_ERR_UNSUPPORTED, "legacy mode is the CASCADE_SCALE_IMAGE branch (the flag the reference passes)");
  if (c->model.has_tilted) return set_error(HC_ERR_UNSUPPORTED, "tilted features");
  for (const hc::ModelNode& n : c->model.nodes)
    for (int k = 0; k < 3; k++)
      for (int j = 0; j < 4; j++)
        if (n.rect[k][j] < 0 || n.rect[k][j] > 127) return set_error(HC_ERR_UNSUPPORTED, "rect coordinate out of range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return set_error(HC_ERR_NO_DEVICE, "no HIP device available (this backend has no CPU fallback)");
  if (cols > 65535 || rows > 65535) return set_error(HC_ERR_UNSUPPORTED, "frame too large");
  std::lock_guard<std::mutex> lock(c->mu);
  std::vector<hc_rect> boxes;
  int rc = detect24(c, gray, rows, cols, step_bytes, scale_factor, min_neighbors, min_w, min_h, max_w, max_h, boxes, taps, taps_capacity);
  if (rc != HC_OK) return rc;
  return hc::deliver_sync_result(c, boxes, out, out_capacity, out_count);
}

int hc_detect_multiscale_cv24(hc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step_bytes, double scale_factor,
                              int min_neighbors, int flags, int min_w, int min_h, int max_w, int max_h, hc_rect* out,
                              int out_capacity, int* out_count) {
  return hc_detect_multiscale_cv24_taps(c, gray, rows, cols, step_bytes, scale_factor, min_neighbors, flags, min_w, min_h, max_w,
                                        max_h, out, out_capacity, out_count, nullptr, 0);
}

int hc_plan_query_cv24(const hc_cascade* c, int rows, int cols, double scale_factor, int min_w, int min_h, int max_w, int max_h,
                       double* factors, int32_t* geometry, int cap, int* nlevels) {
  if (!c || !nlevels) return set_error(HC_ERR_ARG, "null argument");
  if (!(scale_factor > 1.0) || rows <= 0 || cols <= 0) return set_error(HC_ERR_ARG, "bad arguments");
  const std::vector<Level24> lv = plan24(c->model, rows, cols, scale_factor, min_w, min_h, max_w, max_h);
  *nlevels = (int)lv.size();
  if ((int)lv.size() > cap) return set_error(HC_ERR_CAPACITY, "level buffer too small");
  for (size_t i = 0; i < lv.size(); i++) {
    if (factors) factors[i] = lv[i].factor;
    if (geometry) {
      int32_t* g = geometry + 7 * i;
      g[0] = lv[i].w; g[1] = lv[i].h; g[2] = lv[i].ystep; g[3] = lv[i].nx; g[4] = lv[i].ny; g[5] = lv[i].win_w; g[6] = lv[i].win_h;
    }
  }
  return HC_OK;
}

int hc_resize_linear_cv24(const uint8_t* src, int rows, int cols, size_t step_bytes, uint8_t* dst, int drows, int dcols) {
  if (!src || !dst || rows <= 0 || cols <= 0 || drows <= 0 || dcols <= 0 || step_bytes < (size_t)cols)
    return set_error(HC_ERR_ARG, "bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return set_error(HC_ERR_NO_DEVICE, "no HIP device available (this backend has no CPU fallback)");
  hc::Cv24DevLevel d{};
  d.w = dcols; d.h = drows;
  d.resize_mode = (dcols == cols && drows == rows) ? 0 : ((cols == 2 * dcols && rows == 2 * drows) ? 2 : 1);
  std::vector<int> ofs;
  std::vector<short> a0, a1;
  d.xtab = 0;
  linear_table24(cols, dcols, ofs, a0, a1);
  d.ytab = (int)ofs.size();
  linear_table24(rows, drows, ofs, a0, a1);
  DevBuf<uint8_t> d_src, d_dst;
  DevBuf<hc::Cv24DevLevel> d_l;
  DevBuf<int> d_o;
  DevBuf<short> d_a0, d_a1;
  std::vector<hc::Cv24DevLevel> one(1, d);
  if (d_src.alloc((size_t)rows * cols) || d_dst.alloc((size_t)drows * dcols) || d_l.upload(one) || d_o.upload(ofs) || d_a0.upload(a0) ||
      d_a1.upload(a1))
    return set_error(HC_ERR_HIP, "hc_resize_linear_cv24: device allocation failed");
  hipError_t e = hipMemcpy2D(d_src.p, cols, src, step_bytes, cols, rows, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = (hipError_t)hc::launch_cv24_resize(nullptr, 1, drows * dcols, d_src.p, cols, rows, cols, d_l.p, d_o.p, d_a0.p, d_a1.p, d_dst.p);
  if (e == hipSuccess) e = hipMemcpy(dst, d_dst.p, (size_t)drows * dcols, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return set_error(HC_ERR_HIP, std::string("hc_resize_linear_cv24: ") + hipGetErrorString(e));
  return HC_OK;
}

}  // extern "C"
