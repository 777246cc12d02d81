// legacy_cv24.cpp -- host side of the OpenCV-2.4 ("T2") evaluation mode: the semantics the
// reference was literally built against (src/tracker/associator.cc:3 pins OpenCV 2.4) for the call at
// src/tracker/detector.cc:25-31 with its CASCADE_SCALE_IMAGE flag -- cvHaarDetectObjectsForROC, scale-image branch,
// scalar C path, the cascade's stage tree honoured (data/haarcascade_frontalface_alt_tree.xml:1303-1304).
// This file: the planning shared with the context code (scale loop, resize tables, hidden-cascade weights, stage tree)
// and the two stand-alone entry points hc_plan_query_cv24 / hc_resize_linear_cv24.  The call itself runs through a
// context like the OpenCV >= 4 semantics (api_context.cpp, hc_context_set_semantics).  Parity unpinned (no OpenCV reachable).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "hc_internal.h"
#include "kernels.h"
#include "legacy_cv24.h"
#include "legacy_cv24_host.h"

namespace hc {

namespace {
inline int rne24(double v) { return (int)std::nearbyint(v); }
inline int64_t align_up64(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
}  // namespace

std::vector<Level24> plan24(const Model& m, int rows, int cols, double sf, int min_w, int min_h, int max_w, int max_h) {
  std::vector<Level24> out;
  if (max_w == 0 || max_h == 0) { max_w = cols; max_h = rows; }
  for (double factor = 1;; factor *= sf) {
    const int ww = rne24(m.win_w * factor), wh = rne24(m.win_h * factor);
    const int w = rne24(cols / factor), h = rne24(rows / factor);
    if (w - m.win_w + 1 <= 0 || h - m.win_h + 1 <= 0) break;
    if (ww > max_w || wh > max_h) break;
    if (ww < min_w || wh < min_h) continue;
    const int ystep = factor > 2 ? 1 : 2;
    out.push_back(Level24{factor, w, h, ystep, (w - m.win_w + ystep - 1) / ystep, (h - m.win_h + ystep - 1) / ystep, ww, wh});
  }
  return out;
}

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

int build_plan_cv24(const Model& m, int rows, int cols, double sf, int min_w, int min_h, int max_w, int max_h, int band_rows1,
                    int band_rows2, Plan& p, Cv24Host& h) {
  p = Plan{};
  h = Cv24Host{};
  p.band_rows[0] = band_rows1;
  p.band_rows[1] = band_rows2;
  p.rows = rows; p.cols = cols; p.scale_factor = sf;
  p.min_w = min_w; p.min_h = min_h;
  if (!(sf > 1.0) || rows <= 0 || cols <= 0) return HC_ERR_ARG;
  if (max_w == 0 || max_h == 0) { max_w = cols; max_h = rows; }
  p.max_w = max_w; p.max_h = max_h;
  p.algorithmic_bytes = (int64_t)rows * cols;
  h.inv_area = 1. / ((double)(m.win_w - 2) * (m.win_h - 2));
  h.lv = plan24(m, rows, cols, sf, min_w, min_h, max_w, max_h);
  for (size_t li = 0; li < h.lv.size(); li++) {
    const Level24& L = h.lv[li];
    DevLevel D{};
    D.scale = (float)L.factor;
    D.w = L.w; D.h = L.h; D.ystep = L.ystep;
    D.szw_w = std::max(L.w - m.win_w, 0);                 // the last column / row of valid origins is not scanned (B.4)
    D.szw_h = std::max(L.h - m.win_h, 0);
    D.y_end = D.szw_h;                                    // no stripes in 2.4's serial loop
    D.nx = L.nx; D.ny = L.ny; D.ny_all = L.ny;
    D.win_w = L.win_w; D.win_h = L.win_h;
    D.int_pitch = (int)align_up64(L.w + 1 + 31, 32);
    D.pyr_ofs = p.pyr_bytes;
    D.int_ofs = p.int_words + 31;
    D.grid_ofs = p.grid_elems;
    Cv24DevLevel C{};
    C.w = L.w; C.h = L.h; C.ystep = L.ystep; C.nx = L.nx; C.ny = L.ny;
    C.int_pitch = D.int_pitch;
    C.resize_mode = (L.w == cols && L.h == rows) ? 0 : ((cols == 2 * L.w && rows == 2 * L.h) ? 2 : 1);
    C.xtab = (int)h.tab_ofs.size();
    linear_table24(cols, L.w, h.tab_ofs, h.tab_a0, h.tab_a1);
    C.ytab = (int)h.tab_ofs.size();
    linear_table24(rows, L.h, h.tab_ofs, h.tab_a0, h.tab_a1);
    C.pyr_ofs = D.pyr_ofs; C.int_ofs = D.int_ofs; C.grid_ofs = D.grid_ofs;
    // The same tables packed for the fused prep kernel (kernels.hip k_pyramid_integral, SEM 1): index | a0 << 13 |
    // (a0 + a1 - 2047) << 25.  2.4's two special cases are the general formula with particular coefficients: a copy is
    // a0 = 2048, a1 = 0 on both axes ((2048 * (p * 2048 >> 4) >> 16) = 4 p, (4 p + 2) >> 2 = p), the exact half is
    // a0 = a1 = 1024 at index 2 d ((1024 * ((p + q) * 1024 >> 4)) >> 16 = p + q, the 2 x 2 sum + 2 >> 2).
    D.xtab_ofs = (int)p.xtab.size();
    D.ytab_ofs = (int)p.ytab.size();
    for (int axis = 0; axis < 2; axis++) {
      const int n = axis == 0 ? L.w : L.h, first = axis == 0 ? C.xtab : C.ytab;
      std::vector<uint32_t>& out = axis == 0 ? p.xtab : p.ytab;
      for (int d = 0; d < n; d++) {
        int idx = h.tab_ofs[first + d], a0 = h.tab_a0[first + d], a1 = h.tab_a1[first + d];
        if (C.resize_mode == 0) { idx = d; a0 = 2048; a1 = 0; }
        else if (C.resize_mode == 2) { idx = 2 * d; a0 = 1024; a1 = 1024; }
        const int delta = a0 + a1 - 2047;
        if (idx < 0 || idx > 0x1fff || a0 < 0 || a0 > 0xfff || delta < 0 || delta > 3) h.fused_tables = false;
        out.push_back((uint32_t)idx | ((uint32_t)(a0 & 0xfff) << 13) | ((uint32_t)(delta & 3) << 25));
      }
    }
    p.pyr_bytes += align_up64((int64_t)L.w * L.h, 64);
    p.int_words += (int64_t)D.int_pitch * (L.h + 1) + 32;
    p.grid_elems += align_up64((int64_t)L.nx * L.ny, 64);
    p.algorithmic_bytes += 16ll * (L.w + 1) * (L.h + 1);
    h.max_pix = std::max(h.max_pix, L.w * L.h);
    h.max_pos = std::max(h.max_pos, L.nx * L.ny);
    p.levels.push_back(D);
    p.scanned.push_back(1);
    h.cl.push_back(C);
    for (int y0 = 0; y0 < L.h; y0 += 4) p.row_items.push_back(WorkItem{(int)li, y0, 0});
    for (int x0 = 0; x0 < L.w; x0 += WG_THREADS) p.col_items.push_back(WorkItem{(int)li, x0, 0});
    for (int i0 = 0; i0 < L.nx * L.ny; i0 += WG_THREADS) p.var_items.push_back(WorkItem{(int)li, i0, 0});
    const int cls = L.ystep == 1 ? 0 : 1;
    if (L.nx > 0)
      for (int gy0 = 0; gy0 < L.ny; gy0 += p.band_rows[cls]) p.band_items[cls].push_back(WorkItem{(int)li, gy0, 0});
  }
  return HC_OK;
}

void cv24_weights(const Model& m, const ModelNode& n, float w[3]) {
  const double inv_area = 1. / ((double)(m.win_w - 2) * (m.win_h - 2));
  double sum0 = 0, area0 = 0;
  for (int k = 0; k < 3; k++) {
    if (k > 0 && n.weight[k] == 0.f) { w[k] = 0.f; continue; }
    w[k] = (float)(n.weight[k] * inv_area);
    if (k == 0) area0 = n.rect[k][2] * n.rect[k][3];
    else sum0 += w[k] * n.rect[k][2] * n.rect[k][3];
  }
  w[0] = (float)(-sum0 / area0);
}

bool cv24_is_chain(const Model& m) {
  const int nst = (int)m.stages.size();
  for (int i = 0; i < nst; i++) {
    const int parent = (m.stages[i].parent >= 0 && m.stages[i].parent < nst) ? m.stages[i].parent : -1;
    const int next = (m.stages[i].next >= 0 && m.stages[i].next < nst) ? m.stages[i].next : -1;
    if (parent != i - 1 || next != -1) return false;
  }
  return true;
}

int cv24_tree_tables(const Model& m, std::vector<Cv24Node>& nodes, std::vector<Cv24Tree>& trees, std::vector<Cv24Stage>& stages) {
  nodes.assign(m.nodes.size(), Cv24Node{});
  for (size_t i = 0; i < m.nodes.size(); i++) {
    const ModelNode& n = m.nodes[i];
    Cv24Node d{};
    for (int k = 0; k < 3; k++)
      for (int j = 0; j < 4; j++) d.rect[k][j] = (int8_t)n.rect[k][j];
    cv24_weights(m, n, d.w);
    d.thr = n.thr; d.left = n.left; d.right = n.right; d.has3 = n.weight[2] != 0.f;
    nodes[i] = d;
  }
  trees.clear();
  for (const ModelTree& t : m.trees) trees.push_back(Cv24Tree{t.first_node, t.first_leaf});
  stages.clear();
  const int nst = (int)m.stages.size();
  for (const ModelStage& s : m.stages) {
    const int parent = (s.parent >= 0 && s.parent < nst) ? s.parent : -1, next = (s.next >= 0 && s.next < nst) ? s.next : -1;
    stages.push_back(Cv24Stage{s.first_tree, s.ntrees, cv24_stage_thr(s), parent, next, -1});
  }
  for (int i = 0; i < nst; i++)
    if (stages[i].parent >= 0 && stages[stages[i].parent].child < 0) stages[stages[i].parent].child = i;
  // a malformed parent / next graph must not hang the device: every walk visits each stage at most once if children and
  // next-siblings always have larger indices than where they are reached from
  for (int i = 0; i < nst; i++)
    if ((stages[i].child >= 0 && stages[i].child <= i) || (stages[i].next >= 0 && stages[i].next <= i) || stages[i].parent >= i)
      return set_error(HC_ERR_UNSUPPORTED, "stage tree of the cascade is not in forward order");
  return HC_OK;
}

}  // namespace hc

namespace {
using hc::set_error;
using hc::linear_table24;
using hc::plan24;
using hc::Level24;

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
}  // namespace

extern "C" {

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
  if (e == hipSuccess) e = (hipError_t)hc::launch_cv24_resize(nullptr, 1, 1, drows * dcols, d_src.p, 0, cols, rows, cols, d_l.p, d_o.p, d_a0.p, d_a1.p, d_dst.p, 0);
  if (e == hipSuccess) e = hipMemcpy(dst, d_dst.p, (size_t)drows * dcols, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return set_error(HC_ERR_HIP, std::string("hc_resize_linear_cv24: ") + hipGetErrorString(e));
  return HC_OK;
}

}  // extern "C"
