// haar_oracle.cpp -- CPU ORACLE, TEST INFRASTRUCTURE ONLY (see haar_oracle.h).
//
// PARITY UNPINNED: restates OpenCV >= 4.0 semantics of
// cv::CascadeClassifier::{load,detectMultiScale} as used by the reference at
// src/file_io.cc:11 and src/tracker/detector.cc:25-31 on the old-format cascades in
// reference data/haarcascade_frontalface_*.xml.  OpenCV itself is not available here;
// every rule is tagged with the SURVEY.md Appendix-A paragraph it follows and with the
// upstream function it restates (modules/objdetect/src/cascadedetect.{cpp,hpp},
// cascadedetect_convert.cpp, modules/imgproc/src/resize.cpp + fixedpoint.inl.hpp,
// modules/imgproc/src/sumpixels.*, core operations.hpp partition()).
//
// Build: g++ -O2 -ffp-contract=off -fno-fast-math  (no FMA: upstream x86-64 baseline).
// Canonical order everywhere = OpenCV with one thread: scale -> y -> x.

#include "haar_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

// cvRound: round-half-to-even under the default FP environment (A: "cvRound").
inline int cv_round(double v) { return (int)std::lrint(v); }
inline int cv_round(float v) { return (int)std::lrintf(v); }

// ------------------------------------------------------------------ XML (A.1)
struct XNode {
  std::string name;
  std::string type_id;
  std::string text;
  std::vector<XNode*> kids;
  ~XNode() { for (auto* k : kids) delete k; }
  const XNode* child(const char* n) const {
    for (auto* k : kids) if (k->name == n) return k;
    return nullptr;
  }
};

struct XParser {
  const char* p; const char* e;
  bool ok = true;
  void skip_misc() {
    for (;;) {
      while (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
      if (e - p >= 4 && !memcmp(p, "<!--", 4)) {
        const char* q = (const char*)memmem(p + 4, e - p - 4, "-->", 3);
        if (!q) { ok = false; p = e; return; }
        p = q + 3; continue;
      }
      if (e - p >= 2 && !memcmp(p, "<?", 2)) {
        const char* q = (const char*)memmem(p, e - p, "?>", 2);
        if (!q) { ok = false; p = e; return; }
        p = q + 2; continue;
      }
      return;
    }
  }
  XNode* element() {
    skip_misc();
    if (p >= e || *p != '<') { ok = false; return nullptr; }
    ++p;
    XNode* n = new XNode;
    while (p < e && *p != ' ' && *p != '>' && *p != '/' && *p != '\n' && *p != '\t') n->name += *p++;
    // attributes
    for (;;) {
      while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
      if (p >= e) { ok = false; return n; }
      if (*p == '>') { ++p; break; }
      if (*p == '/') { p += 2; return n; }  // <x/>
      std::string an, av;
      while (p < e && *p != '=' && *p != '>' ) an += *p++;
      if (p < e && *p == '=') {
        ++p;
        char q = *p++;
        while (p < e && *p != q) av += *p++;
        ++p;
      }
      if (an == "type_id") n->type_id = av;
    }
    // content
    for (;;) {
      // text run
      while (p < e && *p != '<') n->text += *p++;
      if (p >= e) { ok = false; return n; }
      if (e - p >= 4 && !memcmp(p, "<!--", 4)) { skip_misc(); continue; }
      if (p[1] == '/') {
        const char* q = (const char*)memchr(p, '>', e - p);
        if (!q) { ok = false; return n; }
        p = q + 1;
        return n;
      }
      XNode* k = element();
      if (k) n->kids.push_back(k);
      if (!ok) return n;
    }
  }
};

double parse_real(const std::string& s) { return strtod(s.c_str(), nullptr); }  // A.2: text -> f64
int parse_int(const std::string& s) { return (int)strtol(s.c_str(), nullptr, 10); }

// ------------------------------------------------------------------ model (A.2)
struct Feature { int r[3][4]; float w[3]; bool tilted; };
struct Node { int feat; float thr; int left, right; };   // left/right: <=0 -> -(leaf idx in tree), >0 node idx
struct Tree { int nnodes; int first_node; int first_leaf; };
struct Stage { float thr; int ntrees; int first_tree; int parent, next; double thr_raw; };

}  // namespace

struct orc_cascade {
  int win_w = 0, win_h = 0;
  std::vector<Stage> stages;
  std::vector<Tree> trees;
  std::vector<Node> nodes;
  std::vector<Feature> feats;  // one per node, encounter order
  std::vector<float> leaves;
  int max_nodes = 0;
  bool tilted = false;
};

namespace {

const float THRESHOLD_EPS = 1e-5f;   // A.2, upstream CascadeClassifierImpl::Data::read

void finish_stage_thr(orc_cascade* c) {
  for (auto& s : c->stages) s.thr = (float)s.thr_raw - THRESHOLD_EPS;  // f32 - f32 in f32
}

bool build_from_xml(const XNode* root, orc_cascade* c) {
  // A.1: first child of <opencv_storage> whose type_id is opencv-haar-classifier
  const XNode* top = nullptr;
  for (auto* k : root->kids) if (k->type_id == "opencv-haar-classifier") { top = k; break; }
  if (!top) return false;
  const XNode* sz = top->child("size");
  if (!sz) return false;
  {
    const char* s = sz->text.c_str(); char* end;
    c->win_w = (int)strtol(s, &end, 10);
    c->win_h = (int)strtol(end, &end, 10);
  }
  const XNode* stages = top->child("stages");
  if (!stages) return false;
  for (auto* st : stages->kids) {
    Stage S{};
    const XNode* thr = st->child("stage_threshold");
    const XNode* trees = st->child("trees");
    if (!thr || !trees) return false;
    S.thr_raw = parse_real(thr->text);
    const XNode* par = st->child("parent"); const XNode* nxt = st->child("next");
    S.parent = par ? parse_int(par->text) : -1;   // A.1: parsed but ignored by the T4 converter
    S.next = nxt ? parse_int(nxt->text) : -1;
    S.first_tree = (int)c->trees.size();
    S.ntrees = (int)trees->kids.size();
    for (auto* tr : trees->kids) {
      Tree T{}; T.nnodes = (int)tr->kids.size(); T.first_node = (int)c->nodes.size();
      T.first_leaf = (int)c->leaves.size();
      int nleaves = 0;
      for (auto* nd : tr->kids) {
        Feature F{}; Node N{};
        const XNode* fe = nd->child("feature");
        if (!fe) return false;
        const XNode* rects = fe->child("rects"); const XNode* tl = fe->child("tilted");
        if (!rects) return false;
        F.tilted = tl && parse_int(tl->text) != 0;
        c->tilted = c->tilted || F.tilted;
        int k = 0;
        for (auto* rn : rects->kids) {
          if (k >= 3) return false;
          const char* s = rn->text.c_str(); char* end;
          for (int j = 0; j < 4; j++) { F.r[k][j] = (int)strtol(s, &end, 10); s = end; }
          F.w[k] = (float)strtod(s, &end);          // A.2: weight -> f32
          k++;
        }
        for (; k < 3; k++) { F.r[k][0] = F.r[k][1] = F.r[k][2] = F.r[k][3] = 0; F.w[k] = 0.f; }
        const XNode* th = nd->child("threshold");
        if (!th) return false;
        N.thr = (float)parse_real(th->text);        // A.2: f64 -> f32
        N.feat = (int)c->feats.size();
        // A.2 leaf encoding: leaves stored in encounter order, left then right
        const XNode* lv = nd->child("left_val");
        if (lv) { N.left = -nleaves; c->leaves.push_back((float)parse_real(lv->text)); nleaves++; }
        else { const XNode* ln = nd->child("left_node"); if (!ln) return false; N.left = parse_int(ln->text); }
        const XNode* rv = nd->child("right_val");
        if (rv) { N.right = -nleaves; c->leaves.push_back((float)parse_real(rv->text)); nleaves++; }
        else { const XNode* rn2 = nd->child("right_node"); if (!rn2) return false; N.right = parse_int(rn2->text); }
        c->feats.push_back(F);
        c->nodes.push_back(N);
      }
      if (nleaves != T.nnodes + 1) return false;     // binary tree invariant assumed by A.9 indexing
      c->max_nodes = std::max(c->max_nodes, T.nnodes);
      c->trees.push_back(T);
    }
    c->stages.push_back(S);
  }
  finish_stage_thr(c);
  return !c->stages.empty() && c->win_w > 0 && c->win_h > 0;
}

// rect sum on an integral image with row pitch `p` (A.6a S(r))
inline int rsum(const int32_t* I, int p, int x, int y, int w, int h) {
  const uint32_t* U = (const uint32_t*)I;
  uint32_t v = U[y * p + x] - U[y * p + x + w] - U[(y + h) * p + x] + U[(y + h) * p + x + w];
  return (int)v;
}

struct WinEval {
  const orc_cascade* c;
  const int32_t* sum; const int32_t* sq; int pitch;
  float vnf;
  // A.6a setWindow
  bool set_window(int x, int y) {
    int nw = c->win_w - 2, nh = c->win_h - 2;
    int valsum = rsum(sum, pitch, x + 1, y + 1, nw, nh);
    unsigned valsq = (unsigned)rsum(sq, pitch, x + 1, y + 1, nw, nh);
    double area = (double)(nw * nh);
    double nf = area * valsq - (double)valsum * valsum;
    if (nf > 0.) {
      nf = std::sqrt(nf);
      vnf = (float)(1. / nf);
      return area * vnf < 1e-1;
    }
    vnf = 1.f;
    return false;
  }
  // HaarEvaluator::operator(): OptFeature::calc * varianceNormFactor (A.8)
  inline float feature(const Feature& f, int x, int y) const {
    float ret = f.w[0] * (float)rsum(sum, pitch, x + f.r[0][0], y + f.r[0][1], f.r[0][2], f.r[0][3]) +
                f.w[1] * (float)rsum(sum, pitch, x + f.r[1][0], y + f.r[1][1], f.r[1][2], f.r[1][3]);
    if (f.w[2] != 0.0f)
      ret += f.w[2] * (float)rsum(sum, pitch, x + f.r[2][0], y + f.r[2][1], f.r[2][2], f.r[2][3]);
    return ret * vnf;
  }
  // returns number of stages passed (nstages => accepted). margins optional.
  int run(int x, int y, double* margins, bool keep_going) const {
    int nst = (int)c->stages.size();
    int passed = -1;
    for (int si = 0; si < nst; si++) {
      const Stage& S = c->stages[si];
      double tmp = 0;
      for (int t = 0; t < S.ntrees; t++) {
        const Tree& T = c->trees[S.first_tree + t];
        if (c->max_nodes == 1) {                       // A.8 predictOrderedStump
          const Node& N = c->nodes[T.first_node];
          double value = feature(c->feats[N.feat], x, y);
          tmp += value < N.thr ? c->leaves[T.first_leaf] : c->leaves[T.first_leaf + 1];
        } else {                                       // A.9 predictOrdered
          int idx = 0;
          do {
            const Node& N = c->nodes[T.first_node + idx];
            double val = feature(c->feats[N.feat], x, y);
            idx = val < N.thr ? N.left : N.right;
          } while (idx > 0);
          tmp += c->leaves[T.first_leaf - idx];
        }
      }
      if (margins) margins[si] = tmp - (double)S.thr;
      if (tmp < S.thr) {
        if (passed < 0) passed = si;
        if (!keep_going) return passed;
      }
    }
    return passed < 0 ? nst : passed;
  }
};

// ------------------------------------------------------------------ plan (A.3, A.6b)
int make_plan(const orc_cascade* c, int rows, int cols, double scale_factor,
              int min_w, int min_h, int max_w, int max_h,
              std::vector<orc_level>& L, int* nstripes_out) {
  L.clear();
  *nstripes_out = 0;
  if (!(scale_factor > 1)) return -2;                  // CV_Assert(scaleFactor > 1)
  if (max_h == 0 || max_w == 0) { max_w = cols; max_h = rows; }
  if (rows < c->win_h || cols < c->win_w) return 0;
  std::vector<float> all_scales, scales;
  for (double factor = 1;; factor *= scale_factor) {
    int ww = cv_round(c->win_w * factor), wh = cv_round(c->win_h * factor);
    if (ww > cols || wh > rows) break;
    all_scales.push_back((float)factor);
  }
  for (float s : all_scales) {
    int ww = cv_round(c->win_w * s), wh = cv_round(c->win_h * s);   // int * f32 in f32
    if (ww > max_w || wh > max_h) break;
    if (ww < min_w || wh < min_h) continue;
    scales.push_back(s);
  }
  if (scales.empty() && !all_scales.empty()) {          // A.3 fallback (low confidence, unreachable in configs)
    size_t imin = 0; double dmin = 0;
    for (size_t v = 0; v < all_scales.size(); v++) {
      int ww = cv_round(c->win_w * all_scales[v]), wh = cv_round(c->win_h * all_scales[v]);
      double d = (double)(min_w - ww) * (min_w - ww) + (double)(min_h - wh) * (min_h - wh);
      if (v == 0 || dmin > d) { dmin = d; imin = v; }
    }
    scales.push_back(all_scales[imin]);
  }
  if (scales.empty()) return 0;
  for (float sc : scales) {
    orc_level l{};
    l.scale = sc;
    l.w = cv_round((float)cols / sc);                    // int -> f32 / f32
    l.h = cv_round((float)rows / sc);
    l.ystep = sc >= 2 ? 1 : 2;
    l.szw_w = std::max(l.w + 1 - c->win_w, 0);
    l.szw_h = std::max(l.h + 1 - c->win_h, 0);
    l.win_w = cv_round(c->win_w * sc);
    l.win_h = cv_round(c->win_h * sc);
    L.push_back(l);
  }
  int nstripes = (int)std::ceil(L[0].szw_w / 32.);
  for (auto& l : L) {
    l.stripe = std::max((l.szw_h / l.ystep + nstripes - 1) / std::max(nstripes, 1), 1) * l.ystep;
    l.y_end = std::min(nstripes * l.stripe, l.szw_h);
    l.nx = (l.szw_w + l.ystep - 1) / l.ystep;
    l.ny = (l.y_end + l.ystep - 1) / l.ystep;
  }
  *nstripes_out = nstripes;
  return 0;
}

// ------------------------------------------------------------------ resize (A.4)
struct LinTab { std::vector<int> ofs; std::vector<uint16_t> c0, c1; int dmin, dmax; };

void lin_coeffs(int src, int dst, LinTab& t) {
  // interpolationLinear<uint8_t>: scale = 1/inv_scale, inv_scale = dst/src, all IEEE f64
  double inv_scale = (double)dst / src;
  double scale = 1.0 / inv_scale;
  t.ofs.assign(dst, 0); t.c0.assign(dst, 0); t.c1.assign(dst, 0);
  t.dmin = 0; t.dmax = dst;
  for (int d = 0; d < dst; d++) {
    double fval = scale * ((double)d + 0.5) - 0.5;
    int ival = (int)std::floor(fval);
    if (ival >= 0 && src > 1) {
      if (ival < src - 1) {
        t.ofs[d] = ival;
        double fr = fval - (double)ival;
        int c1 = fr < 0 ? 0 : cv_round(fr * 256.0);     // ufixedpoint16(softdouble)
        t.c1[d] = (uint16_t)c1; t.c0[d] = (uint16_t)(256 - c1);
      } else {
        t.ofs[d] = src - 1;
        t.dmax = std::min(t.dmax, d);
      }
    } else {
      t.dmin = std::max(t.dmin, d + 1);
    }
  }
}

void resize_exact(const uint8_t* src, int rows, int cols, size_t step, uint8_t* dst, int drows, int dcols) {
  if (drows == rows && dcols == cols) {                  // cv::resize: dsize == ssize -> copy
    for (int y = 0; y < rows; y++) memcpy(dst + (size_t)y * dcols, src + (size_t)y * step, cols);
    return;
  }
  LinTab tx, ty;
  lin_coeffs(cols, dcols, tx);
  lin_coeffs(rows, drows, ty);
  std::vector<uint16_t> h0(dcols), h1(dcols);
  auto hline = [&](const uint8_t* s, uint16_t* out) {    // hlineResizeCn<uint8_t, ufixedpoint16, 2, true, 1>
    int d = 0;
    for (; d < tx.dmin; d++) out[d] = (uint16_t)(s[0] << 8);
    for (; d < tx.dmax; d++) {
      const uint8_t* px = s + tx.ofs[d];
      out[d] = (uint16_t)(tx.c0[d] * px[0] + tx.c1[d] * px[1]);
    }
    uint8_t last = s[tx.ofs[dcols - 1]];
    for (; d < dcols; d++) out[d] = (uint16_t)(last << 8);
  };
  for (int dy = 0; dy < drows; dy++) {
    uint8_t* o = dst + (size_t)dy * dcols;
    if (dy < ty.dmin) {                                   // vlineSet from first source row
      hline(src, h0.data());
      for (int d = 0; d < dcols; d++) o[d] = (uint8_t)((h0[d] + 128) >> 8);
    } else if (dy >= ty.dmax) {                           // vlineSet from last source row
      hline(src + (size_t)(rows - 1) * step, h0.data());
      for (int d = 0; d < dcols; d++) o[d] = (uint8_t)((h0[d] + 128) >> 8);
    } else {                                              // vlineResize<uint8_t, ufixedpoint16, 2>
      int iy = ty.ofs[dy];
      hline(src + (size_t)iy * step, h0.data());
      hline(src + (size_t)(iy + 1) * step, h1.data());
      uint32_t cy0 = ty.c0[dy], cy1 = ty.c1[dy];
      for (int d = 0; d < dcols; d++) {
        uint32_t v = (uint32_t)h0[d] * cy0 + (uint32_t)h1[d] * cy1;
        uint32_t r = (v + 32768u) >> 16;
        o[d] = (uint8_t)(r > 255 ? 255 : r);
      }
    }
  }
}

// ------------------------------------------------------------------ integral (A.5)
void integral_32s(const uint8_t* img, int rows, int cols, size_t step, int32_t* sum, int32_t* sq) {
  int p = cols + 1;
  uint32_t* S = (uint32_t*)sum; uint32_t* Q = (uint32_t*)sq;
  for (int x = 0; x < p; x++) { S[x] = 0; Q[x] = 0; }
  for (int y = 0; y < rows; y++) {
    const uint8_t* r = img + (size_t)y * step;
    uint32_t s = 0, q = 0;
    S[(y + 1) * p] = 0; Q[(y + 1) * p] = 0;
    for (int x = 0; x < cols; x++) {
      uint32_t v = r[x];
      s += v; q += v * v;                                 // wrap-around modulo 2^32
      S[(y + 1) * p + x + 1] = S[y * p + x + 1] + s;
      Q[(y + 1) * p + x + 1] = Q[y * p + x + 1] + q;
    }
  }
}

// ------------------------------------------------------------------ grouping (A.10)
bool similar(const orc_rect& a, const orc_rect& b, double eps) {
  double delta = eps * (std::min(a.w, b.w) + std::min(a.h, b.h)) * 0.5;
  return std::abs(a.x - b.x) <= delta && std::abs(a.y - b.y) <= delta &&
         std::abs(a.x + a.w - b.x - b.w) <= delta && std::abs(a.y + a.h - b.y - b.h) <= delta;
}

void group_rects(std::vector<orc_rect>& rl, int thr, double eps) {
  if (thr <= 0 || rl.empty()) return;
  int n = (int)rl.size();
  // cv::partition == connected components; class id = order of first member
  std::vector<int> parent(n);
  for (int i = 0; i < n; i++) parent[i] = i;
  auto find = [&](int a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++)
      if (similar(rl[i], rl[j], eps)) { int a = find(i), b = find(j); if (a != b) parent[std::max(a, b)] = std::min(a, b); }
  std::vector<int> label(n, -1), root_label(n, -1);
  int ncls = 0;
  for (int i = 0; i < n; i++) { int r = find(i); if (root_label[r] < 0) root_label[r] = ncls++; label[i] = root_label[r]; }
  std::vector<orc_rect> rr(ncls, orc_rect{0, 0, 0, 0});
  std::vector<int> cnt(ncls, 0);
  for (int i = 0; i < n; i++) {
    int k = label[i];
    rr[k].x += rl[i].x; rr[k].y += rl[i].y; rr[k].w += rl[i].w; rr[k].h += rl[i].h; cnt[k]++;
  }
  for (int k = 0; k < ncls; k++) {
    float s = 1.f / cnt[k];
    rr[k] = orc_rect{cv_round(rr[k].x * s), cv_round(rr[k].y * s), cv_round(rr[k].w * s), cv_round(rr[k].h * s)};
  }
  rl.clear();
  for (int i = 0; i < ncls; i++) {
    orc_rect r1 = rr[i]; int n1 = cnt[i];
    if (n1 <= thr) continue;
    int j;
    for (j = 0; j < ncls; j++) {
      int n2 = cnt[j];
      if (j == i || n2 <= thr) continue;
      orc_rect r2 = rr[j];
      int dx = cv_round(r2.w * eps), dy = cv_round(r2.h * eps);
      if (r1.x >= r2.x - dx && r1.y >= r2.y - dy && r1.x + r1.w <= r2.x + r2.w + dx &&
          r1.y + r1.h <= r2.y + r2.h + dy && (n2 > std::max(3, n1) || n1 < 3))
        break;
    }
    if (j == ncls) rl.push_back(r1);
  }
}

// ------------------------------------------------------------------ whole call
int detect_impl(const orc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step,
                double sf, int min_neighbors, int min_w, int min_h, int max_w, int max_h,
                std::vector<orc_rect>& grouped, std::vector<orc_rect>* raw_out,
                int8_t* taps, size_t taps_cap) {
  std::vector<orc_level> L; int nstripes = 0;
  int rc = make_plan(c, rows, cols, sf, min_w, min_h, max_w, max_h, L, &nstripes);
  grouped.clear();
  if (raw_out) raw_out->clear();
  if (rc) return rc;
  if (c->tilted) return -4;   // none of the reference cascades has tilted features
  std::vector<orc_rect> cand;
  std::vector<uint8_t> img; std::vector<int32_t> sum, sq;
  size_t tap_ofs = 0;
  for (const orc_level& l : L) {
    img.resize((size_t)l.w * l.h);
    resize_exact(gray, rows, cols, step, img.data(), l.h, l.w);              // A.4 (always from the original)
    sum.resize((size_t)(l.w + 1) * (l.h + 1)); sq.resize(sum.size());
    integral_32s(img.data(), l.h, l.w, l.w, sum.data(), sq.data());           // A.5
    WinEval ev{c, sum.data(), sq.data(), l.w + 1, 1.f};
    int gnx = (l.szw_w + l.ystep - 1) / l.ystep, gny = (l.szw_h + l.ystep - 1) / l.ystep;
    int8_t* tp = nullptr;
    if (taps) {
      if (tap_ofs + (size_t)gnx * gny > taps_cap) return -3;
      tp = taps + tap_ofs; memset(tp, -2, (size_t)gnx * gny);
      tap_ofs += (size_t)gnx * gny;
    }
    int nst = (int)c->stages.size();
    for (int y = 0; y < l.y_end; y += l.ystep) {                              // A.6b
      for (int x = 0; x < l.szw_w; x += l.ystep) {
        int passed;
        if (!ev.set_window(x, y)) passed = -1;                                // A.6a -> result -1
        else passed = ev.run(x, y, nullptr, false);
        if (tp) tp[(y / l.ystep) * gnx + (x / l.ystep)] = (int8_t)passed;
        if (passed == nst)                                                     // A.6c
          cand.push_back(orc_rect{cv_round(x * l.scale), cv_round(y * l.scale), l.win_w, l.win_h});
        if (passed == 0) x += l.ystep;                                        // A.7 (result == 0)
      }
    }
  }
  if (raw_out) *raw_out = cand;
  group_rects(cand, min_neighbors, 0.2);                                       // A.10, GROUP_EPS
  grouped = cand;
  return 0;
}

}  // namespace

// ====================================================================== C API
extern "C" {

int orc_cascade_load_xml(const char* path, orc_cascade** out) {
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  std::string buf((size_t)n, '\0');
  if (fread(&buf[0], 1, (size_t)n, f) != (size_t)n) { fclose(f); return -1; }
  fclose(f);
  XParser P{buf.data(), buf.data() + buf.size()};
  XNode* root = P.element();
  if (!root || !P.ok || root->name != "opencv_storage") { delete root; return -2; }
  orc_cascade* c = new orc_cascade;
  bool ok = build_from_xml(root, c);
  delete root;
  if (!ok) { delete c; return -2; }
  *out = c;
  return 0;
}

// packed fixture format "HCC1" -- written by tools/pack_cascade.py from the reference XML
int orc_cascade_load_hcc(const char* path, orc_cascade** out) {
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> b((size_t)n);
  if (fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); return -1; }
  fclose(f);
  size_t p = 0;
  auto need = [&](size_t k) { return p + k <= b.size(); };
  auto rd32 = [&]() { int32_t v; memcpy(&v, &b[p], 4); p += 4; return v; };
  auto rdf = [&]() { float v; memcpy(&v, &b[p], 4); p += 4; return v; };
  auto rdd = [&]() { double v; memcpy(&v, &b[p], 8); p += 8; return v; };
  if (!need(32) || memcmp(b.data(), "HCC1", 4)) return -2;
  p = 4;
  orc_cascade* c = new orc_cascade;
  c->win_w = rd32(); c->win_h = rd32();
  int nst = rd32(), ntr = rd32(), nnd = rd32(), nlf = rd32(); int flags = rd32();
  c->tilted = flags & 1;
  if (!need((size_t)nst * 20)) { delete c; return -2; }
  for (int i = 0; i < nst; i++) {
    Stage S{}; S.thr_raw = rdd(); S.ntrees = rd32(); S.parent = rd32(); S.next = rd32();
    c->stages.push_back(S);
  }
  if (!need((size_t)ntr * 4)) { delete c; return -2; }
  int fn = 0, fl = 0, acc = 0;
  for (auto& S : c->stages) { S.first_tree = acc; acc += S.ntrees; }
  if (acc != ntr) { delete c; return -2; }
  for (int i = 0; i < ntr; i++) {
    Tree T{}; T.nnodes = rd32(); T.first_node = fn; T.first_leaf = fl;
    fn += T.nnodes; fl += T.nnodes + 1;
    c->max_nodes = std::max(c->max_nodes, T.nnodes);
    c->trees.push_back(T);
  }
  if (fn != nnd || fl != nlf || !need((size_t)nnd * 40 + (size_t)nlf * 4)) { delete c; return -2; }
  for (int i = 0; i < nnd; i++) {
    Feature F{}; Node N{};
    for (int k = 0; k < 3; k++) for (int j = 0; j < 4; j++) F.r[k][j] = (int8_t)b[p++];
    for (int k = 0; k < 3; k++) F.w[k] = rdf();
    N.thr = rdf(); N.left = rd32(); N.right = rd32();
    F.tilted = rd32() != 0;
    N.feat = i;
    c->feats.push_back(F); c->nodes.push_back(N);
  }
  for (int i = 0; i < nlf; i++) c->leaves.push_back(rdf());
  finish_stage_thr(c);
  *out = c;
  return 0;
}

void orc_cascade_free(orc_cascade* c) { delete c; }

int orc_cascade_info(const orc_cascade* c, int32_t info[8]) {
  if (!c) return -1;
  info[0] = c->win_w; info[1] = c->win_h; info[2] = (int)c->stages.size(); info[3] = (int)c->trees.size();
  info[4] = (int)c->nodes.size(); info[5] = (int)c->leaves.size(); info[6] = c->max_nodes; info[7] = c->tilted;
  return 0;
}

int orc_cascade_dump(const orc_cascade* c, float* stage_thr, int32_t* stage_ntrees, int32_t* tree_nnodes,
                     int32_t* node_rects, float* node_thr, int32_t* node_left, int32_t* node_right, float* leaves) {
  if (!c) return -1;
  for (size_t i = 0; i < c->stages.size(); i++) {
    if (stage_thr) stage_thr[i] = c->stages[i].thr;
    if (stage_ntrees) stage_ntrees[i] = c->stages[i].ntrees;
  }
  if (tree_nnodes) for (size_t i = 0; i < c->trees.size(); i++) tree_nnodes[i] = c->trees[i].nnodes;
  for (size_t i = 0; i < c->nodes.size(); i++) {
    const Feature& F = c->feats[c->nodes[i].feat];
    if (node_rects) for (int k = 0; k < 3; k++) {
      for (int j = 0; j < 4; j++) node_rects[i * 15 + k * 5 + j] = F.r[k][j];
      node_rects[i * 15 + k * 5 + 4] = (int)F.w[k];
    }
    if (node_thr) node_thr[i] = c->nodes[i].thr;
    if (node_left) node_left[i] = c->nodes[i].left;
    if (node_right) node_right[i] = c->nodes[i].right;
  }
  if (leaves) memcpy(leaves, c->leaves.data(), c->leaves.size() * sizeof(float));
  return 0;
}

int orc_plan(const orc_cascade* c, int rows, int cols, double scale_factor, int min_w, int min_h,
             int max_w, int max_h, orc_level* levels, int cap, int* nlevels, int* nstripes) {
  std::vector<orc_level> L; int ns = 0;
  int rc = make_plan(c, rows, cols, scale_factor, min_w, min_h, max_w, max_h, L, &ns);
  if (rc) return rc;
  if ((int)L.size() > cap) return -3;
  for (size_t i = 0; i < L.size(); i++) levels[i] = L[i];
  *nlevels = (int)L.size();
  if (nstripes) *nstripes = ns;
  return 0;
}

int orc_resize_linear_exact(const uint8_t* src, int rows, int cols, size_t step, uint8_t* dst, int drows, int dcols) {
  if (rows <= 0 || cols <= 0 || drows <= 0 || dcols <= 0) return -2;
  resize_exact(src, rows, cols, step, dst, drows, dcols);
  return 0;
}

int orc_integral(const uint8_t* img, int rows, int cols, size_t step, int32_t* sum, int32_t* sqsum) {
  integral_32s(img, rows, cols, step, sum, sqsum);
  return 0;
}

int orc_group_rectangles(const orc_rect* in, int n, int group_threshold, double eps, orc_rect* out, int cap, int* nout) {
  std::vector<orc_rect> v(in, in + n);
  group_rects(v, group_threshold, eps);
  if ((int)v.size() > cap) return -3;
  for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
  *nout = (int)v.size();
  return 0;
}

int orc_detect(const orc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step, double sf,
               int min_neighbors, int min_w, int min_h, int max_w, int max_h, orc_rect* out, int cap,
               int* out_count, orc_rect* raw, int raw_cap, int* raw_count, int8_t* taps, size_t taps_cap) {
  std::vector<orc_rect> g, r;
  int rc = detect_impl(c, gray, rows, cols, step, sf, min_neighbors, min_w, min_h, max_w, max_h, g,
                       (raw || raw_count) ? &r : nullptr, taps, taps_cap);
  if (rc) return rc;
  if ((int)g.size() > cap) return -3;
  for (size_t i = 0; i < g.size(); i++) out[i] = g[i];
  *out_count = (int)g.size();
  if (raw_count) *raw_count = (int)r.size();
  if (raw) {
    if ((int)r.size() > raw_cap) return -3;
    for (size_t i = 0; i < r.size(); i++) raw[i] = r[i];
  }
  return 0;
}

int orc_detect_many(const orc_cascade* c, const uint8_t* frames, int nframes, int rows, int cols, double sf,
                    int min_neighbors, int min_w, int min_h, int max_w, int max_h, int nthreads, int32_t* counts) {
  std::atomic<int> next(0), err(0);
  auto work = [&]() {
    std::vector<orc_rect> g;
    for (;;) {
      int i = next.fetch_add(1);
      if (i >= nframes) break;
      int rc = detect_impl(c, frames + (size_t)i * rows * cols, rows, cols, cols, sf, min_neighbors, min_w, min_h,
                           max_w, max_h, g, nullptr, nullptr, 0);
      if (rc) err = rc;
      counts[i] = (int)g.size();
    }
  };
  if (nthreads <= 1) { work(); return err; }
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) th.emplace_back(work);
  for (auto& t : th) t.join();
  return err;
}

// cvtColor(src, gray, COLOR_BGR2GRAY) for 8UC3 / 8UC4 (revised in round 2): OpenCV >= 4 computes the
// 8-bit gray with the 15-bit coefficient set RY15/GY15/BY15 and gray_shift = 15, while the Y of BGR2YCrCb keeps the
// 14-bit R2Y/G2Y/B2Y with yuv_shift = 14 (orc_detect_faces below).  The two differ by +-1 on some pixels.
// [UPSTREAM-RECALL, parity unpinned: no OpenCV here and no fixture in the reference]
static inline int gray15(int b, int g, int r) { return (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15; }

int orc_bgr2gray(const uint8_t* src, int rows, int cols, int channels, size_t step, uint8_t* gray) {
  if (rows <= 0 || cols <= 0 || (channels != 3 && channels != 4)) return -2;
  for (int y = 0; y < rows; y++) {
    const uint8_t* p = src + (size_t)y * step;
    for (int x = 0; x < cols; x++)
      gray[(size_t)y * cols + x] = (uint8_t)gray15(p[(size_t)channels * x], p[(size_t)channels * x + 1], p[(size_t)channels * x + 2]);
  }
  return 0;
}

// FaceDetector::detectFaces (reference src/tracker/detector.cc:10-43), OpenCV 8-bit paths per SURVEY A.13
int orc_detect_faces(const orc_cascade* c, const uint8_t* bgr, int rows, int cols, size_t step, const int lower[3],
                     const int upper[3], double scale_factor, orc_rect* out, int cap, int* out_count,
                     uint8_t* mask_tap, uint8_t* gray_tap) {
  const int shift = 14, half = 1 << (shift - 1);
  const int B2Y = 1868, G2Y = 9617, R2Y = 4899, YCRI = 11682, YCBI = 9241, delta = 128 << shift;
  std::vector<uint8_t> mask((size_t)rows * cols), gray((size_t)rows * cols);
  std::vector<int> hist(256, 0);
  auto sat8 = [](int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); };
  for (int y = 0; y < rows; y++) {
    const uint8_t* p = bgr + (size_t)y * step;
    for (int x = 0; x < cols; x++) {
      const int b = p[3 * x], g = p[3 * x + 1], r = p[3 * x + 2];
      const int Y = (b * B2Y + g * G2Y + r * R2Y + half) >> shift;       // cvtColor(BGR2YCrCb).Y: yuv_shift = 14
      const int Cr = sat8(((r - Y) * YCRI + delta + half) >> shift);
      const int Cb = sat8(((b - Y) * YCBI + delta + half) >> shift);
      const bool in = Y >= lower[0] && Y <= upper[0] && Cr >= lower[1] && Cr <= upper[1] && Cb >= lower[2] && Cb <= upper[2];
      mask[(size_t)y * cols + x] = in ? 255 : 0;                          // inRange
      const int G = gray15(b, g, r);                                      // cvtColor(BGR2GRAY): gray_shift = 15
      gray[(size_t)y * cols + x] = (uint8_t)G;
      hist[G]++;
    }
  }
  // equalizeHist
  const int total = rows * cols;
  int i0 = 0;
  while (!hist[i0]) ++i0;
  uint8_t lut[256] = {0};
  if (hist[i0] == total) {
    for (int i = 0; i < 256; i++) lut[i] = (uint8_t)i0;                   // dst.setTo(i)
  } else {
    const float scale = 255.f / (float)(total - hist[i0]);
    int sum = 0;
    lut[i0] = 0;
    for (int i = i0 + 1; i < 256; i++) {
      sum += hist[i];
      lut[i] = (uint8_t)sat8(cv_round((float)sum * scale));               // saturate_cast<uchar>(sum * scale)
    }
  }
  for (size_t i = 0; i < gray.size(); i++) gray[i] = lut[gray[i]] & mask[i];   // equalizeHist; gray &= mask
  if (mask_tap) memcpy(mask_tap, mask.data(), mask.size());
  if (gray_tap) memcpy(gray_tap, gray.data(), gray.size());
  std::vector<orc_rect> faces;
  int rc = detect_impl(c, gray.data(), rows, cols, cols, scale_factor, 2, 30, 30, 0, 0, faces, nullptr, nullptr, 0);
  if (rc) return rc;
  int n = 0;
  for (const orc_rect& r : faces) {
    // double m = norm(mean(croppedMask)); if (m/256 < 0.8) continue;
    long long sum = 0;
    for (int y = r.y; y < r.y + r.h; y++)
      for (int x = r.x; x < r.x + r.w; x++) sum += mask[(size_t)y * cols + x];
    const double mean0 = (double)sum * (1. / ((double)r.w * r.h));        // cv::mean: sum * (1./N)
    const double m = std::sqrt(mean0 * mean0);                            // cv::norm(Scalar): sqrt of sum of squares
    if (m / 256 < 0.8) continue;
    if (n >= cap) return -3;
    out[n++] = r;
  }
  *out_count = n;
  return 0;
}

// cv::resize(..., INTER_AREA), 8-bit, 1 or 3 interleaved channels, both axes shrinking (scale >= 1).
// Integer factors: resizeAreaFast_ (2x2: (a+b+c+d+2)>>2; else cvRound(sum * (1.f/area))).  Any other ratio:
// computeResizeAreaTab + resizeArea_<uchar, float>: per destination pixel the source cells it covers, weighted
// by the covered fraction (f64 table arithmetic rounded to f32), accumulated in f32 in table order -- per source
// row buf = sum_k S[sx_k] * alpha_k, then sum = beta_0 * buf_0 + beta_1 * buf_1 + ..., then
// saturate_cast<uchar>(sum) = cvRound.  One f32 rounding per written operation, no FMA.  [UPSTREAM-RECALL, unpinned]
namespace {
struct AreaTap { int si; float alpha; };
// taps of every destination index: ofs[d] .. ofs[d+1]
void area_table(int ssize, int dsize, double scale, std::vector<AreaTap>& tab, std::vector<int>& ofs) {
  tab.clear();
  ofs.assign(dsize + 1, 0);
  for (int dx = 0; dx < dsize; dx++) {
    ofs[dx] = (int)tab.size();
    const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    const double cell = std::min(scale, ssize - fsx1);
    int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
    sx2 = std::min(sx2, ssize - 1);
    sx1 = std::min(sx1, sx2);
    if (sx1 - fsx1 > 1e-3) tab.push_back(AreaTap{sx1 - 1, (float)((sx1 - fsx1) / cell)});
    for (int sx = sx1; sx < sx2; sx++) tab.push_back(AreaTap{sx, (float)(1.0 / cell)});
    if (fsx2 - sx2 > 1e-3) tab.push_back(AreaTap{sx2, (float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)});
  }
  ofs[dsize] = (int)tab.size();
}
}  // namespace

int orc_resize_area(const uint8_t* src, int rows, int cols, int channels, size_t step, uint8_t* dst, int drows, int dcols) {
  if (drows <= 0 || dcols <= 0 || drows > rows || dcols > cols || (channels != 1 && channels != 3)) return -4;
  if (rows % drows == 0 && cols % dcols == 0) {
    const int fy = rows / drows, fx = cols / dcols;
    const float scale = 1.f / (float)(fx * fy);
    for (int y = 0; y < drows; y++)
      for (int x = 0; x < dcols; x++)
        for (int ch = 0; ch < channels; ch++) {
          int sum = 0;
          for (int j = 0; j < fy; j++)
            for (int i = 0; i < fx; i++) sum += src[(size_t)(y * fy + j) * step + (size_t)(x * fx + i) * channels + ch];
          int v = (fx == 2 && fy == 2) ? (sum + 2) >> 2 : cv_round((float)sum * scale);
          dst[((size_t)y * dcols + x) * channels + ch] = (uint8_t)(v > 255 ? 255 : v);
        }
    return 0;
  }
  std::vector<AreaTap> xt, yt;
  std::vector<int> xo, yo;
  area_table(cols, dcols, (double)cols / dcols, xt, xo);
  area_table(rows, drows, (double)rows / drows, yt, yo);
  for (int y = 0; y < drows; y++)
    for (int x = 0; x < dcols; x++)
      for (int ch = 0; ch < channels; ch++) {
        float sum = 0.f;
        for (int j = yo[y]; j < yo[y + 1]; j++) {
          const uint8_t* S = src + (size_t)yt[j].si * step;
          float buf = 0.f;
          for (int k = xo[x]; k < xo[x + 1]; k++) {
            const float t = (float)S[(size_t)xt[k].si * channels + ch] * xt[k].alpha;
            buf = buf + t;
          }
          const float t = yt[j].alpha * buf;
          sum = (j == yo[y]) ? t : sum + t;
        }
        const int v = cv_round(sum);
        dst[((size_t)y * dcols + x) * channels + ch] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
      }
  return 0;
}

}  // extern "C"

// ================================================================== T2: OpenCV 2.4.x legacy semantics
// cvHaarDetectObjectsForROC with CV_HAAR_SCALE_IMAGE -- the generation the reference literally compiles against
// (src/tracker/associator.cc:3 includes opencv2/nonfree, removed in 3.0) and the branch its flag selects
// (src/tracker/detector.cc:30).  Restated from memory of modules/objdetect/src/haar.cpp and modules/imgproc/src/
// imgwarp.cpp of 2.4.x; the SCALAR C path is the definition (2.4's AVX build evaluates stumps in f32 and differs).
// [UPSTREAM-RECALL, parity unpinned]  Rules, tagged B.n:
//  B.1 factor stays f64: winSize = cvRound(win*factor), sz = cvRound(W/factor) (f64 division); loop ends when
//      sz - win + 1 <= 0 in either axis or winSize > maxSize; levels with winSize < minSize are skipped.
//  B.2 level image = cvResize(img, CV_INTER_LINEAR): f32 source coordinate (float)((d+0.5)*scale-0.5), 11-bit
//      coefficients (cvRound(c*2048) as short), horizontal pass in int, vertical
//      ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2; exact factor 2 in both axes -> (a+b+c+d+2)>>2.
//  B.3 sum int32, sqsum f64 (exact integers: window sums of squares < 2^32 are taken modulo 2^32 here).
//  B.4 ystep = factor > 2 ? 1 : 2 (strict); scanned positions x < sz.w - win_w, y < sz.h - win_h (the last
//      row / column of valid origins is NOT scanned); no x-skip; no low-variance gate.
//  B.5 per window: mean = S(equRect) * inv_area, var = SQ(equRect) * inv_area - mean^2 (f64), equRect = window
//      shrunk by 1; vnf = var >= 0 ? sqrt(var) : 1.
//  B.6 hidden feature weights: w_k = (float)(w_k_xml * inv_area) for k >= 1, w_0 = (float)(-sum_k>=1(w_k*area_k)/area_0).
//  B.7 node: t = thr(f32->f64) * vnf; sum = (f32)(S(r0) * w0) -> f64, += (f32)(S(r1) * w1), [+= r2]; branch / leaf by sum < t;
//      stage sum f64 of f32 leaves in order; stage threshold = (float)thr_xml - 0.0001f (f32 arithmetic); fail iff sum < thr.
//  B.8 stage TREE honoured (parent / next from the XML, child = first stage naming this one as parent): pass -> child,
//      fail -> next sibling of the nearest ancestor-or-self that has one, none -> reject; no child -> accept.
//  B.9 Rect(cvRound(x*factor), cvRound(y*factor), winSize) in f64; groupRectangles(max(minNeighbors,1), 0.2) unless
//      minNeighbors == 0 (raw candidates returned).
namespace {

// B.2: one axis of cv::resize(INTER_LINEAR) for 8-bit images, as resizeGeneric_ sets it up
struct LinTab24 { std::vector<int> ofs; std::vector<short> a0, a1; };
LinTab24 linear_table24(int ssize, int dsize) {
  LinTab24 t;
  t.ofs.resize(dsize); t.a0.resize(dsize); t.a1.resize(dsize);
  const double inv_scale = (double)dsize / ssize, scale = 1. / inv_scale;
  for (int d = 0; d < dsize; d++) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int sidx = (int)std::floor(f);
    f -= sidx;
    if (sidx < 0) { f = 0; sidx = 0; }
    if (sidx >= ssize - 1) { f = 0; sidx = ssize - 1; }
    t.ofs[d] = sidx;
    const float c0 = 1.f - f, c1 = f;
    t.a0[d] = (short)cv_round(c0 * 2048.f);          // saturate_cast<short>(c * INTER_RESIZE_COEF_SCALE)
    t.a1[d] = (short)cv_round(c1 * 2048.f);
  }
  return t;
}

void resize_linear_cv24(const uint8_t* src, int rows, int cols, size_t step, uint8_t* dst, int drows, int dcols) {
  if (drows == rows && dcols == cols) { for (int y = 0; y < rows; y++) memcpy(dst + (size_t)y * cols, src + (size_t)y * step, cols); return; }
  if (rows == 2 * drows && cols == 2 * dcols) {      // INTER_LINEAR at exactly 1/2 is routed to the area kernel
    for (int y = 0; y < drows; y++)
      for (int x = 0; x < dcols; x++) {
        const uint8_t* p = src + (size_t)(2 * y) * step + 2 * x;
        dst[(size_t)y * dcols + x] = (uint8_t)((p[0] + p[1] + p[step] + p[step + 1] + 2) >> 2);
      }
    return;
  }
  const LinTab24 tx = linear_table24(cols, dcols), ty = linear_table24(rows, drows);
  std::vector<int> h0(dcols), h1(dcols);
  for (int y = 0; y < drows; y++) {
    const int sy = ty.ofs[y], sy1 = std::min(sy + 1, rows - 1);
    const uint8_t *r0 = src + (size_t)sy * step, *r1 = src + (size_t)sy1 * step;
    for (int x = 0; x < dcols; x++) {
      const int sx = tx.ofs[x], sx1 = std::min(sx + 1, cols - 1);
      h0[x] = r0[sx] * tx.a0[x] + r0[sx1] * tx.a1[x];
      h1[x] = r1[sx] * tx.a0[x] + r1[sx1] * tx.a1[x];
    }
    const int b0 = ty.a0[y], b1 = ty.a1[y];
    for (int x = 0; x < dcols; x++)
      dst[(size_t)y * dcols + x] = (uint8_t)((((b0 * (h0[x] >> 4)) >> 16) + ((b1 * (h1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

struct Cv24Level { double factor; int w, h, ystep, nx, ny, win_w, win_h; };

void cv24_plan(const orc_cascade* c, int rows, int cols, double sf, int min_w, int min_h, int max_w, int max_h,
               std::vector<Cv24Level>& out) {
  out.clear();
  if (max_w == 0 || max_h == 0) { max_w = cols; max_h = rows; }
  for (double factor = 1;; factor *= sf) {                                 // B.1
    const int ww = cv_round(c->win_w * factor), wh = cv_round(c->win_h * factor);
    const int w = cv_round(cols / factor), h = cv_round(rows / factor);
    const int s1w = w - c->win_w + 1, s1h = h - c->win_h + 1;
    if (s1w <= 0 || s1h <= 0) break;
    if (ww > max_w || wh > max_h) break;
    if (ww < min_w || wh < min_h) continue;
    Cv24Level L{factor, w, h, factor > 2 ? 1 : 2, 0, 0, ww, wh};          // B.4
    const int lim_x = w - c->win_w, lim_y = h - c->win_h;                  // exclusive
    L.nx = (lim_x + L.ystep - 1) / L.ystep;
    L.ny = (lim_y + L.ystep - 1) / L.ystep;
    out.push_back(L);
  }
}

// hidden cascade of icvCreateHidHaarClassifierCascade + cvSetImagesForHaarClassifierCascade(scale 1)
struct Cv24Hid {
  std::vector<float> w;            // 3 per node (B.6)
  std::vector<float> stage_thr;    // B.7
  std::vector<int> child;          // B.8
  double inv_area;
};

Cv24Hid cv24_hidden(const orc_cascade* c) {
  Cv24Hid H;
  H.inv_area = 1. / ((double)(c->win_w - 2) * (c->win_h - 2));
  H.w.resize(c->nodes.size() * 3);
  for (size_t i = 0; i < c->nodes.size(); i++) {
    const Feature& F = c->feats[c->nodes[i].feat];
    double sum0 = 0, area0 = 0;
    for (int k = 0; k < 3; k++) {
      if (k > 0 && F.w[k] == 0.f) { H.w[i * 3 + k] = 0.f; continue; }     // absent third rect
      H.w[i * 3 + k] = (float)(F.w[k] * H.inv_area);
      if (k == 0) area0 = F.r[k][2] * F.r[k][3];
      else sum0 += H.w[i * 3 + k] * F.r[k][2] * F.r[k][3];
    }
    H.w[i * 3] = (float)(-sum0 / area0);
  }
  const float bias = 0.0001f;
  for (const Stage& s : c->stages) H.stage_thr.push_back((float)s.thr_raw - bias);
  H.child.assign(c->stages.size(), -1);
  for (size_t i = 0; i < c->stages.size(); i++) {
    const int p = c->stages[i].parent;
    if (p >= 0 && p < (int)c->stages.size() && H.child[p] < 0) H.child[p] = (int)i;
  }
  return H;
}

inline int box(const int32_t* I, int p, int x, int y, const int r[4]) {
  const int32_t* a = I + (size_t)(y + r[1]) * p + x + r[0];
  return a[0] - a[r[2]] - a[(size_t)r[3] * p] + a[(size_t)r[3] * p + r[2]];
}

// one stage at one window: B.7
inline bool cv24_stage(const orc_cascade* c, const Cv24Hid& H, int si, const int32_t* S, int p, int x, int y, double vnf) {
  const Stage& st = c->stages[si];
  double stage_sum = 0.0;
  for (int t = 0; t < st.ntrees; t++) {
    const Tree& T = c->trees[st.first_tree + t];
    int idx = 0;
    do {
      const int ni = T.first_node + idx;
      const Node& N = c->nodes[ni];
      const Feature& F = c->feats[N.feat];
      const double thr = (double)N.thr * vnf;
      double sum = (double)((float)box(S, p, x, y, F.r[0]) * H.w[ni * 3]);
      sum += (double)((float)box(S, p, x, y, F.r[1]) * H.w[ni * 3 + 1]);
      if (F.w[2] != 0.f) sum += (double)((float)box(S, p, x, y, F.r[2]) * H.w[ni * 3 + 2]);
      idx = sum < thr ? N.left : N.right;
    } while (idx > 0);
    stage_sum += (double)c->leaves[T.first_leaf - idx];
  }
  return !(stage_sum < (double)H.stage_thr[si]);
}

}  // namespace

extern "C" int orc_resize_linear_cv24(const uint8_t* src, int rows, int cols, size_t step, uint8_t* dst, int drows, int dcols) {
  if (rows <= 0 || cols <= 0 || drows <= 0 || dcols <= 0) return -2;
  resize_linear_cv24(src, rows, cols, step, dst, drows, dcols);
  return 0;
}

extern "C" int orc_plan_cv24(const orc_cascade* c, int rows, int cols, double scale_factor, int min_w, int min_h, int max_w, int max_h,
                             double* factors, int32_t* geom, int cap, int* nlevels) {
  std::vector<Cv24Level> L;
  cv24_plan(c, rows, cols, scale_factor, min_w, min_h, max_w, max_h, L);
  if ((int)L.size() > cap) return -3;
  for (size_t i = 0; i < L.size(); i++) {
    factors[i] = L[i].factor;
    int32_t* g = geom + 7 * i;
    g[0] = L[i].w; g[1] = L[i].h; g[2] = L[i].ystep; g[3] = L[i].nx; g[4] = L[i].ny; g[5] = L[i].win_w; g[6] = L[i].win_h;
  }
  *nlevels = (int)L.size();
  return 0;
}

// taps: int8 per scanned position, levels concatenated, [gy][gx] with nx x ny of orc_plan_cv24: number of stages PASSED along
// the path the window took through the stage tree (for a linear cascade: stages passed before the first failure; an accepted
// window counts every stage on its path), accepted windows additionally carry bit 6 (value | 64).
extern "C" int orc_detect_cv24(const orc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step, double scale_factor,
                               int min_neighbors, int min_w, int min_h, int max_w, int max_h, orc_rect* out, int cap, int* out_count,
                               orc_rect* raw, int raw_cap, int* raw_count, int8_t* taps, size_t taps_cap) {
  if (!c || !gray || !out_count || !(scale_factor > 1.0)) return -2;
  std::vector<Cv24Level> levels;
  cv24_plan(c, rows, cols, scale_factor, min_w, min_h, max_w, max_h, levels);
  const Cv24Hid H = cv24_hidden(c);
  std::vector<orc_rect> cands;
  std::vector<uint8_t> img;
  std::vector<int32_t> S, Q;
  size_t tap_at = 0;
  const int eq_w = c->win_w - 2, eq_h = c->win_h - 2;
  for (const Cv24Level& L : levels) {
    img.resize((size_t)L.w * L.h);
    resize_linear_cv24(gray, rows, cols, step, img.data(), L.h, L.w);          // B.2
    S.resize((size_t)(L.w + 1) * (L.h + 1)); Q.resize(S.size());
    integral_32s(img.data(), L.h, L.w, L.w, S.data(), Q.data());               // B.3
    const int p = L.w + 1;
    if (taps && tap_at + (size_t)L.nx * L.ny > taps_cap) return -3;
    for (int gy = 0; gy < L.ny; gy++)
      for (int gx = 0; gx < L.nx; gx++) {
        const int x = gx * L.ystep, y = gy * L.ystep;
        const int eq[4] = {1, 1, eq_w, eq_h};
        const double mean = box(S.data(), p, x, y, eq) * H.inv_area;           // B.5
        const double sq = (double)(uint32_t)box(Q.data(), p, x, y, eq);
        double vnf = sq * H.inv_area - mean * mean;
        vnf = vnf >= 0. ? std::sqrt(vnf) : 1.;
        int si = 0, passed = 0;
        bool accept = false;
        for (;;) {                                                              // B.8
          if (cv24_stage(c, H, si, S.data(), p, x, y, vnf)) {
            ++passed;
            if (H.child[si] < 0) { accept = true; break; }
            si = H.child[si];
          } else {
            while (si >= 0 && c->stages[si].next < 0) si = c->stages[si].parent;
            if (si < 0) break;
            si = c->stages[si].next;
          }
        }
        if (taps) taps[tap_at + (size_t)gy * L.nx + gx] = (int8_t)(std::min(passed, 63) | (accept ? 64 : 0));
        if (accept) cands.push_back(orc_rect{cv_round(x * L.factor), cv_round(y * L.factor), L.win_w, L.win_h});   // B.9
      }
    tap_at += (size_t)L.nx * L.ny;
  }
  if (raw_count) *raw_count = (int)cands.size();
  if (raw) {
    if ((int)cands.size() > raw_cap) return -3;
    for (size_t i = 0; i < cands.size(); i++) raw[i] = cands[i];
  }
  if (min_neighbors != 0) group_rects(cands, std::max(min_neighbors, 1), 0.2);
  if ((int)cands.size() > cap) return -3;
  for (size_t i = 0; i < cands.size(); i++) out[i] = cands[i];
  *out_count = (int)cands.size();
  return 0;
}

extern "C" {

int orc_eval_patch(const orc_cascade* c, const uint8_t* patch, double* margins) {
  std::vector<int32_t> sum((size_t)(c->win_w + 1) * (c->win_h + 1)), sq(sum.size());
  integral_32s(patch, c->win_h, c->win_w, c->win_w, sum.data(), sq.data());
  WinEval ev{c, sum.data(), sq.data(), c->win_w + 1, 1.f};
  if (!ev.set_window(0, 0)) return -1;
  return ev.run(0, 0, margins, margins != nullptr);
}

}  // extern "C"
