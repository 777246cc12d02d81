// api_model.cpp -- the cascade-level entry points (replacing cv::CascadeClassifier::load, reference src/file_io.cc:11) and
// the device tables a context derives from a model: per tile geometry one node table in the dense path's kind order and
// one in the finish's bank-spreading order, stage / tree / leaf records, the 16-bit-tile variants, the 2.4 stage tree.

#include <algorithm>
#include <cmath>
#include <cstring>

#include "hc_context.h"

namespace hc {

// LDS tile geometry + per-geometry corner offsets of every node
static constexpr size_t LDS_PER_CU = 160 * 1024;   // gfx950

static hc::TileGeom make_geom(const hc::Model& m, int ystep, int band_rows) {
  hc::TileGeom g{};
  g.cols = (hc::TILE_POS - 1) * ystep + m.win_w + 1;
  g.rows = (band_rows - 1) * ystep + m.win_h + 1;
  if (ystep == 1) { g.half = 0; g.pitch = g.cols; }
  else { g.half = (g.cols + 1) / 2; g.pitch = 2 * g.half; }
  return g;
}

static void build_nodes(const hc::Model& m, const hc::TileGeom& g, int ystep, std::vector<hc::DevNode>& out, bool p16 = false) {
  out.clear();
  out.reserve(m.nodes.size());
  auto word = [&](int dx, int dy) {
    if (p16) return (uint16_t)((dy * g.pitch + dx) * 2);   // 16-bit tile (scan step 1): 2-byte entries
    const int col = ystep == 1 ? dx : ((dx & 1) * g.half + (dx >> 1));
#ifndef HC_OFF_WORDS
    return (uint16_t)((dy * g.pitch + col) * 4);   // byte offsets: the kernel adds them to a byte base (kernels.hip hc_corner_addr)
#else
    return (uint16_t)(dy * g.pitch + col);
#endif
  };
  for (const hc::ModelTree& t : m.trees) {
    for (int k = 0; k < t.nnodes; k++) {
      const hc::ModelNode& n = m.nodes[t.first_node + k];
      hc::DevNode d{};
      for (int r = 0; r < 3; r++) {
        const int x = n.rect[r][0], y = n.rect[r][1], w = n.rect[r][2], h = n.rect[r][3];
        d.off[4 * r + 0] = word(x, y);
        d.off[4 * r + 1] = word(x + w, y);
        d.off[4 * r + 2] = word(x, y + h);
        d.off[4 * r + 3] = word(x + w, y + h);
        d.w[r] = n.weight[r];
      }
      d.thr = n.thr;
      if (m.is_stump()) {
        const float l = m.leaves[t.first_leaf + 0], r = m.leaves[t.first_leaf + 1];
        memcpy(&d.a, &l, 4);
        memcpy(&d.b, &r, 4);
      } else {
        d.a = (uint32_t)n.left;
        d.b = (uint32_t)n.right;
      }
      out.push_back(d);
    }
  }
}

// corner-sharing kind of a stump (device_types.h NodeKind): compares corner coordinates
static int node_kind(const hc::ModelNode& n) {
  auto cx = [&](int r, int k) { return n.rect[r][0] + ((k & 1) ? n.rect[r][2] : 0); };
  auto cy = [&](int r, int k) { return n.rect[r][1] + ((k & 2) ? n.rect[r][3] : 0); };
  auto same = [&](int r1, int k1, int r2, int k2) { return cx(r1, k1) == cx(r2, k2) && cy(r1, k1) == cy(r2, k2); };
  if (n.weight[2] == 0.f) {
    if (same(1, 2, 0, 2) && same(1, 3, 0, 3)) return hc::NK_BOTTOM;
    if (same(1, 1, 0, 1) && same(1, 3, 0, 3)) return hc::NK_RIGHT;
    if (same(1, 0, 0, 0) && same(1, 2, 0, 2)) return hc::NK_LEFT;
    if (same(1, 0, 0, 0) && same(1, 1, 0, 1)) return hc::NK_TOP;
    return hc::NK_NONE2;
  }
  if (same(1, 1, 0, 1) && same(2, 1, 1, 2) && same(2, 2, 0, 2)) return hc::NK_ANTIDIAG;
  if (same(1, 0, 0, 0) && same(2, 0, 1, 3) && same(2, 3, 0, 3)) return hc::NK_DIAG;
  return hc::NK_NONE3;
}

// True when every partial sum of the stage's leaf values, in any order, is exactly
// representable in f64: all leaves are integer multiples of q = 2^emin and sum |leaf| / q < 2^53.
// The sequential f64 accumulation of upstream then equals the mathematical sum,
// and so does any re-association of it.
static bool stage_sum_is_exact(const hc::Model& m, const hc::ModelStage& s) {
  int emin = INT32_MAX;
  const int t0 = s.first_tree, t1 = s.first_tree + s.ntrees;
  const int l0 = m.trees[t0].first_leaf, l1 = m.trees[t1 - 1].first_leaf + m.trees[t1 - 1].nnodes + 1;
  for (int i = l0; i < l1; i++) {
    const float v = m.leaves[i];
    if (v == 0.f) continue;
    if (!std::isfinite(v)) return false;
    int e;
    std::frexp(v, &e);              // |v| = f * 2^e, f in [0.5, 1): the f32 ulp is 2^(e-24)
    emin = std::min(emin, e - 24);
  }
  if (emin == INT32_MAX) return true;
  long double units = 0;            // 64-bit mantissa: exact for these magnitudes
  for (int i = l0; i < l1; i++) {
    const long double u = std::ldexp((long double)std::fabs(m.leaves[i]), -emin);
    if (u > 0x1p60L) return false;
    units += u;
  }
  // The device also forms bsum = sum(right) and adds [v < thr] * (left - right) per stump: every partial sum of that is
  // bounded by sum |right| + sum (|left| + |right|) <= 2 * sum |leaf|, so twice the plain bound keeps all of it exact.
  // The sliced finish (cascade_body.inc wave_scan_f64) runs one prefix scan over the partial sums of up to 64 windows of
  // a stage: 64 * sum (|left| + |right|), plus bsum, stays below 128 * sum |leaf|.  (The four reference cascades have
  // 7.6 to 15 bits to spare on the plain bound.)
  return 128 * units < 0x1p53L;
}

// True when upstream's f32 feature value w0*(float)S0 + w1*(float)S1 [+ w2*(float)S2] is an exact integer computation for
// every node: w0 == -1, w1 and w2 integers, and the largest magnitude any partial value can take (rect sums are at most
// 255 * area) stays below 2^24.  The kernel then forms it with 24-bit integer multiply-adds and converts once.
bool weights_are_small_integers(const hc::Model& m) {
  for (const hc::ModelNode& n : m.nodes) {
    if (n.weight[0] != -1.f) return false;
    double bound = 0;
    for (int k = 0; k < 3; k++) {
      const float w = n.weight[k];
      if (w != std::nearbyintf(w) || std::fabs(w) > 127.f) return false;
      bound += std::fabs((double)w) * 255.0 * n.rect[k][2] * n.rect[k][3];
    }
    if (!(bound < 8388608.0)) return false;     // 2^23: operands and results of v_mul_i32_i24 / the f32 mantissa
  }
  return true;
}

// Sparse finish (one window per wavefront, lane = stump): a wave-instruction reads corner slot j of 64 different stumps of
// the stage, so which lanes of a 32-lane group collide on an LDS bank is decided by the stumps' own offsets (the window's
// base shifts all of them alike).  The stage's f64 sum is order-free, so its stumps may be dealt to the 32-lane groups in
// any order: greedily, each next stump is the one that raises the group's LDS cycles (sum over the corner slots of the
// fullest bank) least, ties by the fewest bank mates.  In file order a slot costs about 3.5 cycles per group of 32.
static void order_stage_for_banks(std::vector<hc::DevNode>& nodes, int first, int count) {
  const auto bank = [](const hc::DevNode& n, int j) {
#ifndef HC_OFF_WORDS
    return (int)((n.off[j] >> 2) & 31u);
#else
    return (int)(n.off[j] & 31u);
#endif
  };
  const auto nslots = [](const hc::DevNode& n) { uint32_t w2; memcpy(&w2, &n.w[2], 4); return w2 ? 12 : 8; };
  std::vector<hc::DevNode> in(nodes.begin() + first, nodes.begin() + first + count), out;
  std::vector<char> used(count, 0);
  out.reserve(count);
  for (int done = 0; done < count;) {
    int cnt[12][32] = {};
    int mx[12] = {};
    const int gsz = std::min(32, count - done);
    for (int k = 0; k < gsz; k++) {
      int best = -1, best_rise = 1 << 30, best_mates = 1 << 30;
      for (int t = 0; t < count; t++) {
        if (used[t]) continue;
        int rise = 0, mates = 0;
        const int ns = nslots(in[t]);
        for (int j = 0; j < ns; j++) {
          const int c0 = cnt[j][bank(in[t], j)];
          mates += c0;
          if (c0 + 1 > mx[j]) rise++;
        }
        if (rise < best_rise || (rise == best_rise && mates < best_mates)) { best = t; best_rise = rise; best_mates = mates; }
      }
      used[best] = 1;
      const int ns = nslots(in[best]);
      for (int j = 0; j < ns; j++) {
        const int b = bank(in[best], j);
        mx[j] = std::max(mx[j], ++cnt[j][b]);
      }
      out.push_back(in[best]);
      done++;
    }
  }
  std::copy(out.begin(), out.end(), nodes.begin() + first);
}

// FNV-1a over the fields that decide every result of the cascade, in the order tools/gen_spec.py model_hash walks them:
// the compiled early stages (csrc/spec/*.inc) carry the hash of the cascade they were generated from.
unsigned long long model_hash(const hc::Model& m) {
  unsigned long long h = 0xcbf29ce484222325ull;
  auto feed = [&](const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 0x100000001b3ull;
  };
  const int32_t head[3] = {m.win_w, m.win_h, (int32_t)m.stages.size()};
  feed(head, sizeof head);
  for (const hc::ModelStage& s : m.stages) { feed(&s.ntrees, 4); feed(&s.thr, 4); }
  for (const hc::ModelTree& t : m.trees) feed(&t.nnodes, 4);
  for (const hc::ModelNode& n : m.nodes) { feed(n.rect, 12); feed(n.weight, 12); feed(&n.thr, 4); feed(&n.left, 4); feed(&n.right, 4); }
  feed(m.leaves.data(), m.leaves.size() * 4);
  return h;
}
void free_model_tables(hc_context* c) {
  void* ptrs[] = {c->d_nodes[0], c->d_nodes[1], c->d_nodes_sparse[0], c->d_nodes_sparse[1], c->d_stages, c->d_trees, c->d_leaves,
                  c->d_kinds, c->d_cvnodes, c->d_cvtrees, c->d_cvstages, c->d_tree_pass, c->d_tree_fail, c->d_tree_order,
                  c->d_nodes_p16, c->d_nodes_sparse_p16, c->d_kinds_p16, c->d_stages_seg, c->d_kinds_seg};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  c->d_tree_pass = c->d_tree_fail = c->d_tree_order = nullptr;
  c->d_stages_seg = nullptr; c->d_kinds_seg = nullptr; c->cv_seg = false;
  c->d_nodes_p16 = c->d_nodes_sparse_p16 = nullptr; c->d_kinds_p16 = nullptr; c->p16 = false; c->spec_id_p16 = 0;
  c->d_nodes[0] = c->d_nodes[1] = c->d_nodes_sparse[0] = c->d_nodes_sparse[1] = nullptr;
  c->d_stages = nullptr; c->d_trees = nullptr; c->d_leaves = nullptr; c->d_kinds = nullptr;
  c->d_cvnodes = nullptr; c->d_cvtrees = nullptr; c->d_cvstages = nullptr;
}

// every corner offset of the 16-bit tile fits the 16-bit offset fields (trivially, for windows this build accepts)
static bool node_kinds_fit_p16(const hc::Model& m) { return m.win_w <= 40 && m.win_h <= 40; }

// Checkerboard stump whose rect 0 -- the 2 x 2 grid -- has more than 257 pixels: rects 1 and 2 are two diagonal quadrants with
// weight 2, so 2 (S1 + S2) - S0 = S1 + S2 - S3 - S4 with the other two quadrants, all four of at most 257 pixels, and all their
// corners among the nine the stump reads anyway.  1: rects 1 / 2 are the top-left / bottom-right quadrants (NK_DIAG), 2: top-right
// / bottom-left (NK_ANTIDIAG), 0: not of this form.  The device evaluates such a node in "checker form" (cascade_body.inc).
static int p16_checker_form(const hc::ModelNode& n) {
  if (n.weight[0] != -1.f || n.weight[1] != 2.f || n.weight[2] != 2.f) return 0;
  const int kind = node_kind(n);
  if (kind != hc::NK_DIAG && kind != hc::NK_ANTIDIAG) return 0;
  const int x = n.rect[0][0], y = n.rect[0][1], w = n.rect[0][2], h = n.rect[0][3];
  const int xm = kind == hc::NK_DIAG ? n.rect[2][0] : n.rect[1][0];
  const int ym = kind == hc::NK_DIAG ? n.rect[2][1] : n.rect[1][1] + n.rect[1][3];
  if (xm <= x || xm >= x + w || ym <= y || ym >= y + h) return 0;
  const int qa[4] = {(xm - x) * (ym - y), (x + w - xm) * (ym - y), (xm - x) * (y + h - ym), (x + w - xm) * (y + h - ym)};
  // rects 1 and 2 must BE those quadrants (node_kind only compares three corner pairs)
  const int a1 = n.rect[1][2] * n.rect[1][3], a2 = n.rect[2][2] * n.rect[2][3];
  if (kind == hc::NK_DIAG ? (a1 != qa[0] || a2 != qa[3]) : (a1 != qa[1] || a2 != qa[2])) return 0;
  for (int q : qa) if (q > 257) return 0;
  return kind == hc::NK_DIAG ? 1 : 2;
}

// The cascade re-expressed so that every rect sum the 16-bit tile forms is of a rect of at most 257 pixels (rect sums modulo
// 2^16 are then exact); the feature value -- an integer expression, weights_are_small_integers -- is unchanged.  A rect 0
// (weight -1) of more than 257 pixels
//   - in a two-rect stump is cut into two halves, the second in the node's free third slot with the same weight; when the
//     halves are still too large (24 x 24 windows: up to 576 pixels) and rect 1 spans rect 0's full height or width,
//     w1 S1 - S0 = (w1 - 1) S1 - (the strips of rect 0 beside rect 1): rect 0 becomes one strip, rect 2 the other (weight -1);
//   - in a three-rect stump must be a checkerboard's grid (p16_checker_form): the node stays as it is and is flagged.
// False when some node fits none of these.
static bool p16_model(const hc::Model& m, hc::Model& out) {
  out = m;
  for (hc::ModelNode& n : out.nodes) {
    int big = -1;
    for (int r = 0; r < 3; r++)
      if (n.weight[r] != 0.f && n.rect[r][2] * n.rect[r][3] > 257) { if (big >= 0) return false; big = r; }
    if (big < 0) continue;
    if (big != 0 || n.weight[0] != -1.f) return false;
    if (n.weight[2] != 0.f) { if (p16_checker_form(n)) continue; return false; }
    const int x = n.rect[0][0], y = n.rect[0][1], w = n.rect[0][2], h = n.rect[0][3];
    const int hh = h / 2, wh = w / 2;
    if (h >= w ? (w * hh <= 257 && w * (h - hh) <= 257) : (wh * h <= 257 && (w - wh) * h <= 257)) {
      if (h >= w) {
        n.rect[0][3] = (int8_t)hh;
        n.rect[2][0] = (int8_t)x; n.rect[2][1] = (int8_t)(y + hh); n.rect[2][2] = (int8_t)w; n.rect[2][3] = (int8_t)(h - hh);
      } else {
        n.rect[0][2] = (int8_t)wh;
        n.rect[2][0] = (int8_t)(x + wh); n.rect[2][1] = (int8_t)y; n.rect[2][2] = (int8_t)(w - wh); n.rect[2][3] = (int8_t)h;
      }
      n.weight[2] = -1.f;
      continue;
    }
    // complement form
    const int x1 = n.rect[1][0], y1 = n.rect[1][1], w1 = n.rect[1][2], h1 = n.rect[1][3];
    int sa[4], sb[4];                       // the strips of rect 0 beside rect 1 (either may be empty)
    if (y1 == y && h1 == h && x <= x1 && x1 + w1 <= x + w) {
      sa[0] = x; sa[1] = y; sa[2] = x1 - x; sa[3] = h;
      sb[0] = x1 + w1; sb[1] = y; sb[2] = x + w - x1 - w1; sb[3] = h;
    } else if (x1 == x && w1 == w && y <= y1 && y1 + h1 <= y + h) {
      sa[0] = x; sa[1] = y; sa[2] = w; sa[3] = y1 - y;
      sb[0] = x; sb[1] = y1 + h1; sb[2] = w; sb[3] = y + h - y1 - h1;
    } else {
      return false;
    }
    if (sa[2] * sa[3] == 0) { for (int k = 0; k < 4; k++) { sa[k] = sb[k]; sb[k] = 0; } }
    if (sa[2] * sa[3] == 0 || sa[2] * sa[3] > 257 || sb[2] * sb[3] > 257 || w1 * h1 > 257 || n.weight[1] - 1.f == 0.f) return false;
    for (int k = 0; k < 4; k++) { n.rect[0][k] = (int8_t)sa[k]; n.rect[2][k] = (int8_t)sb[k]; }
    n.weight[1] -= 1.f;
    n.weight[2] = sb[2] * sb[3] ? -1.f : 0.f;
  }
  return true;
}

// Node tables of the 16-bit-tile build (scan step 1) from the split model: same stages, same leaf words, same ordering
// rules (kind groups, bank-spreading order of the sparse copy) as the standard tables below.
static int upload_p16_tables(hc_context* c, const hc::Model& m16, const std::vector<hc::DevStage>& st) {
  std::vector<hc::DevNode> nd;
  build_nodes(m16, c->geom_p16, 1, nd, true);
  // checker form: the node's unused w0 word (w0 == -1 is a premise of the integer build) carries 1 / 2 (p16_checker_form)
  for (size_t i = 0; i < nd.size(); i++) {
    const hc::ModelNode& mn = m16.nodes[i];
    if (mn.weight[2] != 0.f && mn.rect[0][2] * mn.rect[0][3] > 257) nd[i].w[0] = (float)p16_checker_form(mn);
  }
  std::vector<hc::DevStageKinds> kinds;
  for (size_t si = 0; si < m16.stages.size(); si++) {
    const hc::ModelStage& s = m16.stages[si];
    const int first = st[si].first;
    for (int i = 0; i < s.ntrees; i++) {
      const hc::ModelTree& t = m16.trees[s.first_tree + i];
      const double diff = (double)m16.leaves[t.first_leaf] - (double)m16.leaves[t.first_leaf + 1];
      uint64_t bits;
      memcpy(&bits, &diff, 8);
      nd[t.first_node].a = (uint32_t)bits;
      nd[t.first_node].b = (uint32_t)(bits >> 32);
    }
    std::vector<int> order(s.ntrees);
    for (int i = 0; i < s.ntrees; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return node_kind(m16.nodes[first + x]) < node_kind(m16.nodes[first + y]); });
    std::vector<hc::DevNode> tmp(s.ntrees);
    hc::DevStageKinds kd{};
    int cnt[hc::NK_COUNT] = {0};
    for (int i = 0; i < s.ntrees; i++) { tmp[i] = nd[first + order[i]]; cnt[node_kind(m16.nodes[first + i])]++; }
    std::copy(tmp.begin(), tmp.end(), nd.begin() + first);
    int acc = 0;
    for (int k2 = 0; k2 < hc::NK_COUNT; k2++) { acc += cnt[k2]; kd.end[k2] = (uint16_t)acc; }
    kinds.push_back(kd);
  }
  for (hc::DevNode& d : nd)
    for (int k = 0; k < 3; k++) {
      const int32_t wi = (int32_t)d.w[k];
      memcpy(&d.w[k], &wi, 4);
    }
  std::vector<hc::DevNode> ns = nd;
  if (c->k.bank_order)
    for (const hc::DevStage& d : st) order_stage_for_banks(ns, d.first, d.count);
  int rc;
  if ((rc = dev_upload(&c->d_nodes_p16, nd)) || (rc = dev_upload(&c->d_nodes_sparse_p16, ns)) || (rc = dev_upload(&c->d_kinds_p16, kinds))) return rc;
  return HC_OK;
}

int upload_model(hc_context* c) {
  ensure_env_plugins();
  free_model_tables(c);
  const hc::Model& m = c->casc->model;
  const bool cv24 = c->semantics == 1;
  c->cv_chain = true;
  c->cv_tree_tiled = false;
  std::vector<hc::DevNode> nd[2];
  for (int k = 0; k < 2; k++) {
    const int ystep = k + 1;
    c->geom[k] = make_geom(m, ystep, c->k.band_rows[k]);
    const size_t words = ((size_t)c->geom[k].pitch * c->geom[k].rows + 1) & ~(size_t)1;   // even: 8-byte aligned LDS regions behind the tile
#ifndef HC_OFF_WORDS
    if (words * 4 > 65535) return hc::set_err(HC_ERR_UNSUPPORTED, "window too large for 16-bit LDS byte offsets");
#else
    if (words > 65535) return hc::set_err(HC_ERR_UNSUPPORTED, "window too large for 16-bit LDS offsets");
#endif
    c->tile_words[k] = (int)words;
    // k_cascade LDS layout: tile | vnf[band_rows*64] f32 | 2 queues x band_rows*64 u16 | 3 counters (+pad)
    const size_t vnf_bytes = cv24 ? 8 : 4;      // 2.4 semantics: f64 variance factors
#ifndef HC_PLAIN_QUEUES
    // k_cascade LDS layout: tile | vnf[band_rows*64] f32 | bank matrix + survivor list, band_rows*64 u16 each | 32 counter words + ticket
    c->lds_bytes[k] = words * 4 + (size_t)c->k.band_rows[k] * hc::TILE_POS * (vnf_bytes + 2 * 2) + 136;
#else
    c->lds_bytes[k] = words * 4 + (size_t)c->k.band_rows[k] * hc::TILE_POS * (vnf_bytes + 2 * 2) + 16;
#endif
    build_nodes(m, c->geom[k], ystep, nd[k]);
    if (cv24)       // B.6: the hidden cascade's area-normalised weights (the node records are in model order here)
      for (size_t i = 0; i < nd[k].size(); i++) hc::cv24_weights(m, m.nodes[i], nd[k][i].w);
  }
  std::vector<hc::DevStage> st;
  struct PairWords { int node; uint32_t a, b; };
  std::vector<PairWords> pair_words;                 // dense node table only: the sparse finish keeps the child links
  std::vector<std::pair<int, int>> pair_left;        // tree -> DevTree::pad
  std::vector<hc::DevTree> tr;
  std::vector<hc::DevStageKinds> kinds;
  for (const hc::ModelStage& s : m.stages) {
    hc::DevStage d{};
    d.first = m.is_stump() ? m.trees[s.first_tree].first_node : s.first_tree;
    d.count = s.ntrees;
    d.thr = cv24 ? hc::cv24_stage_thr(s) : s.thr;
    d.exact = stage_sum_is_exact(m, s) ? 1 : 0;
    d.bsum = 0.0;
    if (m.is_stump() && d.exact) {
      // leaf accumulation as bsum + sum([v < thr] * (left - right)), all in exact f64 (kernels.hip stump_acc)
      long double bs = 0;
      for (int i = 0; i < s.ntrees; i++) {
        const hc::ModelTree& t = m.trees[s.first_tree + i];
        const double left = (double)m.leaves[t.first_leaf], right = (double)m.leaves[t.first_leaf + 1];
        const double diff = left - right;
        bs += right;
        uint64_t bits;
        memcpy(&bits, &diff, 8);
        for (int g = 0; g < 2; g++) {
          nd[g][t.first_node].a = (uint32_t)bits;
          nd[g][t.first_node].b = (uint32_t)(bits >> 32);
        }
      }
      d.bsum = (double)bs;
    }
    if (!m.is_stump() && d.exact && c->k.pair_form) {
      // pair form of a tree stage (device_types.h DevStage::pair): every tree a root with one leaf branch and one
      // second node whose branches are both leaves
      bool ok = true;
      for (int i = 0; i < s.ntrees && ok; i++) {
        const hc::ModelTree& t = m.trees[s.first_tree + i];
        if (t.nnodes != 2) { ok = false; break; }
        const hc::ModelNode &r = m.nodes[t.first_node], &q = m.nodes[t.first_node + 1];
        ok = ((r.left == 1 && r.right <= 0) || (r.right == 1 && r.left <= 0)) && q.left <= 0 && q.right <= 0;
      }
      if (ok) {
        long double ps = 0;
        for (int i = 0; i < s.ntrees; i++) {
          const hc::ModelTree& t = m.trees[s.first_tree + i];
          const hc::ModelNode &r = m.nodes[t.first_node], &q = m.nodes[t.first_node + 1];
          const bool leftnode = r.left == 1;
          const double other = (double)m.leaves[t.first_leaf - (leftnode ? r.right : r.left)];
          const double la = (double)m.leaves[t.first_leaf - q.left], lb = (double)m.leaves[t.first_leaf - q.right];
          const double d1 = lb - other, d2 = la - lb;          // exact: differences of leaves of one stage (stage_sum_is_exact)
          ps += other;
          uint64_t b1, b2;
          memcpy(&b1, &d1, 8);
          memcpy(&b2, &d2, 8);
          pair_words.push_back({t.first_node, (uint32_t)b1, (uint32_t)(b1 >> 32)});
          pair_words.push_back({t.first_node + 1, (uint32_t)b2, (uint32_t)(b2 >> 32)});
          pair_left.push_back({s.first_tree + i, leftnode ? 1 : 0});
        }
        d.pair = 1;
        d.psum = (double)ps;
      }
    }
    st.push_back(d);
    if (m.is_stump()) {
      // Stumps of a stage may be regrouped by kind only when the stage's f64 sum is exact in any order
      // (d.exact).  Any other stage keeps file order and is evaluated by the ordered generic path on the
      // device (its end[] stays zero and is never read).
      hc::DevStageKinds kd{};
      if (s.ntrees > 65535) return hc::set_err(HC_ERR_UNSUPPORTED, "stage too long");
      if (d.exact) {
        std::vector<int> order(s.ntrees);
        for (int i = 0; i < s.ntrees; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
          return node_kind(m.nodes[d.first + x]) < node_kind(m.nodes[d.first + y]);
        });
        for (int g = 0; g < 2; g++) {
          std::vector<hc::DevNode> tmp(s.ntrees);
          for (int i = 0; i < s.ntrees; i++) tmp[i] = nd[g][d.first + order[i]];
          std::copy(tmp.begin(), tmp.end(), nd[g].begin() + d.first);
        }
        int cnt[hc::NK_COUNT] = {0};
        for (int i = 0; i < s.ntrees; i++) cnt[node_kind(m.nodes[d.first + i])]++;
        int acc = 0;
        for (int k2 = 0; k2 < hc::NK_COUNT; k2++) { acc += cnt[k2]; kd.end[k2] = (uint16_t)acc; }
      }
      kinds.push_back(kd);
    }
  }
  // exact == 2: this stage AND every later one have an order-free sum (the kernel may then finish a
  // tile's last few windows without re-queuing them between stages)
  for (int i = (int)st.size() - 1, all = 1; i >= 0; i--) {
    all = all && st[i].exact;
    if (all) st[i].exact = 2;
  }
  for (const hc::ModelTree& t : m.trees) tr.push_back(hc::DevTree{t.first_node, t.nnodes, t.first_leaf, 0});
  for (const auto& pl : pair_left) tr[pl.first].pad = pl.second;
  c->int_weights = !cv24 && c->k.allow_int_weights && weights_are_small_integers(m);
  c->spec_id = c->int_weights && c->k.allow_spec ? hc::spec_lookup(model_hash(m), m.win_w, m.win_h, nullptr) : 0;
  int cv24_spec_stages = 0;
  if (cv24 && c->k.allow_spec)      // compiled early stages with the 2.4 node arithmetic (csrc/spec/spec_*24.hip)
    c->spec_id = hc::spec_lookup(model_hash(m), m.win_w, m.win_h, &cv24_spec_stages, false, true);
  if (cv24) {
    c->cv_chain = hc::cv24_is_chain(m);
    if (!c->cv_chain) {           // a real stage tree: tables of the window-at-a-time kernel
      std::vector<hc::Cv24Node> cn;
      std::vector<hc::Cv24Tree> ct;
      std::vector<hc::Cv24Stage> cs;
      int rc0 = hc::cv24_tree_tables(m, cn, ct, cs);
      if (rc0 != HC_OK) return rc0;
      if ((rc0 = dev_upload(&c->d_cvnodes, cn)) || (rc0 = dev_upload(&c->d_cvtrees, ct)) || (rc0 = dev_upload(&c->d_cvstages, cs))) return rc0;
      // A stage tree of the form "a spine, then up to four chains tried one after the other" (haarcascade_frontalface_alt_tree:
      // stages 0-4, then 5 -> 7 -> .. -> 39 and 6 -> 8 -> .. -> 46) runs on the chain kernel itself, segment after
      // segment (cascade_body.inc "alternative chains"): the stage tables once more in evaluation order.
      c->cv_seg = false;
      if (m.is_stump() && c->k.cv24_seg) {
        std::vector<int> order, heads;
        bool ok = !cs.empty() && cs[0].parent < 0 && cs[0].next < 0;
        for (int sidx = 0; ok;) {
          order.push_back(sidx);
          const int ch = cs[sidx].child;
          if (ch < 0) { ok = false; break; }                        // a plain chain (cv24_is_chain would have said so)
          if (cs[ch].next >= 0) {                                   // the branch point: the children of the spine's last stage
            for (int h = ch; h >= 0; h = cs[h].next) heads.push_back(h);
            break;
          }
          sidx = ch;
        }
        const int spine_len = (int)order.size();
        int seg_end[4] = {0, 0, 0, 0};
        ok = ok && !heads.empty() && heads.size() <= 4 && spine_len >= std::max(2, c->k.row_stages);
        for (size_t g = 0; ok && g < heads.size(); g++) {
          for (int t = heads[g]; t >= 0;) {
            order.push_back(t);
            const int ch = cs[t].child;
            if (ch >= 0 && cs[ch].next >= 0) { ok = false; break; }   // a branch inside a chain: the general sweep takes it
            t = ch;
          }
          seg_end[g] = (int)order.size();
        }
        std::vector<char> seen(cs.size(), 0);
        for (int v : order) { if (v < 0 || v >= (int)cs.size() || seen[v]) ok = false; else seen[v] = 1; }
        ok = ok && order.size() == cs.size() && order.size() < 64;   // (6 bits of a list entry count the stages passed)
        // (compiled stages are generated in FILE order: a unit applies only while the evaluation order keeps it)
        for (int i = 0; ok && i < cv24_spec_stages; i++)
          if (order[i] != i) { c->spec_id = 0; break; }
        if (ok) {
          std::vector<hc::DevStage> sst(order.size());
          std::vector<hc::DevStageKinds> skd(order.size());
          for (size_t i = 0; i < order.size(); i++) { sst[i] = st[order[i]]; skd[i] = kinds[order[i]]; }
          for (int i = (int)sst.size() - 1, all = 1; i >= 0; i--) {   // exact == 2: this stage and every later one IN THIS ORDER
            if (sst[i].exact) sst[i].exact = 1;
            all = all && sst[i].exact;
            if (all) sst[i].exact = 2;
          }
          if ((rc0 = dev_upload(&c->d_stages_seg, sst)) || (rc0 = dev_upload(&c->d_kinds_seg, skd))) return rc0;
          c->cv_seg = true;
          for (int k = 0; k < 2; k++) c->lds_bytes[k] += (size_t)c->k.band_rows[k] * hc::TILE_POS * 2;   // the retry list, behind the tile
          c->seg_n = (int)heads.size();
          c->seg_spine = spine_len;
          for (int g = 0; g < 4; g++) c->seg_end[g] = seg_end[g];
        }
      }
      if (!c->cv_seg) c->spec_id = 0;     // the preorder sweep and the window-at-a-time kernel evaluate every stage generically
      // the same tree for the tiled kernel (stump cascades): per stage where a window goes when it passes (first child,
      // or accepted) and when it fails (next sibling of the nearest ancestor-or-self that has one, or rejected)
      c->cv_tree_tiled = m.is_stump() && cs.size() <= 250;
      if (c->cv_tree_tiled) {
        std::vector<uint8_t> tp(cs.size()), tf(cs.size());
        for (size_t i = 0; i < cs.size(); i++) {
          tp[i] = cs[i].child >= 0 ? (uint8_t)cs[i].child : (uint8_t)254;
          int si = (int)i;
          while (si >= 0 && cs[si].next < 0) si = cs[si].parent;
          tf[i] = si >= 0 ? (uint8_t)cs[si].next : (uint8_t)255;
        }
        // preorder: a stage, its child subtree, then its next sibling (roots are chained through `next` as well)
        std::vector<uint8_t> order;
        std::vector<int> stack;
        for (int root = 0; root >= 0 && root < (int)cs.size(); root = -1) stack.push_back(root);
        std::vector<char> seen(cs.size(), 0);
        while (!stack.empty()) {
          const int v = stack.back();
          stack.pop_back();
          if (v < 0 || seen[v]) continue;
          seen[v] = 1;
          order.push_back((uint8_t)v);
          if (cs[v].next >= 0) stack.push_back(cs[v].next);       // visited after the whole child subtree
          if (cs[v].child >= 0) stack.push_back(cs[v].child);
        }
        for (size_t i = 0; i < cs.size(); i++)                      // stages no walk reaches: harmless, keep the table complete
          if (!seen[i]) order.push_back((uint8_t)i);
        if ((rc0 = dev_upload(&c->d_tree_pass, tp)) || (rc0 = dev_upload(&c->d_tree_fail, tf)) || (rc0 = dev_upload(&c->d_tree_order, order)))
          return rc0;
      }
    }
  }
  if (c->int_weights)
    for (int g = 0; g < 2; g++)
      for (hc::DevNode& d : nd[g])
        for (int k = 0; k < 3; k++) {
          const int32_t wi = (int32_t)d.w[k];
          memcpy(&d.w[k], &wi, 4);
        }
  int rc;
  std::vector<hc::DevNode> ns[2] = {nd[0], nd[1]};   // the sparse finish reads its own copy of the node records (trees: with the child links)
  for (int g = 0; g < 2; g++)
    for (const PairWords& pw : pair_words) { nd[g][pw.node].a = pw.a; nd[g][pw.node].b = pw.b; }
  if ((rc = dev_upload(&c->d_nodes[0], nd[0]))) return rc;
  if ((rc = dev_upload(&c->d_nodes[1], nd[1]))) return rc;
  {
    // ... stump stages with an order-free sum in bank-spreading order
    if (m.is_stump() && c->k.bank_order)
      for (int g = 0; g < 2; g++)
        for (const hc::DevStage& d : st)
          if (d.exact) order_stage_for_banks(ns[g], d.first, d.count);
    if ((rc = dev_upload(&c->d_nodes_sparse[0], ns[0]))) return rc;
    if ((rc = dev_upload(&c->d_nodes_sparse[1], ns[1]))) return rc;
  }
  if ((rc = dev_upload(&c->d_stages, st))) return rc;
  if (m.is_stump() && (rc = dev_upload(&c->d_kinds, kinds))) return rc;
  if ((rc = dev_upload(&c->d_trees, tr))) return rc;
  if ((rc = dev_upload(&c->d_leaves, m.leaves))) return rc;
  // ---- 16-bit-tile build of the scan-step-1 launch, when the cascade and the launch shape qualify
  const bool p16_shape = (c->k.band_rows[0] == 16 && c->k.band_waves[0] == 4) || (c->k.band_rows[0] == 32 && c->k.band_waves[0] == 8);
  if (!cv24 && c->k.allow_p16 && c->k.allow_spec && c->int_weights && m.is_stump() && p16_shape &&
      c->k.row_stages <= 2 && !st.empty() && st[0].exact == 2) {
    const int prows = c->k.band_rows[0];
    int pst = 0;
    const int id = hc::spec_lookup(model_hash(m), m.win_w, m.win_h, &pst, true);
    hc::Model m16;
    if (id > 0 && p16_model(m, m16)) {
      hc::TileGeom g = make_geom(m, 1, prows);
      g.pitch = (g.cols + 1) & ~1;
      const size_t tile_bytes = ((size_t)g.pitch * g.rows * 2 + 7) & ~(size_t)7;   // one copy (cascade_body.inc P16_COPY)
      // vnf | copy B (the survivor list later) | bank matrix | 64 counters + ticket | copy A
      const size_t fixed = (size_t)prows * hc::TILE_POS * (4 + 2) + 264 + tile_bytes;
      if (tile_bytes >= (size_t)prows * hc::TILE_POS * 2 && node_kinds_fit_p16(m16)) {
        c->geom_p16 = g;
        c->tile_words_p16 = (int)(tile_bytes / 4);
        c->lds_bytes_p16 = fixed + tile_bytes;
        if ((rc = upload_p16_tables(c, m16, st))) return rc;
        c->spec_id_p16 = id;
        c->p16 = true;
      }
    }
  }
  // ---- a band one grid row short of the kernel's ROWS, where that row is what keeps one more workgroup off a CU (default's
  // scan-step-2 workgroup: 41,768 bytes of LDS, 808 over a fourth per CU; with 15-row bands 40,552).  The tile loses `ystep`
  // rows, nothing else moves (node offsets depend on the pitch only); the kernel stops at the band's own rows (CascadeArgs::band_step).
  for (int k = 0; k < 2; k++) {
    c->band_step[k] = c->k.band_rows[k];
    if (!c->k.band_trim || cv24 || (k == 0 && c->p16) || c->k.band_rows[k] < 16) continue;
    const auto per_cu = [&](size_t lds) { return std::min<size_t>(LDS_PER_CU / lds, 32 / (size_t)c->k.band_waves[k]); };
    hc::TileGeom g = c->geom[k];
    g.rows -= k + 1;
    const size_t words = ((size_t)g.pitch * g.rows + 1) & ~(size_t)1;
    const size_t lds = c->lds_bytes[k] - ((size_t)c->tile_words[k] - words) * 4;
    if (per_cu(lds) > per_cu(c->lds_bytes[k])) {
      c->geom[k] = g;
      c->tile_words[k] = (int)words;
      c->lds_bytes[k] = lds;
      c->band_step[k] = c->k.band_rows[k] - 1;
    }
  }
  return HC_OK;
}

}  // namespace hc

// =============================================================================== C ABI
extern "C" {

int hc_cascade_load(const char* path, hc_cascade** out) {
  if (!path || !out) return hc::set_err(HC_ERR_ARG, "null argument");
  *out = nullptr;
  hc_cascade* c = new hc_cascade;
  c->sync_cache_size = hc::sync_cache_size_from_env();
  std::string err;
  int rc = hc::load_model(path, c->model, err);
  if (rc != HC_OK) { delete c; return hc::set_err(rc, err); }
  *out = c;
  return HC_OK;
}

void hc_cascade_free(hc_cascade* c) {
  if (!c) return;
  for (hc_context* ctx : c->sync_ctxs) hc_context_destroy(ctx);
  delete c;
}

int hc_cascade_info(const hc_cascade* c, int* win_w, int* win_h, int* nstages, int* is_tree) {
  if (!c) return hc::set_err(HC_ERR_ARG, "null cascade");
  if (win_w) *win_w = c->model.win_w;
  if (win_h) *win_h = c->model.win_h;
  if (nstages) *nstages = (int)c->model.stages.size();
  if (is_tree) *is_tree = c->model.is_stump() ? 0 : 1;
  return HC_OK;
}

int hc_cascade_counts(const hc_cascade* c, int* ntrees, int* nnodes, int* nleaves) {
  if (!c) return hc::set_err(HC_ERR_ARG, "null cascade");
  if (ntrees) *ntrees = (int)c->model.trees.size();
  if (nnodes) *nnodes = (int)c->model.nodes.size();
  if (nleaves) *nleaves = (int)c->model.leaves.size();
  return HC_OK;
}

int hc_cascade_compiled_stages(const hc_cascade* c, uint64_t* hash, int* nstages) {
  if (!c) return hc::set_err(HC_ERR_ARG, "null cascade");
  const hc::Model& m = c->model;
  const unsigned long long h = hc::model_hash(m);
  int n = 0;
  hc::ensure_env_plugins();
  if (hc::weights_are_small_integers(m)) (void)hc::spec_lookup(h, m.win_w, m.win_h, &n);
  if (hash) *hash = (uint64_t)h;
  if (nstages) *nstages = n;
  return HC_OK;
}

int hc_cascade_dump(const hc_cascade* c, float* stage_thr, int32_t* stage_ntrees, int32_t* tree_nnodes,
                    int32_t* node_rects, float* node_thr, int32_t* node_left, int32_t* node_right, float* leaves) {
  if (!c) return hc::set_err(HC_ERR_ARG, "null cascade");
  const hc::Model& m = c->model;
  for (size_t i = 0; i < m.stages.size(); i++) {
    if (stage_thr) stage_thr[i] = m.stages[i].thr;
    if (stage_ntrees) stage_ntrees[i] = m.stages[i].ntrees;
  }
  if (tree_nnodes) for (size_t i = 0; i < m.trees.size(); i++) tree_nnodes[i] = m.trees[i].nnodes;
  for (size_t i = 0; i < m.nodes.size(); i++) {
    const hc::ModelNode& n = m.nodes[i];
    if (node_rects)
      for (int r = 0; r < 3; r++) {
        for (int j = 0; j < 4; j++) node_rects[i * 15 + r * 5 + j] = n.rect[r][j];
        node_rects[i * 15 + r * 5 + 4] = (int)n.weight[r];
      }
    if (node_thr) node_thr[i] = n.thr;
    if (node_left) node_left[i] = n.left;
    if (node_right) node_right[i] = n.right;
  }
  if (leaves) memcpy(leaves, m.leaves.data(), m.leaves.size() * sizeof(float));
  return HC_OK;
}

}  // extern "C"
