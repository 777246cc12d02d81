// cascade_model.cpp -- loader for the old "opencv-haar-classifier" XML (e.g. reference
// data/haarcascade_frontalface_alt.xml:46-48) and for the packed HCC1 container.
//
// The XML reader is a streaming tag scanner driven by element depth below the
// classifier node:
//   1 size | stages     2 <_> stage      3 trees | stage_threshold | parent | next
//   4 <_> tree          5 <_> node       6 feature | threshold | left_val | right_val |
//   left_node | right_node               7 rects | tilted        8 <_> rect "x y w h weight"
// Number handling follows OpenCV >= 4's in-memory conversion of old cascades
// (SURVEY.md A.2): text -> f64 (strtod) -> f32 for node thresholds, leaf values and
// rect weights; stage threshold f64 -> (float)t - 1e-5f in f32.

#include "cascade_model.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/hip_cascade.h"

namespace hc {
namespace {

struct Scanner {
  const char* p;
  const char* end;
  enum Kind { OPEN, CLOSE, TEXT, END, BAD };
  std::string name, type_id, text;

  static bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; }

  Kind next() {
    for (;;) {
      if (p >= end) return END;
      if (*p != '<') {
        const char* s = p;
        while (p < end && *p != '<') ++p;
        // skip whitespace-only runs
        const char* a = s;
        while (a < p && is_space(*a)) ++a;
        if (a == p) continue;
        const char* b = p;
        while (b > a && is_space(b[-1])) --b;
        text.assign(a, b);
        return TEXT;
      }
      if (end - p >= 4 && !memcmp(p, "<!--", 4)) {
        const char* q = (const char*)memmem(p + 4, end - p - 4, "-->", 3);
        if (!q) return BAD;
        p = q + 3;
        continue;
      }
      if (end - p >= 2 && p[1] == '?') {
        const char* q = (const char*)memmem(p, end - p, "?>", 2);
        if (!q) return BAD;
        p = q + 2;
        continue;
      }
      const char* q = (const char*)memchr(p, '>', end - p);
      if (!q) return BAD;
      if (p[1] == '/') {
        name.assign(p + 2, q);
        p = q + 1;
        return CLOSE;
      }
      const char* s = p + 1;
      const char* n = s;
      while (n < q && !is_space(*n) && *n != '/') ++n;
      name.assign(s, n);
      type_id.clear();
      const char* t = (const char*)memmem(n, q - n, "type_id=", 8);
      if (t) {
        t += 8;
        char quote = *t++;
        const char* u = t;
        while (u < q && *u != quote) ++u;
        type_id.assign(t, u);
      }
      p = q + 1;
      return OPEN;   // self-closing tags do not occur in this format
    }
  }
};

int fail(std::string& err, const char* what) { err = what; return HC_ERR_FORMAT; }

int parse_xml(const std::string& buf, Model& m, std::string& err) {
  Scanner sc{buf.data(), buf.data() + buf.size(), {}, {}, {}};
  int depth = -1;            // depth below the classifier element; -1 = not inside yet
  ModelStage st{};
  ModelTree tr{};
  ModelNode nd{};
  int nrect = 0, nleaf = 0;
  std::vector<std::string> path;   // names at depth 1..8
  bool seen_size = false;
  for (;;) {
    Scanner::Kind k = sc.next();
    if (k == Scanner::BAD) return fail(err, "malformed XML");
    if (k == Scanner::END) break;
    if (depth < 0) {
      if (k == Scanner::OPEN) {
        if (sc.type_id == "opencv-haar-classifier") { depth = 0; path.assign(1, sc.name); }
      }
      continue;
    }
    if (k == Scanner::OPEN) {
      ++depth;
      path.resize(depth + 1);
      path[depth] = sc.name;
      if (depth == 2 && path[1] == "stages") { st = ModelStage{}; st.parent = st.next = -1; st.first_tree = (int)m.trees.size(); }
      else if (depth == 4 && path[3] == "trees") { tr = ModelTree{}; tr.first_node = (int)m.nodes.size(); tr.first_leaf = (int)m.leaves.size(); nleaf = 0; }
      else if (depth == 5 && path[3] == "trees") { nd = ModelNode{}; nrect = 0; nd.left = nd.right = INT32_MIN; }
    } else if (k == Scanner::TEXT) {
      const std::string& t = sc.text;
      const std::string& tag = path[depth];
      if (depth == 1 && tag == "size") {
        char* e;
        m.win_w = (int)strtol(t.c_str(), &e, 10);
        m.win_h = (int)strtol(e, &e, 10);
        seen_size = true;
      } else if (depth == 3 && tag == "stage_threshold") st.thr_raw = strtod(t.c_str(), nullptr);
      else if (depth == 3 && tag == "parent") st.parent = (int)strtol(t.c_str(), nullptr, 10);
      else if (depth == 3 && tag == "next") st.next = (int)strtol(t.c_str(), nullptr, 10);
      else if (depth == 8 && path[7] == "rects") {
        if (nrect >= 3) return fail(err, "feature with more than 3 rects");
        const char* s = t.c_str();
        char* e;
        for (int j = 0; j < 4; j++) { long v = strtol(s, &e, 10); if (v < -128 || v > 127) return fail(err, "rect out of range"); nd.rect[nrect][j] = (int8_t)v; s = e; }
        nd.weight[nrect] = (float)strtod(s, &e);
        ++nrect;
      } else if (depth == 7 && tag == "tilted") { if (strtol(t.c_str(), nullptr, 10) != 0) m.has_tilted = true; }
      else if (depth == 6 && tag == "threshold") nd.thr = (float)strtod(t.c_str(), nullptr);
      else if (depth == 6 && tag == "left_val") { nd.left = -nleaf; m.leaves.push_back((float)strtod(t.c_str(), nullptr)); ++nleaf; }
      else if (depth == 6 && tag == "right_val") { nd.right = -nleaf; m.leaves.push_back((float)strtod(t.c_str(), nullptr)); ++nleaf; }
      else if (depth == 6 && tag == "left_node") nd.left = (int)strtol(t.c_str(), nullptr, 10);
      else if (depth == 6 && tag == "right_node") nd.right = (int)strtol(t.c_str(), nullptr, 10);
    } else {  // CLOSE
      if (depth == 5 && path[3] == "trees") {
        if (nd.left == INT32_MIN || nd.right == INT32_MIN || nrect < 1) return fail(err, "incomplete tree node");
        m.nodes.push_back(nd);
        ++tr.nnodes;
      } else if (depth == 4 && path[3] == "trees") {
        if (tr.nnodes < 1 || nleaf != tr.nnodes + 1) return fail(err, "tree is not a full binary tree");
        if (tr.nnodes > m.max_nodes_per_tree) m.max_nodes_per_tree = tr.nnodes;
        m.trees.push_back(tr);
        ++st.ntrees;
      } else if (depth == 2 && path[1] == "stages") {
        if (st.ntrees < 1) return fail(err, "empty stage");
        m.stages.push_back(st);
      }
      if (depth == 0) { depth = -1; break; }   // only the first classifier node is used
      --depth;
    }
  }
  if (!seen_size || m.stages.empty() || m.win_w <= 2 || m.win_h <= 2) return fail(err, "no opencv-haar-classifier content");
  return HC_OK;
}

int parse_hcc(const std::string& b, Model& m, std::string& err) {
  size_t p = 4;
  auto need = [&](size_t n) { return p + n <= b.size(); };
  auto i32 = [&]() { int32_t v; memcpy(&v, b.data() + p, 4); p += 4; return v; };
  auto f32 = [&]() { float v; memcpy(&v, b.data() + p, 4); p += 4; return v; };
  auto f64 = [&]() { double v; memcpy(&v, b.data() + p, 8); p += 8; return v; };
  if (!need(28)) return fail(err, "truncated HCC1 header");
  m.win_w = i32(); m.win_h = i32();
  int nst = i32(), ntr = i32(), nnd = i32(), nlf = i32(), flags = i32();
  m.has_tilted = (flags & 1) != 0;
  if (nst < 1 || ntr < nst || nnd < ntr || nlf != nnd + ntr) return fail(err, "inconsistent HCC1 counts");
  if (!need((size_t)nst * 20 + (size_t)ntr * 4 + (size_t)nnd * 40 + (size_t)nlf * 4)) return fail(err, "truncated HCC1 body");
  int acc = 0;
  for (int i = 0; i < nst; i++) {
    ModelStage s{};
    s.thr_raw = f64(); s.ntrees = i32(); s.parent = i32(); s.next = i32();
    s.first_tree = acc; acc += s.ntrees;
    m.stages.push_back(s);
  }
  if (acc != ntr) return fail(err, "stage/tree count mismatch");
  int fn = 0, fl = 0;
  for (int i = 0; i < ntr; i++) {
    ModelTree t{}; t.nnodes = i32(); t.first_node = fn; t.first_leaf = fl;
    if (t.nnodes < 1) return fail(err, "empty tree");
    fn += t.nnodes; fl += t.nnodes + 1;
    if (t.nnodes > m.max_nodes_per_tree) m.max_nodes_per_tree = t.nnodes;
    m.trees.push_back(t);
  }
  if (fn != nnd) return fail(err, "tree/node count mismatch");
  for (int i = 0; i < nnd; i++) {
    ModelNode n{};
    for (int k = 0; k < 3; k++) for (int j = 0; j < 4; j++) n.rect[k][j] = (int8_t)b[p++];
    for (int k = 0; k < 3; k++) n.weight[k] = f32();
    n.thr = f32(); n.left = i32(); n.right = i32();
    if (i32() != 0) m.has_tilted = true;
    m.nodes.push_back(n);
  }
  for (int i = 0; i < nlf; i++) m.leaves.push_back(f32());
  return HC_OK;
}

// structural checks the device code relies on
int validate(Model& m, std::string& err) {
  static const float kThresholdEps = 1e-5f;
  for (auto& s : m.stages) s.thr = (float)s.thr_raw - kThresholdEps;
  if (m.has_tilted) { err = "tilted Haar features are not supported"; return HC_ERR_UNSUPPORTED; }
  // both parsers end here: everything the host planner and the kernels index without further checks
  if (m.win_w <= 2 || m.win_h <= 2 || m.win_w > 127 || m.win_h > 127) { err = "window size out of range"; return HC_ERR_FORMAT; }
  if (m.stages.empty() || m.trees.empty()) { err = "empty cascade"; return HC_ERR_FORMAT; }
  {
    size_t next_tree = 0;
    for (const auto& s : m.stages) {
      if (s.ntrees < 1 || s.first_tree < 0 || (size_t)s.first_tree != next_tree ||
          (size_t)s.first_tree + (size_t)s.ntrees > m.trees.size()) {
        err = "stage with no trees / tree range outside the model";
        return HC_ERR_FORMAT;
      }
      next_tree += (size_t)s.ntrees;
    }
    if (next_tree != m.trees.size()) { err = "trees not covered by the stages"; return HC_ERR_FORMAT; }
    size_t next_node = 0, next_leaf = 0;
    for (const auto& t : m.trees) {
      if (t.nnodes < 1 || (size_t)t.first_node != next_node || (size_t)t.first_leaf != next_leaf) {
        err = "tree node / leaf ranges inconsistent";
        return HC_ERR_FORMAT;
      }
      next_node += (size_t)t.nnodes;
      next_leaf += (size_t)t.nnodes + 1;
    }
    if (next_node != m.nodes.size() || next_leaf != m.leaves.size()) { err = "node / leaf counts inconsistent"; return HC_ERR_FORMAT; }
  }
  for (const auto& t : m.trees) {
    for (int k = 0; k < t.nnodes; k++) {
      ModelNode& n = m.nodes[t.first_node + k];
      for (int side = 0; side < 2; side++) {
        int c = side ? n.right : n.left;
        // children must come after their parent (single forward pass on the device) and leaf
        // indices must address the tree's own nnodes+1 leaves
        if (c > 0 ? (c <= k || c >= t.nnodes) : (-c > t.nnodes)) { err = "malformed tree topology"; return HC_ERR_UNSUPPORTED; }
      }
      for (int r = 0; r < 3; r++) {
        if (n.weight[r] == 0.f) { n.rect[r][0] = n.rect[r][1] = n.rect[r][2] = n.rect[r][3] = 0; continue; }   // contributes 0 * S(r); the device tables zero its offsets
        if (n.rect[r][0] < 0 || n.rect[r][1] < 0 || n.rect[r][2] < 1 || n.rect[r][3] < 1 ||
            n.rect[r][0] + n.rect[r][2] > m.win_w || n.rect[r][1] + n.rect[r][3] > m.win_h) {
          err = "feature rect outside the window"; return HC_ERR_UNSUPPORTED;
        }
      }
    }
  }
  return HC_OK;
}

}  // namespace

int load_model(const std::string& path, Model& m, std::string& err) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { err = "cannot open " + path; return HC_ERR_IO; }
  std::string buf;
  char tmp[1 << 16];
  size_t n;
  while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, n);
  fclose(f);
  m = Model{};
  int rc = (buf.size() >= 4 && !memcmp(buf.data(), "HCC1", 4)) ? parse_hcc(buf, m, err) : parse_xml(buf, m, err);
  if (rc != HC_OK) return rc;
  return validate(m, err);
}

}  // namespace hc
