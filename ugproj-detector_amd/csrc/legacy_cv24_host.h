// legacy_cv24_host.h -- host-side planning of the OpenCV-2.4 ("T2") evaluation mode: scale loop,
// resize coefficient tables, hidden-cascade weights and the stage tree, shared by the context code (api_model.cpp, api_context.cpp) and the
// stand-alone entry points (legacy_cv24.cpp).
#pragma once

#include <vector>

#include "cascade_model.h"
#include "legacy_cv24.h"
#include "scale_plan.h"

namespace hc {

struct Level24 { double factor; int w, h, ystep, nx, ny, win_w, win_h; };

// B.1 / B.4: the scale loop of cvHaarDetectObjectsForROC (factor stays f64; the last row and column of valid window
// origins are not scanned)
std::vector<Level24> plan24(const Model& m, int rows, int cols, double sf, int min_w, int min_h, int max_w, int max_h);
// B.2: coefficient table of one axis of cv::resize(INTER_LINEAR), 8-bit (appended)
void linear_table24(int ssize, int dsize, std::vector<int>& ofs, std::vector<short>& a0, std::vector<short>& a1);

struct Cv24Host {
  std::vector<Level24> lv;
  std::vector<Cv24DevLevel> cl;          // same level order / offsets as Plan::levels
  std::vector<int> tab_ofs;
  std::vector<short> tab_a0, tab_a1;
  int max_pix = 0, max_pos = 0;
  double inv_area = 0;
  bool fused_tables = true;              // every table entry fits the packed form of the fused prep kernel (Plan::xtab / ytab)
};
// Plan (DevLevel records, integral / variance / band work lists, workspace sizes) + the 2.4-specific tables
int build_plan_cv24(const Model& m, int rows, int cols, double sf, int min_w, int min_h, int max_w, int max_h, int band_rows1,
                    int band_rows2, Plan& p, Cv24Host& h);

// B.6: hidden (area-normalised, rect-0-balanced) f32 weights of one node
void cv24_weights(const Model& m, const ModelNode& n, float w[3]);
// B.8: stage threshold as evaluated, (float)thr - 0.0001f
inline float cv24_stage_thr(const ModelStage& s) { return (float)s.thr_raw - 0.0001f; }
// every stage's parent is its predecessor and it has no sibling: the stage tree is the chain 0 -> 1 -> 2 ...
bool cv24_is_chain(const Model& m);
// tables of the window-at-a-time kernel (real stage trees); HC_ERR_UNSUPPORTED when the tree is not in forward order
int cv24_tree_tables(const Model& m, std::vector<Cv24Node>& nodes, std::vector<Cv24Tree>& trees, std::vector<Cv24Stage>& stages);

}  // namespace hc
