// scale_plan.h -- per-(frame size, call parameters) plan of the detectMultiScale call
// at reference src/tracker/detector.cc:25-31: scale ladder, pyramid level geometry,
// exact-bilinear coefficient tables, workspace layout and launch work lists.
#pragma once

#include <cstdint>
#include <vector>

#include "cascade_model.h"
#include "device_types.h"

namespace hc {

struct Plan {
  int rows = 0, cols = 0;
  double scale_factor = 1.1;
  int min_w = 0, min_h = 0, max_w = 0, max_h = 0;
  int nstripes = 0;

  std::vector<DevLevel> levels;       // device-visible per-level record
  std::vector<uint8_t> scanned;       // per level: in this cascade's own ladder (always 1 unless the prep pass is shared)
  std::vector<uint32_t> xtab, ytab;   // resize tables, all levels concatenated: (ofs << 9) | c1
  std::vector<WorkItem> resize_items; // 64x4-pixel tiles
  std::vector<WorkItem> row_items;    // 4 rows per block
  std::vector<WorkItem> col_items;    // 256 columns per block
  std::vector<WorkItem> var_items;    // 256 grid positions of one grid row segment per block
  std::vector<WorkItem> band_items[2];  // cascade kernel: bands of band_rows[c] grid rows; c = 0: ystep 1, 1: ystep 2
  int band_rows[2] = {16, 16};

  int64_t pyr_bytes = 0;    // per frame
  int64_t int_words = 0;    // per frame, per integral array
  int64_t grid_elems = 0;   // per frame: sum nx * ny_all  (ny_all = ceil(szw_h / ystep))
  int64_t algorithmic_bytes = 0;  // W*H + sum 16*(w+1)*(h+1)
};

// Fills `p`. Returns HC_OK; an image smaller than the window or an empty ladder yields a
// plan with zero levels (detectMultiScale then returns no rectangles).
// prep_scales (optional): the levels to lay out -- a superset of the cascade's own ladder, ascending -- when
// several cascades share one pyramid + integral pass (hc_context_configure_shared).
int build_plan(const Model& m, int rows, int cols, double scale_factor, int min_w, int min_h,
               int max_w, int max_h, int band_rows_ystep1, int band_rows_ystep2, Plan& p,
               const std::vector<float>* prep_scales = nullptr);

// the f32 scale ladder of the call; empty when the image is smaller than the window
std::vector<float> scale_ladder(const Model& m, int rows, int cols, double scale_factor, int min_w, int min_h, int max_w,
                                int max_h);

// (ofs << 9) | c1 for every destination index of a src -> dst exact bilinear resize
void build_resize_table(int src, int dst, std::vector<uint32_t>& out);

// cv::resize(INTER_AREA) tap table of one axis for a shrinking, non-integer ratio (computeResizeAreaTab):
// destination index d covers taps [ofs[d], ofs[d+1]) of (si[k], alpha[k]); f64 arithmetic, alpha rounded to f32.
void build_area_table(int ssize, int dsize, std::vector<int>& ofs, std::vector<int>& si, std::vector<float>& alpha);

}  // namespace hc
