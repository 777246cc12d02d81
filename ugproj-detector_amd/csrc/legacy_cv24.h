// legacy_cv24.h -- device structs and launchers of the OpenCV-2.4 ("T2") evaluation mode.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace hc {

struct Cv24DevLevel {
  int32_t w, h, ystep, nx, ny;   // level image, scan step, scanned grid (x < w - win_w, y < h - win_h)
  int32_t int_pitch;             // integral row pitch (words)
  int32_t resize_mode;           // 0 copy, 1 bilinear tables, 2 exact half (area kernel)
  int32_t xtab, ytab;            // offsets into the coefficient tables
  int64_t pyr_ofs, int_ofs, grid_ofs;
};

struct Cv24Node {
  int8_t rect[3][4];             // x, y, w, h in window coordinates
  float w[3];                    // hidden (area-normalised, zero-balanced) rect weights, rule B.6
  float thr;
  int32_t left, right;           // > 0: node index inside the tree; <= 0: -(leaf index inside the tree)
  int32_t has3;
};
struct Cv24Tree { int32_t first_node, first_leaf; };
struct Cv24Stage { int32_t first_tree, ntrees; float thr; int32_t parent, next, child; };

struct Cv24Args {
  const Cv24DevLevel* levels;
  const uint32_t* sum;           // frame f at + f * int_stride
  const uint32_t* sq;
  size_t int_stride, grid_stride;
  const Cv24Stage* stages;
  const Cv24Tree* trees;
  const Cv24Node* nodes;
  const float* leaves;
  double inv_area;
  int win_w, win_h;
  unsigned long long* cand;      // frame << 40 | level << 32 | gy << 16 | gx
  unsigned int* cand_count;
  unsigned int cand_cap;
  int8_t* taps;                  // nullable
};

// frame f of the batch: src + f * frame_stride -> pyr + f * pyr_stride
int launch_cv24_resize(hipStream_t st, int nframes, int nlevels, int max_pixels, const uint8_t* src, size_t frame_stride, size_t row_step,
                       int rows, int cols, const Cv24DevLevel* levels, const int* tab_ofs, const short* tab_a0, const short* tab_a1,
                       uint8_t* pyr, size_t pyr_stride);
// window-at-a-time evaluation with the cascade file's stage tree (cascades whose stages do not form a linear chain)
int launch_cv24_cascade(hipStream_t st, int nframes, int nlevels, int max_positions, const Cv24Args& args);

}  // namespace hc
