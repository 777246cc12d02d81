// cascade_model.h -- host-side Haar cascade model (product code, no OpenCV).
// Replaces the state cv::CascadeClassifier::load builds (reference src/file_io.cc:11)
// from the old-format files reference data/haarcascade_frontalface_*.xml.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace hc {

struct ModelNode {
  int8_t rect[3][4];   // up to 3 rects (x, y, w, h) inside the window; missing -> zeros
  float weight[3];     // raw XML weights (not area-normalised), missing -> 0
  float thr;           // node threshold, text -> f64 -> f32
  int32_t left, right; // <= 0: -(leaf index inside the tree); > 0: node index inside the tree
};

struct ModelTree { int32_t nnodes, first_node, first_leaf; };

struct ModelStage {
  double thr_raw;      // as parsed
  float thr;           // (float)thr_raw - 1e-5f, evaluated in f32
  int32_t ntrees, first_tree;
  int32_t parent, next;  // parsed, ignored by the OpenCV>=4 evaluation order (linear chain)
};

struct Model {
  int win_w = 0, win_h = 0;
  std::vector<ModelStage> stages;
  std::vector<ModelTree> trees;
  std::vector<ModelNode> nodes;
  std::vector<float> leaves;
  int max_nodes_per_tree = 0;
  bool has_tilted = false;
  bool is_stump() const { return max_nodes_per_tree == 1; }
};

// returns 0 (HC_OK) or a negative hc_status; err receives a human-readable reason
int load_model(const std::string& path, Model& m, std::string& err);

}  // namespace hc
