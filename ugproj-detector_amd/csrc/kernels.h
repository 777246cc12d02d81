// kernels.h -- launch interface of kernels.hip (host side sees only plain pointers).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <vector>

#include "device_types.h"

namespace hc {

struct CascadeArgs {
  const uint32_t* sum;        // per-frame integral arrays, frame stride int_stride words
  size_t int_stride;
  const float* vnf;           // per-frame grid arrays, frame stride grid_stride elements
  const double* vnf64;        // OpenCV-2.4 semantics (casc_cv24 build): the f64 variance factors, same layout
  size_t grid_stride;
  const DevLevel* levels;
  const WorkItem* bands;      // bands of ONE ystep class (all share geometry g)
  int nbands;
  int band_step;              // grid rows a band covers (<= the kernel's ROWS: hc_context::band_step); 0 = ROWS
  const DevNode* nodes;       // corner offsets resolved for g
  const DevNode* nodes_sparse; // the same records, exact stump stages in bank-spreading order: read by the sparse finish only
  const DevStage* stages;
  const DevStageKinds* kinds;  // per stage: stumps grouped by NodeKind (null for tree cascades)
  int nstages;
  const DevTree* trees;
  const float* leaves;
  TileGeom g;
  int row_stages;             // leading stages evaluated in row form (>= 1) before survivors are compacted
  int sparse_max;             // stages with <= sparse_max survivors in a tile run in sparse mode
  int tile_words;             // LDS words reserved for the integral tile
  unsigned long long* cand;   // raw candidate keys: frame<<40 | level<<32 | gy<<16 | gx
  unsigned int* cand_count;
  unsigned int cand_cap;
  int8_t* taps;               // nullable: per-position stage-pass counts
  const uint8_t* tree_pass;   // OpenCV-2.4 stage tree (k_cv24_tree): stage a window goes to when it passes / fails stage s;
  const uint8_t* tree_fail;   //   254 = accepted, 255 = rejected
  const uint8_t* tree_order;  // the stages in preorder of the stage tree (a stage, its child subtree, then its next sibling): both
                              //   targets of a stage come later in this order
  int int_weights;            // the node weight words hold int32 (w0 == -1 proven): run the exact-integer build
  int spec;                   // >= 1: id of the compiled early stages matching this cascade (spec_lookup), 0: none
  int frame_chunk;            // > 0: the grid is walked in chunks of this many frames (a multiple of 8), every band of a chunk
                              //   before the next chunk, so neighbouring bands of a frame stay close in time whatever the batch
  // OpenCV-2.4 stage tree as alternative chains (casc_cv24 build; 0 segments: a plain chain of nstages stages).  The stage
  // tables are laid out [spine, chain 1 | chain 2 | ..]; segment g covers device stages [seg_end[g - 1], seg_end[g]) and
  // segment 0 holds the spine (its first spine_len stages: a failure there rejects) and the first chain.
  int nseg, spine_len;
  int seg_end[4];
};

// Compiled early stages of one known cascade (csrc/spec/spec_<tag>.hip): what the generic launcher needs to know.
struct SpecUnit {
  unsigned long long hash;   // api_model.cpp model_hash of the cascade the unit was generated from
  int win_w, win_h, nstages; // window; number of leading stages compiled
  bool tree;                 // the cascade has two nodes per tree (k_cascade's TREE instantiation)
  // 0 / hipError_t, or -1 when the unit has no kernel for this launch shape (the generic kernel runs then)
  int (*launch)(hipStream_t st, int nframes, bool segments, int rows, int nwaves, size_t lds_bytes, const CascadeArgs& args);
  bool p16;                  // unit of the 16-bit-tile build (cascade_body.inc HC_P16): scan step 1, whole bands, its own node tables
  bool cv24 = false;         // unit of the OpenCV-2.4 build (cascade_body.inc HC_CV24): whole bands, the context's 2.4 tables
};
// id (>= 1) of the unit matching a model, 0 if none; *nstages (nullable) = stages it compiles
int spec_lookup(unsigned long long hash, int win_w, int win_h, int* nstages, bool p16 = false, bool cv24 = false);
const SpecUnit* spec_unit(int id);   // the unit of an id returned by spec_lookup (null: none)
// adds a unit that was not linked into the library (hc_spec_plugin_load); the object must outlive the process' use of it
int spec_register(const SpecUnit* u);
// layout check between the library and a plug-in built from another checkout of these sources
// (a plug-in carries the kernel TEXT of its checkout: besides the struct sizes, HC_KERNEL_LAYOUT names the version of everything
// the text assumes without a struct saying so -- LDS layout of k_cascade, node-word encodings, TileGeom meaning, the 16-bit
// tile's tables.  Bump it with any such change in cascade_body.inc / spec_unit.inc / api_model.cpp's table builders.)
constexpr int HC_KERNEL_LAYOUT = 8;
constexpr int SPEC_PLUGIN_ABI = (HC_KERNEL_LAYOUT * 1000 + (int)sizeof(CascadeArgs)) * 4096 +
                                (((int)sizeof(DevStage) * 31 + (int)sizeof(DevNode)) * 31 + (int)sizeof(TileGeom)) * 31 + (int)sizeof(DevTree) +
                                (int)sizeof(DevStageKinds) * 7 + (int)sizeof(DevLevel);

// k_pyramid_integral: the context's knobs that reach the launcher, and what a launch then did (hc_context_last_launch)
struct PrepKnobs { int frame_chunk = 32; bool var_ring = true, var_xring = true, var_split = true; int strip_comp_max = 0; };
struct PrepLaunchInfo {
  int frame_chunk = 0;     // chunk size the kernel was given
  int remapped = 0;        // 1: grid wider than the chunk, the workgroup remap ran
  int tail_frames = 0;     // frames in the short last chunk (0: none)
  int rows_per_chunk = 0;  // R
  int route = 0;           // hc_variance_route of the fused factors (0: none asked for)
  bool sq_stored = true;   // the squares' integral was written to global memory
};

// YCrCb skin bounds of cv::inRange (reference src/tracker/detector.cc:18, cfg.detection.skin_lower/upper)
struct SkinBounds { int32_t lo[3], hi[3]; };

// all launchers return 0 or a hipError_t value
int launch_resize(hipStream_t st, int nframes, int nitems, const uint8_t* frames, size_t frame_stride,
                  size_t row_step, int rows, int cols, const DevLevel* levels, const uint32_t* xtab,
                  const uint32_t* ytab, const WorkItem* items, uint8_t* pyr, size_t pyr_stride);
// fused resize + row scan + column scan: one workgroup per (level, frame); pyr may be null.  vnf non-null: the
// variance factor (A.6a) of every scanned window is produced as well (inner rect norm_w x norm_h, [gy][gx] per level
// at DevLevel::grid_ofs, frame stride grid_stride) and no k_variance launch is needed.
int launch_pyramid_integral(hipStream_t st, int nframes, int nlevels, int max_w, const uint8_t* frames,
                            size_t frame_stride, size_t row_step, int rows, int cols, const DevLevel* levels,
                            const uint32_t* xtab, const uint32_t* ytab, uint32_t* sum, uint32_t* sq, size_t int_stride,
                            uint8_t* pyr, size_t pyr_stride, const PrepKnobs& knobs, float* vnf = nullptr, size_t grid_stride = 0,
                            int norm_w = 0, int norm_h = 0, int max_nx = 0, bool store_sq = true, PrepLaunchInfo* info = nullptr,
                            int semantics = 0);
// ---- the same fused pass for levels of any width (k_pyramid_strips): a level is cut into column strips, one workgroup each
struct StripPlan {
  std::vector<DevStrip> strips;    // widest levels first, strips of a level left to right
  int lds_pitch = 0, src_pitch = 0;
  size_t lds_bytes = 0;
  size_t carry_words = 0;          // per frame
  int norm_w = 0, norm_h = 0;
  bool ok = false;                 // false: this geometry has no strip route (odd inner rect, table format, LDS)
};
// levels / xtab: the host copies of the plan (scale_plan.h); semantics 1: the 2.4 table packing
void plan_pyramid_strips(const std::vector<DevLevel>& levels, const std::vector<uint32_t>& xtab, int cols, int norm_w, int norm_h,
                         int semantics, const PrepKnobs& knobs, StripPlan& out);
// carry: nframes x plan.carry_words words; progress: nframes x strips words + 1 error word at the end (zeroed by the launcher)
int launch_pyramid_strips(hipStream_t st, int nframes, const StripPlan& plan, const DevStrip* d_strips, const uint8_t* frames,
                          size_t frame_stride, size_t row_step, int rows, int cols, const DevLevel* levels, const uint32_t* xtab,
                          const uint32_t* ytab, uint32_t* sum, uint32_t* sq, size_t int_stride, uint8_t* pyr, size_t pyr_stride,
                          const PrepKnobs& knobs, float* vnf, size_t grid_stride, bool store_sq, uint32_t* carry,
                          unsigned int* progress, PrepLaunchInfo* info, int semantics);
bool pyramid_integral_has_ring(const PrepKnobs& knobs, int max_w, int cols, int norm_w, int norm_h);
bool pyramid_integral_takes_cv24(const PrepKnobs& knobs, int max_w, int max_nx, int cols, int norm_w, int norm_h);
// FaceDetector::detectFaces pre-steps (face_kernels.hip): BGR -> mask + equalised, masked gray
int launch_face_prep(hipStream_t st, int nframes, const uint8_t* bgr, size_t frame_stride, size_t row_step, int rows,
                     int cols, const SkinBounds& sb, uint8_t* gray, uint8_t* mask, size_t plane_stride, uint32_t* hist,
                     uint8_t* lut);
// INTER_AREA decimation by integer factors (scan-size resize of the tracker)
int launch_area_decimate(hipStream_t st, int nframes, const uint8_t* src, size_t src_frame_stride, size_t src_row_step,
                         int channels, int fx, int fy, uint8_t* dst, size_t dst_frame_stride, int drows, int dcols);
// INTER_AREA, any shrinking ratio: per destination column/row the taps [ofs[d], ofs[d+1]) of (si, alpha)
int launch_area_resize(hipStream_t st, int nframes, const uint8_t* src, size_t src_frame_stride, size_t src_row_step,
                       int channels, const int* xofs, const int* xsi, const float* xal, const int* yofs, const int* ysi,
                       const float* yal, uint8_t* dst, size_t dst_frame_stride, int drows, int dcols);
// cvtColor(BGR2GRAY / BGRA2GRAY): channels = 3 or 4
int launch_bgr2gray(hipStream_t st, int nframes, const uint8_t* src, size_t frame_stride, size_t row_step, int channels,
                    int rows, int cols, uint8_t* gray, size_t plane_stride);
// boxes: nboxes x {frame, x, y, w, h}; sums[i] = sum of mask over box i
int launch_box_mask_sums(hipStream_t st, const uint32_t* integ, size_t int_stride, int pitch, const int* boxes, int nboxes,
                         uint32_t* sums);
int launch_integral_rows(hipStream_t st, int nframes, int nitems, const uint8_t* pyr, size_t pyr_stride,
                         const DevLevel* levels, const WorkItem* items, uint32_t* sum, uint32_t* sq,
                         size_t int_stride);
int launch_integral_cols(hipStream_t st, int nframes, int nitems, const DevLevel* levels, const WorkItem* items,
                         uint32_t* sum, uint32_t* sq, size_t int_stride);
int launch_publish(hipStream_t st, const unsigned int* count, const unsigned long long* cand, unsigned int cap,
                   unsigned long long* h_cand, unsigned int* h_pub, unsigned int seq, unsigned int* ticket,
                   const unsigned int* err = nullptr);
// K6: raw candidate keys -> per-frame bins -> grouped boxes and per-frame counts in page-locked host memory (0xFFFFFFFF: the
// frame overflowed a bin or its output slot and is left to the host); frame_count must be zero on entry
int launch_group(hipStream_t st, int nframes, const unsigned int* count, const unsigned long long* cand, unsigned int cap,
                 unsigned int* frame_count, unsigned long long* bins, const GroupLevel* levels, int semantics, int threshold,
                 GroupBox* h_boxes, unsigned int* h_nboxes, int bin_cap = HC_GROUP_BIN, int out_cap = HC_GROUP_OUT);
int launch_variance(hipStream_t st, int nframes, int nitems, const DevLevel* levels, const WorkItem* items,
                    const uint32_t* sum, const uint32_t* sq, size_t int_stride, int norm_w, int norm_h,
                    float* vnf, size_t grid_stride);
// rows = band height in grid rows (8, 16 or 32), nwaves = wavefronts per workgroup (4 or 8);
// lds_bytes = tile + vnf + 2 queues + counters
// *compiled (nullable): number of leading stages that ran as compiled code (SpecUnit), 0 for the generic kernel
int launch_cascade(hipStream_t st, int nframes, bool tree, bool segments, int rows, int nwaves, size_t lds_bytes,
                   const CascadeArgs& args, int* compiled = nullptr, int* variant = nullptr);

}  // namespace hc
