// hc_context.h -- the context object behind the C ABI and what its translation units share (never installed).
//
// The host side of the library is cut along its seams, one file each:
//   api_core.cpp      error text, allocation counting, worker pool, page-locked host memory
//   api_model.cpp     cascade-level entry points, node / stage tables of a context (upload_model)
//   api_plugins.cpp   compiled-stage plug-ins (hc_spec_plugin_load, HC_SPEC_PLUGINS)
//   api_context.cpp   Knobs, context create / destroy / configure / configure_shared / set_semantics, route policy
//   api_enqueue.cpp   the enqueue routes (one function per prep route, one for the cascade launches)
//   api_wait.cpp      result wait, candidate decode, grouping
//   api_faces.cpp     FaceDetector::detectFaces pre / post steps
//   api_sync.cpp      the synchronous drop-in calls and their context cache
//   api_inspect.cpp   plan queries, debug taps, timing, launch report
//   api_boxes.cpp     box files, hc_group_rectangles
#pragma once

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hip_cascade.h"
#include "cascade_model.h"
#include "hc_internal.h"
#include "kernels.h"
#include "legacy_cv24.h"
#include "legacy_cv24_host.h"
#include "scale_plan.h"

namespace hc {

// Every environment variable that changes what a context does, read ONCE when the context is created
// (hc_context_create) and carried by the context from then on: a test that sets a variable and then creates a
// context gets that value, whatever another context of the process read earlier.  Nothing below the ABI calls
// getenv for a product decision (HC_SPEC_PLUGINS is a process-wide registry and stays a once-per-process read).
struct Knobs {
  // ---- k_cascade tiling and modes
  int band_rows[2] = {16, 16};   // HC_BAND_ROWS1/2: grid rows per band, per scan-step class (index 0: step 1, 1: step 2)
  bool band_trim = true;         // HC_BAND_TRIM: a band may cover one grid row less than the kernel's ROWS when that fits one more workgroup per CU
  int band_waves[2] = {4, 8};    // HC_BAND_WAVES1/2: wavefronts per cascade workgroup (4 or 8); measured best on MI355X
  int sparse_max = 32;           // HC_SPARSE_MAX: a tile's last <= this many windows take the finish
  int row_stages = 2;            // HC_ROW_STAGES: leading stages in row form (conflict-free LDS reads while 88 % of the windows live)
  int frame_chunk = 128;         // HC_FRAME_CHUNK: k_cascade walks a large batch in chunks of this many frames (0: one sweep)
  bool allow_int_weights = true; // HC_INT_WEIGHTS=0: always the f32 feature arithmetic (A/B, debugging)
  bool bank_order = true;        // HC_BANK_ORDER=0: the finish reads the stumps in the dense path's order
  bool allow_spec = true;        // HC_SPEC=0: never the compiled early stages (csrc/spec), always the generic kernel
  bool pair_form = true;         // HC_PAIR_FORM=0: tree stages always walk their trees per lane (A/B, debugging)
  bool allow_p16 = true;         // HC_P16=0: never the 16-bit-tile build
  int debug_max_stages = 0;      // HC_DEBUG_MAX_STAGES: truncate the cascade (wrong results; kernel-time experiments only)
  int debug_lds_pad[2] = {0, 0}; // HC_DEBUG_LDS_PAD1/2: extra LDS bytes per cascade workgroup (occupancy experiments only)
  // ---- prep route
  int fused_prep = -1;           // HC_FUSED_PREP: -1 by batch size (cost model), 0 never fused, 1 always fused
  double fused_min_batch = -1;   // HC_FUSED_MIN_BATCH: a plain threshold in 720p-frame equivalents instead of the cost model
  bool fuse_variance = true;     // HC_FUSE_VARIANCE=0: always the separate k_variance kernel
  bool fuse_variance_forced = false;   // HC_FUSE_VARIANCE=1: inside the fused prep kernel whatever route that takes
  int pi_frame_chunk = 32;       // HC_PI_FRAME_CHUNK: k_pyramid_integral walks the batch in chunks of this many frames (0: level-major)
  bool var_ring = true;          // HC_VAR_RING=0: variance factors by read-back instead of the register rings
  bool var_xring = true;         // HC_VAR_XRING=0: grid columns beyond the register rings read back instead of LDS rings
  bool var_split = true;         // HC_VAR_SPLIT=0: never the column-strip form of the fused prep kernel (k_pyramid_strips): the
                                 // one-workgroup-per-level kernel with its own ring / read-back / k_variance routes (round 3)
  int strip_comp_max = 0;        // HC_STRIP_COMP_MAX: pixel columns a strip computes at most; 0: the widest that fits (tests: small values
                                 // cut small frames into several strips)
  bool cv24_fused = true;        // HC_CV24_FUSED=0: 2.4 semantics always through the four separate prep kernels
  bool cv24_seg = true;          // HC_CV24_SEG=0: a 2.4 stage tree always through the preorder sweep (k_cv24_tree), never as alternative chains
  bool share_vnf = true;         // HC_SHARE_VNF=0: every member of a shared pass computes its own variance factors
  // ---- small-batch routes, streams, host
  int side_max_batch = 8;        // HC_SIDE_MAX_BATCH: up to this many 720p equivalents the two cascade launches run side by side
  int seg_max_batch = 8;         // HC_SEG_MAX_BATCH: ... and bands are walked in segments
  int seg_tiles = 1;             // HC_SEG_TILES: tiles per segment
  bool side_prio = true;         // HC_SIDE_PRIO=0: the side stream at normal priority
  bool copy_prio = true;         // HC_COPY_PRIO=0: the copy stream at normal priority
  int host_threads = 4;          // HC_HOST_THREADS: post-processing threads per context
  bool device_group = true;      // HC_DEVICE_GROUP=0: rectangles are grouped on the host (hc_context_set_device_grouping)
  int group_min_batch = 8;       // HC_GROUP_MIN_BATCH: batches of fewer frames are grouped on the host (one frame: two more launches and a
                                 //   one-workgroup sort cost 0.075 ms of latency, the host groups its ~200 rectangles in 0.01; even at 8 frames)
  int group_bin_cap = HC_GROUP_BIN;   // HC_GROUP_BIN_CAP / HC_GROUP_OUT_CAP (tests): raw candidates / grouped boxes per frame beyond
  int group_out_cap = HC_GROUP_OUT;   //   which a frame is left to the host, at most the compiled sizes
};
Knobs knobs_from_env();
// HC_SYNC_CONTEXTS, read when a cascade is loaded: how many contexts the synchronous calls keep per cascade
int sync_cache_size_from_env();

}  // namespace hc

// Small persistent worker pool for the per-frame host post-processing (candidate decode + rectangle grouping): frames
// are independent, so a batch is a parallel-for over frames.
class WorkerPool {
 public:
  explicit WorkerPool(int nthreads);
  ~WorkerPool();
  // calls fn(i) for i in [0, n), on the pool plus the calling thread; returns when all are done
  void parallel_for(int n, const std::function<void(int)>& fn);

 private:
  void drain();
  void loop();
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> next_{0};
  int n_ = 0, active_ = 0;
  unsigned long long gen_ = 0;
  bool stop_ = false;
};

struct hc_context {
  const hc_cascade* casc = nullptr;
  int device = 0;
  hc::Knobs k;                         // parsed once in hc_context_create
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t copy_stream = nullptr;   // D2H of results: never queues behind a later batch on `stream`
  hipEvent_t ev_done = nullptr;        // recorded on `stream` after the last kernel of a batch
  // Recorded on copy_stream behind every device-to-host copy the host then needs.  The host waits for THIS EVENT, never
  // for the stream: the runtime multiplexes streams onto a few hardware queues (four by default), a stream
  // synchronisation places a marker at the tail of the hardware queue, and with four or more contexts alive the queue of
  // one context's copy stream is also the queue of another's -- whose copy waits for a batch that is still running.
  // Measured on a two-cascade shared pass (two pipeline slots = four contexts): every wait returned one whole batch
  // late, the software pipeline ran one deep and the GPU idled 4 ms of every 41 ms step.
  hipEvent_t ev_copied = nullptr;      // (now only the skin filter's box sums; a batch's results are published, see h_pub)
  // {sequence number, raw-candidate count} in coherent page-locked host memory, written by the batch's last kernel
  // (k_publish); the candidates themselves arrive in h_cand the same way.  wait_core polls the sequence word.
  unsigned int* h_pub = nullptr;
  unsigned int* d_ticket = nullptr;    // k_publish: workgroups that have copied their share of the candidates
  unsigned int pub_seq = 0;
  hipStream_t side_stream = nullptr;   // small batches: the ystep-2 cascade launch runs beside the ystep-1 launch
  // Large host batches go up on a stream of their own (made on first use) and `stream` waits for ev_uploaded: contexts
  // that share one stream (kernels of consecutive batches back to back, the fastest arrangement) still get batch i+1's
  // frames across PCIe while batch i computes.
  hipStream_t upload_stream = nullptr;
  hipEvent_t ev_uploaded = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;

  hc::Plan plan;
  bool configured = false;
  // Which generation of the call the context evaluates (hc_context_set_semantics): 0 = OpenCV >= 4,
  // 1 = OpenCV 2.4.x (Appendix B): its own scale loop, cvResize pyramid, f64 variance factors and node arithmetic; the
  // LDS-tiled band kernel when the cascade's stage tree is a chain, the window-at-a-time kernel otherwise (alt_tree).
  int semantics = 0;
  hc::Cv24Host cv;
  bool cv_chain = true;
  double* d_vnf64 = nullptr;
  hc::Cv24DevLevel* d_cvlevels = nullptr;
  int* d_cvtab_ofs = nullptr;
  short *d_cvtab_a0 = nullptr, *d_cvtab_a1 = nullptr;
  hc::Cv24Node* d_cvnodes = nullptr;
  hc::Cv24Tree* d_cvtrees = nullptr;
  hc::Cv24Stage* d_cvstages = nullptr;
  uint8_t *d_tree_pass = nullptr, *d_tree_fail = nullptr, *d_tree_order = nullptr;   // stage tree on the tiles (k_cv24_tree)
  bool cv_tree_tiled = false;
  // stage tree as alternative chains on the chain kernel (api_model.cpp, cascade_body.inc): stage tables in evaluation order
  bool cv_seg = false;
  int seg_n = 0, seg_spine = 0, seg_end[4] = {0, 0, 0, 0};
  hc::DevStage* d_stages_seg = nullptr;
  hc::DevStageKinds* d_kinds_seg = nullptr;
  // several cascades over the same frames (hc_context_configure_shared): the leader owns the pyramid integrals of
  // the union ladder, followers read them.  prep_seq counts the leader's enqueues; a follower scans each once.
  hc_context* leader = nullptr;
  std::vector<hc_context*> followers;
  unsigned long long prep_seq = 0, seen_seq = 0;
  int prep_frames = 0;
  hipEvent_t ev_prep = nullptr;        // leader: recorded after the prep kernels of a batch
  hipEvent_t ev_var = nullptr;         // leader: recorded after its variance factors (followers with the leader's window read them)
  bool vnf_shared = false;             // follower whose window and grid equal the leader's: d_vnf IS the leader's, no k_variance of its own
  bool vnf_from_leader = false;        // follower with another window whose factors the leader's fused prep kernel emits (second ring height)
  bool scan_in_flight = false;         // follower: ev_done of its last scan has been recorded and not yet waited by the leader
  int max_batch = 0, cand_per_frame = 0, min_neighbors = 0;
  size_t cand_cap = 0;                   // raw-candidate capacity of d_cand / h_cand for a whole batch (grows on demand)
  // cascade-kernel tiling per ystep class (index 0: ystep 1, index 1: ystep 2)
  hc::TileGeom geom[2] = {};
  size_t lds_bytes[2] = {0, 0};
  int tile_words[2] = {0, 0};

  // device.  d_frames, d_color, d_pyr and d_taps are allocated on first use: the device-resident
  // throughput path needs none of them (two 3840x2160 contexts would not fit otherwise)
  uint8_t* d_frames = nullptr;   // staging for the host-input path
  uint8_t* d_color = nullptr;    // staging for multi-channel input to detectMultiScale (BGR / BGRA)
  size_t color_bytes = 0;
  uint8_t* d_pyr = nullptr;      // pyramid bytes: the unfused prep kernels and the debug tap
  uint32_t* d_sum = nullptr;
  uint32_t* d_sq = nullptr;
  float* d_vnf = nullptr;
  int8_t* d_taps = nullptr;
  hc::DevLevel* d_levels = nullptr;
  uint32_t* d_xtab = nullptr;
  uint32_t* d_ytab = nullptr;
  // column strips of the fused prep kernel (k_pyramid_strips): geometries the one-workgroup-per-level kernel has no ring route
  // for (levels wider than 1024 columns, frames whose source rows do not fit its staging).  use_strips: decided at configure time
  hc::StripPlan strips;
  bool use_strips = false;
  bool strips_ran = false;                 // the batch in flight went through k_pyramid_strips
  hc::DevStrip* d_strips = nullptr;
  uint32_t* d_carry = nullptr;             // per frame: the finished last-column integrals a strip hands to its right neighbour
  unsigned int* d_strip_progress = nullptr;   // per (frame, strip): rows published; + one error word
  hc::WorkItem *d_resize_items = nullptr, *d_row_items = nullptr, *d_col_items = nullptr, *d_var_items = nullptr,
               *d_band_items[2] = {nullptr, nullptr}, *d_seg_items[2] = {nullptr, nullptr};
  int n_seg_items[2] = {0, 0};           // bands cut into segments of seg_tiles tiles: the work list of small batches
  hc::DevNode* d_nodes[2] = {nullptr, nullptr};   // corner offsets resolved for geom[c]
  hc::DevNode* d_nodes_sparse[2] = {nullptr, nullptr};   // the same records, every exact stage in bank-spreading order (the finish)
  bool int_weights = false;              // every node: w0 == -1, w1 / w2 integers, weighted rect sums below 2^24 (upload_model)
  int spec_id = 0;                       // compiled early stages matching this model (hc::spec_lookup), 0: none
  // 16-bit-tile build of the scan-step-1 launch (cascade_body.inc HC_P16): its own tile geometry, node tables (2-byte
  // entry offsets, rects of more than 257 pixels split in two) and compiled unit; whole-band launches only
  bool p16 = false;
  int spec_id_p16 = 0;
  hc::TileGeom geom_p16 = {};
  size_t lds_bytes_p16 = 0;
  int tile_words_p16 = 0;
  hc::DevNode *d_nodes_p16 = nullptr, *d_nodes_sparse_p16 = nullptr;
  hc::DevStageKinds* d_kinds_p16 = nullptr;
  int last_compiled = 0;                 // stages that ran as compiled code in EVERY cascade launch of the last enqueued batch
  int band_step[2] = {16, 16};           // grid rows per band of each scan-step class (upload_model: k.band_rows or one less)
  hc_launch_info last_launch = {};       // what the last enqueue launched (hc_context_last_launch)
  hc::DevStage* d_stages = nullptr;
  hc::DevStageKinds* d_kinds = nullptr;
  hc::DevTree* d_trees = nullptr;
  float* d_leaves = nullptr;
  unsigned long long* d_cand = nullptr;
  unsigned int* d_count = nullptr;
  // pinned host
  unsigned long long* h_cand = nullptr;

  // FaceDetector::detectFaces path (reference src/tracker/detector.cc:10-43)
  bool faces_enabled = false, faces_pending = false;
  uint8_t *d_bgr = nullptr, *d_gray = nullptr, *d_mask = nullptr, *d_lut = nullptr;
  uint32_t *d_hist = nullptr, *d_msum = nullptr, *d_boxsums = nullptr;
  int* d_boxes = nullptr;
  hc::DevLevel* d_idlevel = nullptr;      // identity "level" (w = cols, h = rows) for the mask integral
  uint32_t *d_idx = nullptr, *d_idy = nullptr;
  hc::WorkItem* d_mask_items[2] = {nullptr, nullptr};   // row / column scan work lists of the mask plane (small batches)
  int n_mask_items[2] = {0, 0};
  int* h_boxes = nullptr;                 // pinned: {frame, x, y, w, h} per grouped rect
  uint32_t* h_boxsums = nullptr;
  int box_cap = 0, mask_pitch = 0;
  // scan-size resize (reference src/tracker/face_tracker.cc:112-114): source frames are src_rows x src_cols,
  // decimated by INTER_AREA to the configured rows x cols before detectFaces
  int src_rows = 0, src_cols = 0;
  uint8_t* d_src_bgr = nullptr;
  bool area_general = false;              // non-integer ratio: tap tables instead of the box kernel
  int *d_area_xofs = nullptr, *d_area_xsi = nullptr, *d_area_yofs = nullptr, *d_area_ysi = nullptr;
  float *d_area_xal = nullptr, *d_area_yal = nullptr;
  size_t mask_int_words = 0;

  int pending = 0;               // frames enqueued and not yet waited
  int last_frames = 0;
  // the batch in flight, remembered so that hc_detect_batch_wait can run it again after growing the
  // candidate workspace (the caller keeps its frames valid until the wait returns)
  const uint8_t* inflight_src = nullptr;
  size_t inflight_frame_stride = 0, inflight_row_step = 0;
  // Re-running needs the frames as they were at the enqueue.  Staging planes of the context (host input, colour input,
  // the detectFaces gray plane) always qualify; a caller-owned device buffer only when the caller has promised to
  // keep it unmodified until the wait returns (hc_context_set_rerun_on_overflow).
  bool rerun_caller_frames = false;
  const void* prep_src = nullptr;        // leader of a shared pass: the frame pointer its caller passed for the current batch
  bool taps_enabled = false, timing_enabled = false, timed_last = false;
  bool sq_valid = true;           // the last batch left the squares' integral in d_sq (the fused prep kernel keeps it in LDS unless a tap asks)
  hipEvent_t ev[HC_NKERNELS + 1] = {};
  float last_ms[HC_NKERNELS] = {};
  // Raw candidates of the last waited batch: the published keys are copied out at the wait and sorted / decoded into
  // last_raw only when somebody asks (hc_batch_raw_candidates, a frame the device left to the host, host grouping)
  std::vector<unsigned long long> raw_keys;
  bool raw_decoded = true;
  std::vector<hc_rect> last_raw;
  std::vector<int32_t> last_raw_ofs;
  // K6: rectangle grouping on the device (group_kernels.hip).  Per-frame bins of raw keys, the level table the decode
  // needs, and per-frame output slots + box counts in coherent page-locked memory (made on first use).
  bool device_group = true;              // hc_context_set_device_grouping (initially Knobs::device_group)
  bool group_ran = false;                // the batch in flight groups its own rectangles
  int last_group_host_frames = 0;        // frames of the last waited batch that were grouped on the host
  unsigned long long* d_gbins = nullptr;
  hc::GroupLevel* d_glevels = nullptr;
  hc::GroupBox* h_gboxes = nullptr;
  unsigned int* h_gcount = nullptr;
  std::unique_ptr<WorkerPool> pool;   // host post-processing workers (created on first multi-frame wait)

  void free_workspace();
};

namespace hc {

// ---- api_core.cpp
int set_err(int code, const std::string& what);     // records the thread's error text, returns `code`
const std::string& last_err();
int alloc_bytes(void** p, size_t bytes);             // counted hipMalloc
int upload_bytes(void** p, const void* src, size_t bytes);
template <class T>
int dev_alloc(T** p, size_t n) { return alloc_bytes((void**)p, (n ? n : 1) * sizeof(T)); }
template <class T>
int dev_upload(T** p, const std::vector<T>& v) { return upload_bytes((void**)p, v.data(), v.size() * sizeof(T)); }
int host_alloc_coherent(void** p, size_t bytes);     // counted hipHostMalloc(coherent | mapped)

// ---- api_model.cpp
int upload_model(hc_context* c);
void free_model_tables(hc_context* c);
unsigned long long model_hash(const Model& m);
bool weights_are_small_integers(const Model& m);

// ---- api_plugins.cpp
void ensure_env_plugins();
int plugin_load(const char* path);

// ---- api_context.cpp: route policy
bool wants_fused_variance(const hc_context* c);
bool fused_kernel_has_rings(const hc_context* c);   // either form of the fused kernel produces this context's factors from rings
bool use_fused_prep(const hc_context* c, int nframes, bool fused_variance);
PrepKnobs prep_knobs(const hc_context* c);
int ensure_frames(hc_context* c);
int ensure_pyr(hc_context* c);
int ensure_taps(hc_context* c);
int ensure_group(hc_context* c);

// ---- api_enqueue.cpp
int shared_enqueue_check(hc_context* c, const void* frames, int nframes);
int enqueue_device(hc_context* c, const uint8_t* d_frames, size_t frame_stride, size_t row_step, int nframes);

// ---- api_wait.cpp
int wait_event_polling(hipEvent_t ev);
int wait_core(hc_context* ctx, std::vector<std::vector<hc_rect>>& grouped);
int group_threshold(const hc_context* ctx);    // 0: the raw rectangles are the result (nothing to group)
void decode_raw_public(hc_context* ctx);
int emit_rects(const std::vector<std::vector<hc_rect>>& grouped, hc_rect* out, int out_capacity, int32_t* offsets);

// ---- api_faces.cpp
int faces_wait_core(hc_context* ctx, std::vector<std::vector<hc_rect>>& grouped);

// ---- api_sync.cpp
int sync_context(hc_cascade* c, int rows, int cols, double scale_factor, int min_neighbors, int min_w, int min_h, int max_w,
                 int max_h, hc_context** out, int semantics = 0);

// ---- the 2.4 generation's launchers (cv24_cascade.hip)
int launch_cascade_cv24(hipStream_t st, int nframes, bool tree, int rows, int nwaves, size_t lds_bytes, const CascadeArgs& args,
                        int* compiled = nullptr);
int launch_cv24_tree(hipStream_t st, int nframes, int nwaves, size_t lds_bytes, const CascadeArgs& args);
int launch_cv24_variance(hipStream_t st, int nframes, int nitems, const DevLevel* levels, const WorkItem* items, const uint32_t* sum,
                         const uint32_t* sq, size_t int_stride, int norm_w, int norm_h, double inv_area, double* vnf, size_t grid_stride);

}  // namespace hc

#define HIPCHK(expr)                                                                             \
  do {                                                                                           \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess)                                                                       \
      return hc::set_err(HC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));        \
  } while (0)
