// hc_api.cpp -- C ABI of the MI355X Haar-cascade backend (include/hip_cascade.h).
//
// Replaces, behind a plain-C boundary, the two OpenCV calls the reference makes on its
// cv::CascadeClassifier: load (src/file_io.cc:11) and detectMultiScale
// (src/tracker/detector.cc:25-31).  No CPU fallback exists in this library: every compute
// entry point needs a HIP device and reports HC_ERR_NO_DEVICE / HC_ERR_HIP otherwise.

#include <hip/hip_runtime_api.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <chrono>
#include <thread>
#include <memory>
#include <string>
#include <vector>

#include "../../include/hip_cascade.h"
#include "cascade_model.h"
#include "group_rects.h"
#include "hc_internal.h"
#include "kernels.h"
#include "legacy_cv24.h"
#include "legacy_cv24_host.h"
#include "scale_plan.h"

namespace hc {
int launch_cascade_cv24(hipStream_t st, int nframes, bool tree, int rows, int nwaves, size_t lds_bytes, const CascadeArgs& args);
int launch_cv24_tree(hipStream_t st, int nframes, int nwaves, size_t lds_bytes, const CascadeArgs& args);
int launch_cv24_variance(hipStream_t st, int nframes, int nitems, const DevLevel* levels, const WorkItem* items, const uint32_t* sum,
                         const uint32_t* sq, size_t int_stride, int norm_w, int norm_h, double inv_area, double* vnf, size_t grid_stride);
}

namespace {

thread_local std::string g_err;

int set_err(int code, const std::string& what) { g_err = what; return code; }

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess)                                                                   \
      return set_err(HC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));        \
  } while (0)

const char* const kKernelNames[HC_NKERNELS] = {"k_pyramid_integral", "k_integral_rows", "k_integral_cols", "k_variance",
                                               "k_cascade"};

std::atomic<long long> g_alloc_count{0};

template <class T>
int dev_alloc(T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  g_alloc_count++;
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) return set_err(HC_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
  return HC_OK;
}

template <class T>
int dev_upload(T** p, const std::vector<T>& v) {
  int rc = dev_alloc(p, v.size());
  if (rc) return rc;
  if (!v.empty()) HIPCHK(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return HC_OK;
}

}  // namespace

// Small persistent worker pool for the per-frame host post-processing (candidate decode +
// rectangle grouping): frames are independent, so a batch is a parallel-for over frames.
class WorkerPool {
 public:
  explicit WorkerPool(int nthreads) {
    for (int i = 0; i < nthreads; i++) workers_.emplace_back([this] { loop(); });
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> l(mu_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  // calls fn(i) for i in [0, n), on the pool plus the calling thread; returns when all are done
  void parallel_for(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    { std::lock_guard<std::mutex> l(mu_); fn_ = &fn; n_ = n; next_ = 0; active_ = (int)workers_.size(); ++gen_; }
    cv_.notify_all();
    drain();
    std::unique_lock<std::mutex> l(mu_);
    done_.wait(l, [this] { return active_ == 0; });
    fn_ = nullptr;
  }

 private:
  void drain() {
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= n_) break;
      (*fn_)(i);
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
      }
      drain();
      { std::lock_guard<std::mutex> l(mu_); if (--active_ == 0) done_.notify_all(); }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> next_{0};
  int n_ = 0, active_ = 0;
  unsigned long long gen_ = 0;
  bool stop_ = false;
};

namespace hc {
int set_error(int code, const std::string& what) { return set_err(code, what); }   // for the other translation units
void count_alloc() { g_alloc_count++; }
int deliver_sync_result(hc_cascade* c, const std::vector<hc_rect>& boxes, hc_rect* out, int out_capacity, int* out_count) {
  c->last_result_valid = false;
  *out_count = (int)boxes.size();
  if ((int)boxes.size() > out_capacity || (!out && !boxes.empty())) {
    c->last_result = boxes;            // the detection is done: the caller re-reads it instead of running it again
    c->last_result_valid = true;
    return set_err(HC_ERR_CAPACITY, "output rectangle buffer too small (*out_count holds the size needed; "
                                    "hc_detect_multiscale_reread returns the rectangles without running the detection again)");
  }
  for (size_t i = 0; i < boxes.size(); i++) out[i] = boxes[i];
  return HC_OK;
}
}

struct hc_context {
  const hc_cascade* casc = nullptr;
  int device = 0;
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
  // (k_publish_flag); the candidates themselves arrive in h_cand the same way.  wait_core polls the sequence word.
  unsigned int* h_pub = nullptr;
  unsigned int pub_seq = 0;
  hipStream_t side_stream = nullptr;   // small batches: the ystep-2 cascade launch runs beside the ystep-1 launch
  // Large host batches go up on a stream of their own (made on first use) and `stream` waits for ev_uploaded: contexts
  // that share one stream (kernels of consecutive batches back to back, the fastest arrangement) still get batch i+1's
  // frames across PCIe while batch i computes.
  hipStream_t upload_stream = nullptr;
  hipEvent_t ev_uploaded = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int side_max_batch = 8;              // ... up to this many frames (a full batch fills the chip with one launch)

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
  // several cascades over the same frames (hc_context_configure_shared): the leader owns the pyramid integrals of
  // the union ladder, followers read them.  prep_seq counts the leader's enqueues; a follower scans each once.
  hc_context* leader = nullptr;
  std::vector<hc_context*> followers;
  unsigned long long prep_seq = 0, seen_seq = 0;
  int prep_frames = 0;
  hipEvent_t ev_prep = nullptr;        // leader: recorded after the prep kernels of a batch
  hipEvent_t ev_var = nullptr;         // leader: recorded after its k_variance (followers with the leader's window read its factors)
  bool vnf_shared = false;             // follower whose window and grid equal the leader's: d_vnf IS the leader's, no k_variance of its own
  bool scan_in_flight = false;         // follower: ev_done of its last scan has been recorded and not yet waited by the leader
  int max_batch = 0, cand_per_frame = 0, min_neighbors = 0;
  size_t cand_cap = 0;                   // raw-candidate capacity of d_cand / h_cand for a whole batch (grows on demand)
  // cascade-kernel tiling per ystep class (index 0: ystep 1, index 1: ystep 2)
  int band_rows[2] = {16, 16};
  int band_waves[2] = {4, 8};            // wavefronts per cascade workgroup (4 or 8); measured best on MI355X
  hc::TileGeom geom[2] = {};
  size_t lds_bytes[2] = {0, 0};
  int tile_words[2] = {0, 0};
  int sparse_max = 32;
  // k_pyramid_integral (one workgroup marching down each level: least HBM traffic, best for full batches) or
  // k_resize + k_integral_rows + k_integral_cols (fully parallel: 0.07 instead of 0.39 ms for ONE 720p frame).
  // -1: by batch size (fused above fused_min_batch frames), 0: never fused, 1: always fused (HC_FUSED_PREP)
  int fused_prep = -1;
  bool allow_int_weights = true; // HC_INT_WEIGHTS=0: always the f32 feature arithmetic (A/B, debugging)
  bool bank_order = true;        // HC_BANK_ORDER=0: the sparse finish reads the stumps in the dense path's order
  bool allow_spec = true;        // HC_SPEC=0: never the compiled early stages (csrc/spec), always the generic kernel
  bool pair_form = true;         // HC_PAIR_FORM=0: tree stages always walk their trees per lane (A/B, debugging)
  int frame_chunk = 128;         // HC_FRAME_CHUNK: k_cascade walks a large batch in chunks of this many frames (0: one sweep over all frames)
  bool fuse_variance = true;     // HC_FUSE_VARIANCE=0: always the separate k_variance kernel (A/B, debugging)
  bool fuse_variance_forced = false;   // HC_FUSE_VARIANCE=1: inside the fused prep kernel whatever route that takes
  double fused_min_batch = -1;   // HC_FUSED_MIN_BATCH: a fixed threshold in 720p-frame equivalents instead of the cost model of use_fused_prep()
  int row_stages = 2;            // stages 0 and 1 in row form (conflict-free LDS reads while 88 % of the windows are alive)
  int debug_max_stages = 0;   // HC_DEBUG_MAX_STAGES: truncate the cascade (wrong results; kernel-time experiments only)

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
  hc::WorkItem *d_resize_items = nullptr, *d_row_items = nullptr, *d_col_items = nullptr, *d_var_items = nullptr,
               *d_band_items[2] = {nullptr, nullptr}, *d_seg_items[2] = {nullptr, nullptr};
  int n_seg_items[2] = {0, 0};           // bands cut into segments of seg_tiles tiles: the work list of small batches
  int seg_tiles = 1, seg_max_batch = 8;  // HC_SEG_TILES, HC_SEG_MAX_BATCH (measured: tools/exp_small_batch.sh)
  hc::DevNode* d_nodes[2] = {nullptr, nullptr};   // corner offsets resolved for geom[c]
  hc::DevNode* d_nodes_sparse[2] = {nullptr, nullptr};   // the same records, every exact stage in bank-spreading order (sparse finish)
  bool int_weights = false;              // every node: w0 == -1, w1 / w2 integers, weighted rect sums below 2^24 (upload_model)
  int spec_id = 0;                       // compiled early stages matching this model (hc::spec_lookup), 0: none
  // 16-bit-tile build of the scan-step-1 launch (cascade_body.inc HC_P16): its own tile geometry, node tables (2-byte
  // entry offsets, rects of more than 257 pixels split in two) and compiled unit; whole-band launches only
  bool p16 = false, allow_p16 = true;    // HC_P16=0: never
  int spec_id_p16 = 0;
  hc::TileGeom geom_p16 = {};
  size_t lds_bytes_p16 = 0;
  int tile_words_p16 = 0;
  hc::DevNode *d_nodes_p16 = nullptr, *d_nodes_sparse_p16 = nullptr;
  hc::DevStageKinds* d_kinds_p16 = nullptr;
  int last_compiled = 0;                 // stages that ran as compiled code in EVERY cascade launch of the last enqueued batch
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
  std::vector<hc_rect> last_raw;
  std::vector<int32_t> last_raw_ofs;
  std::unique_ptr<WorkerPool> pool;   // host post-processing workers (created on first multi-frame wait)
  int host_threads = 4;   // 1 thread already hides the 64-frame grouping under the next batch (tools/exp_host_threads.sh)

  void free_workspace() {
    if (leader) {                      // follower: the integrals belong to the leader
      d_sum = d_sq = nullptr;
      if (vnf_shared) d_vnf = nullptr;
      vnf_shared = false;
      auto& f = leader->followers;
      f.erase(std::remove(f.begin(), f.end(), this), f.end());
      leader = nullptr;
    }
    for (hc_context* f : followers) {  // leader going away: followers lose their integrals and must be reconfigured
      f->d_sum = f->d_sq = nullptr;
      if (f->vnf_shared) f->d_vnf = nullptr;
      f->vnf_shared = false;
      f->leader = nullptr;
      f->configured = false;
    }
    followers.clear();
    void* ptrs[] = {d_frames, d_color, d_pyr, d_sum, d_sq, d_vnf, d_taps, d_levels, d_xtab, d_ytab, d_resize_items,
                    d_row_items, d_col_items, d_var_items, d_band_items[0], d_band_items[1], d_seg_items[0], d_seg_items[1], d_cand,
                    d_count, d_area_xofs, d_area_xsi, d_area_xal, d_area_yofs, d_area_ysi, d_area_yal};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    d_frames = d_color = d_pyr = nullptr; color_bytes = 0;
    d_area_xofs = d_area_xsi = d_area_yofs = d_area_ysi = nullptr; d_area_xal = d_area_yal = nullptr; area_general = false; d_sum = d_sq = nullptr; d_vnf = nullptr; d_taps = nullptr; d_levels = nullptr;
    d_xtab = d_ytab = nullptr;
    d_resize_items = d_row_items = d_col_items = d_var_items = d_band_items[0] = d_band_items[1] = nullptr;
    d_seg_items[0] = d_seg_items[1] = nullptr;
    d_cand = nullptr; d_count = nullptr;
    if (h_cand) (void)hipHostFree(h_cand);
    if (h_pub) (void)hipHostFree(h_pub);
    h_cand = nullptr; h_pub = nullptr; pub_seq = 0;
    if (d_src_bgr) (void)hipFree(d_src_bgr);
    d_src_bgr = nullptr; src_rows = src_cols = 0;
    void* cptrs[] = {d_vnf64, d_cvlevels, d_cvtab_ofs, d_cvtab_a0, d_cvtab_a1};
    for (void* q : cptrs) if (q) (void)hipFree(q);
    d_vnf64 = nullptr; d_cvlevels = nullptr; d_cvtab_ofs = nullptr; d_cvtab_a0 = d_cvtab_a1 = nullptr;
    void* fptrs[] = {d_bgr, d_gray, d_mask, d_lut, d_hist, d_msum, d_boxsums, d_boxes, d_idlevel, d_idx, d_idy,
                     d_mask_items[0], d_mask_items[1]};
    for (void* q : fptrs) if (q) (void)hipFree(q);
    d_bgr = d_gray = d_mask = d_lut = nullptr; d_hist = d_msum = d_boxsums = nullptr; d_boxes = nullptr;
    d_idlevel = nullptr; d_idx = d_idy = nullptr; d_mask_items[0] = d_mask_items[1] = nullptr;
    if (h_boxes) (void)hipHostFree(h_boxes);
    if (h_boxsums) (void)hipHostFree(h_boxsums);
    h_boxes = nullptr; h_boxsums = nullptr;
    faces_enabled = faces_pending = false;
    configured = false;
  }
};

namespace {

// LDS tile geometry + per-geometry corner offsets of every node
hc::TileGeom make_geom(const hc::Model& m, int ystep, int band_rows) {
  hc::TileGeom g{};
  g.cols = (hc::TILE_POS - 1) * ystep + m.win_w + 1;
  g.rows = (band_rows - 1) * ystep + m.win_h + 1;
  if (ystep == 1) { g.half = 0; g.pitch = g.cols; }
  else { g.half = (g.cols + 1) / 2; g.pitch = 2 * g.half; }
  return g;
}

void build_nodes(const hc::Model& m, const hc::TileGeom& g, int ystep, std::vector<hc::DevNode>& out, bool p16 = false) {
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
int node_kind(const hc::ModelNode& n) {
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
bool stage_sum_is_exact(const hc::Model& m, const hc::ModelStage& s) {
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
void order_stage_for_banks(std::vector<hc::DevNode>& nodes, int first, int count) {
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

// compiled-stage units from HC_SPEC_PLUGINS (colon-separated shared objects), loaded once before the first lookup
int plugin_load(const char* path);
void ensure_env_plugins() {
  static std::once_flag once;
  std::call_once(once, [] {
    const char* e = getenv("HC_SPEC_PLUGINS");
    if (!e) return;
    std::string s(e);
    size_t a = 0;
    while (a <= s.size()) {
      const size_t b = s.find(':', a);
      const std::string one = s.substr(a, b == std::string::npos ? std::string::npos : b - a);
      if (!one.empty() && plugin_load(one.c_str()) != HC_OK) fprintf(stderr, "hip_cascade: HC_SPEC_PLUGINS: %s\n", g_err.c_str());
      if (b == std::string::npos) break;
      a = b + 1;
    }
  });
}

int plugin_load(const char* path) {
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return set_err(HC_ERR_IO, std::string("cannot load the compiled-stage plug-in: ") + dlerror());
  typedef int (*abi_fn)(void);
  typedef const void* (*unit_fn)(void);
  const abi_fn abi = (abi_fn)dlsym(h, "hc_spec_plugin_abi");
  const unit_fn unit = (unit_fn)dlsym(h, "hc_spec_plugin_unit");
  if (!abi || !unit) { dlclose(h); return set_err(HC_ERR_FORMAT, std::string(path) + " is not a hip_cascade compiled-stage plug-in"); }
  if (abi() != hc::SPEC_PLUGIN_ABI) {
    dlclose(h);
    return set_err(HC_ERR_UNSUPPORTED, std::string(path) + " was built against another version of the kernel sources: rebuild it "
                                                             "(tools/build_spec_plugin.sh)");
  }
  if (hc::spec_register((const hc::SpecUnit*)unit()) != 0) { dlclose(h); return set_err(HC_ERR_FORMAT, std::string(path) + ": empty unit"); }
  return HC_OK;                       // the object stays loaded for the life of the process: its kernels are launched from it
}

void free_model_tables(hc_context* c) {
  void* ptrs[] = {c->d_nodes[0], c->d_nodes[1], c->d_nodes_sparse[0], c->d_nodes_sparse[1], c->d_stages, c->d_trees, c->d_leaves,
                  c->d_kinds, c->d_cvnodes, c->d_cvtrees, c->d_cvstages, c->d_tree_pass, c->d_tree_fail, c->d_tree_order,
                  c->d_nodes_p16, c->d_nodes_sparse_p16, c->d_kinds_p16};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  c->d_tree_pass = c->d_tree_fail = c->d_tree_order = nullptr;
  c->d_nodes_p16 = c->d_nodes_sparse_p16 = nullptr; c->d_kinds_p16 = nullptr; c->p16 = false; c->spec_id_p16 = 0;
  c->d_nodes[0] = c->d_nodes[1] = c->d_nodes_sparse[0] = c->d_nodes_sparse[1] = nullptr;
  c->d_stages = nullptr; c->d_trees = nullptr; c->d_leaves = nullptr; c->d_kinds = nullptr;
  c->d_cvnodes = nullptr; c->d_cvtrees = nullptr; c->d_cvstages = nullptr;
}

// every corner offset of the 16-bit tile fits the 16-bit offset fields (trivially, for windows this build accepts)
bool node_kinds_fit_p16(const hc::Model& m) { return m.win_w <= 40 && m.win_h <= 40; }

// The cascade with every rect of more than 257 pixels cut into two halves (same weight; the second half takes the
// node's free third slot): rect sums modulo 2^16 are then exact for every rect, and the feature value -- an integer
// expression, weights_are_small_integers -- is unchanged.  False when some node has no free slot or a half is still too big.
bool p16_model(const hc::Model& m, hc::Model& out) {
  out = m;
  for (hc::ModelNode& n : out.nodes) {
    int big = -1;
    for (int r = 0; r < 3; r++)
      if (n.weight[r] != 0.f && n.rect[r][2] * n.rect[r][3] > 257) { if (big >= 0) return false; big = r; }
    if (big < 0) continue;
    if (n.weight[2] != 0.f || big == 2) return false;
    const int x = n.rect[big][0], y = n.rect[big][1], w = n.rect[big][2], h = n.rect[big][3];
    if (h >= w) {
      const int h1 = h / 2;
      n.rect[big][3] = (int8_t)h1;
      n.rect[2][0] = (int8_t)x; n.rect[2][1] = (int8_t)(y + h1); n.rect[2][2] = (int8_t)w; n.rect[2][3] = (int8_t)(h - h1);
    } else {
      const int w1 = w / 2;
      n.rect[big][2] = (int8_t)w1;
      n.rect[2][0] = (int8_t)(x + w1); n.rect[2][1] = (int8_t)y; n.rect[2][2] = (int8_t)(w - w1); n.rect[2][3] = (int8_t)h;
    }
    n.weight[2] = n.weight[big];
    if (n.rect[big][2] * n.rect[big][3] > 257 || n.rect[2][2] * n.rect[2][3] > 257) return false;
  }
  return true;
}

// Node tables of the 16-bit-tile build (scan step 1) from the split model: same stages, same leaf words, same ordering
// rules (kind groups, bank-spreading order of the sparse copy) as the standard tables below.
int upload_p16_tables(hc_context* c, const hc::Model& m16, const std::vector<hc::DevStage>& st) {
  std::vector<hc::DevNode> nd;
  build_nodes(m16, c->geom_p16, 1, nd, true);
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
  if (c->bank_order)
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
    c->geom[k] = make_geom(m, ystep, c->band_rows[k]);
    const size_t words = ((size_t)c->geom[k].pitch * c->geom[k].rows + 1) & ~(size_t)1;   // even: 8-byte aligned LDS regions behind the tile
#ifndef HC_OFF_WORDS
    if (words * 4 > 65535) return set_err(HC_ERR_UNSUPPORTED, "window too large for 16-bit LDS byte offsets");
#else
    if (words > 65535) return set_err(HC_ERR_UNSUPPORTED, "window too large for 16-bit LDS offsets");
#endif
    c->tile_words[k] = (int)words;
    // k_cascade LDS layout: tile | vnf[band_rows*64] f32 | 2 queues x band_rows*64 u16 | 3 counters (+pad)
    const size_t vnf_bytes = cv24 ? 8 : 4;      // 2.4 semantics: f64 variance factors
#ifndef HC_PLAIN_QUEUES
    // k_cascade LDS layout: tile | vnf[band_rows*64] f32 | bank matrix + survivor list, band_rows*64 u16 each | 32 counter words + ticket
    c->lds_bytes[k] = words * 4 + (size_t)c->band_rows[k] * hc::TILE_POS * (vnf_bytes + 2 * 2) + 136;
#else
    c->lds_bytes[k] = words * 4 + (size_t)c->band_rows[k] * hc::TILE_POS * (vnf_bytes + 2 * 2) + 16;
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
    if (!m.is_stump() && d.exact && c->pair_form) {
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
      if (s.ntrees > 65535) return set_err(HC_ERR_UNSUPPORTED, "stage too long");
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
  c->int_weights = !cv24 && c->allow_int_weights && weights_are_small_integers(m);
  c->spec_id = c->int_weights && c->allow_spec ? hc::spec_lookup(model_hash(m), m.win_w, m.win_h, nullptr) : 0;
  if (cv24) {
    c->cv_chain = hc::cv24_is_chain(m);
    if (!c->cv_chain) {           // a real stage tree: tables of the window-at-a-time kernel
      std::vector<hc::Cv24Node> cn;
      std::vector<hc::Cv24Tree> ct;
      std::vector<hc::Cv24Stage> cs;
      int rc0 = hc::cv24_tree_tables(m, cn, ct, cs);
      if (rc0 != HC_OK) return rc0;
      if ((rc0 = dev_upload(&c->d_cvnodes, cn)) || (rc0 = dev_upload(&c->d_cvtrees, ct)) || (rc0 = dev_upload(&c->d_cvstages, cs))) return rc0;
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
    if (m.is_stump() && c->bank_order)
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
  if (!cv24 && c->allow_p16 && c->allow_spec && c->int_weights && m.is_stump() && c->band_rows[0] == 16 && c->band_waves[0] == 4 &&
      c->row_stages <= 2 && !st.empty() && st[0].exact == 2) {
    int pst = 0;
    const int id = hc::spec_lookup(model_hash(m), m.win_w, m.win_h, &pst, true);
    hc::Model m16;
    if (id > 0 && p16_model(m, m16)) {
      hc::TileGeom g = make_geom(m, 1, 16);
      g.pitch = (g.cols + 1) & ~1;
      const size_t tile_bytes = ((size_t)g.pitch * g.rows * 2 + 7) & ~(size_t)7;
      const size_t fixed = (size_t)16 * hc::TILE_POS * (4 + 6 + 2) + 264;  // vnf | copy B (6 KB; the list later) | bank matrix | 64 counters + ticket
      if (tile_bytes <= (size_t)16 * hc::TILE_POS * 6 && node_kinds_fit_p16(m16)) {
        c->geom_p16 = g;
        c->tile_words_p16 = (int)(tile_bytes / 4);
        c->lds_bytes_p16 = fixed + tile_bytes;
        if ((rc = upload_p16_tables(c, m16, st))) return rc;
        c->spec_id_p16 = id;
        c->p16 = true;
      }
    }
  }
  return HC_OK;
}

hc_level to_public(const hc::DevLevel& L) {
  hc_level o{};
  o.scale = L.scale; o.w = L.w; o.h = L.h; o.ystep = L.ystep; o.szw_w = L.szw_w; o.szw_h = L.szw_h;
  o.y_end = L.y_end; o.nx = L.nx; o.ny = L.ny; o.win_w = L.win_w; o.win_h = L.win_h; o.int_pitch = L.int_pitch;
  o.pyr_ofs = L.pyr_ofs; o.int_ofs = L.int_ofs; o.grid_ofs = L.grid_ofs;
  return o;
}

// A group that shares one prep pass is enqueued leader first, then every follower once, with the same batch.
int shared_enqueue_check(hc_context* c, const void* frames, int nframes) {
  if (c->leader) {
    hc_context* L = c->leader;
    if (L->prep_seq == c->seen_seq) return set_err(HC_ERR_STATE, "shared prep: enqueue the leader's batch first (once per follower scan)");
    if (L->prep_frames != nframes) return set_err(HC_ERR_ARG, "shared prep: follower batch size differs from the leader's");
    // A follower scans whatever the leader's LAST prep pass left in the integrals.  A leader that was enqueued again
    // before this follower scanned the previous batch would make it silently return the newer batch's rectangles for
    // the older frames: the frame pointer names the batch, so a follower enqueued with another one is refused.
    if (L->prep_src != frames)
      return set_err(HC_ERR_STATE, "shared prep: the follower was enqueued with a batch other than the leader's current one "
                                   "(per batch: the leader first, then every follower with the same frame pointer)");
    c->seen_seq = L->prep_seq;
  } else if (!c->followers.empty()) {
    c->prep_seq++;
    c->prep_frames = nframes;
    c->prep_src = frames;
  }
  return HC_OK;
}

// Which prep route a batch takes.  The fused kernel moves the fewest bytes and wins at full batches, but it is one
// workgroup marching down each level, so its time has a floor -- the march down the LARGEST level -- that grows faster
// than the level's area (more columns per thread, fewer workgroups per CU); the three parallel kernels have no such
// floor.  Measured on MI355X (ms; equiv = batch size in 1280x720-frame equivalents, A = area of level 0 relative to
// 795x447, the first level of a 720p frame): fused 0.0088 / equiv (0.0135 with the variance factors fused in), floor
// 0.4 * A^1.45 (0.6 * ..: one 720p frame 0.39, sixteen 3840x2160 frames 18.6); parallel kernels 0.0233 / equiv, plus
// 0.005 / equiv for k_variance where the fused kernel would have produced the factors.
// Variance factors inside the fused prep kernel: a context on its own (a shared pass serves cascades with different windows,
// which run k_variance each), and only where the kernel has its ring route -- the read-back route of wider levels loses to a
// separate k_variance launch (HC_FUSE_VARIANCE=1 forces it, =0 never fuses).
bool wants_fused_variance(const hc_context* c) {
  if (!c->fuse_variance || !c->followers.empty()) return false;
  if (c->fuse_variance_forced) return true;
  int max_w = 0;
  for (const hc::DevLevel& L : c->plan.levels) max_w = std::max(max_w, L.w);
  const hc::Model& m = c->casc->model;
  return hc::pyramid_integral_has_ring(max_w, c->plan.cols, m.win_w - 2, m.win_h - 2);
}

bool use_fused_prep(const hc_context* c, int nframes, bool fused_variance) {
  if (c->fused_prep >= 0) return c->fused_prep > 0;                         // HC_FUSED_PREP pins the route
  const hc::Plan& P = c->plan;
  const double equiv = (double)nframes * (double)P.rows * (double)P.cols / (1280.0 * 720.0);
  if (c->fused_min_batch >= 0) return equiv > c->fused_min_batch;           // HC_FUSED_MIN_BATCH: plain threshold
  if (P.levels.empty()) return false;
  const double A = (double)P.levels[0].w * (double)P.levels[0].h / (795.0 * 447.0);
  const double t_fused = std::max(equiv * (fused_variance ? 0.0135 : 0.0088), (fused_variance ? 0.6 : 0.4) * std::pow(A, 1.45));
  const double t_parallel = equiv * (fused_variance ? 0.0283 : 0.0233);
  return t_fused < t_parallel;
}

// first-use allocations (outside the steady state: a context that has run one batch of a kind never allocates again)
int ensure_frames(hc_context* c) {
  if (c->d_frames) return HC_OK;
  return dev_alloc(&c->d_frames, (size_t)c->max_batch * c->plan.rows * c->plan.cols);
}
int ensure_pyr(hc_context* c) {
  if (c->d_pyr) return HC_OK;
  return dev_alloc(&c->d_pyr, (size_t)c->max_batch * c->plan.pyr_bytes);
}
int ensure_taps(hc_context* c) {
  if (c->d_taps) return HC_OK;
  return dev_alloc(&c->d_taps, (size_t)c->max_batch * c->plan.grid_elems);
}

int enqueue_device(hc_context* c, const uint8_t* d_frames, size_t frame_stride, size_t row_step, int nframes) {
  const hc::Plan& P = c->plan;
  const hc::Model& m = c->casc->model;
  hipStream_t s = c->stream;
  c->inflight_src = d_frames; c->inflight_frame_stride = frame_stride; c->inflight_row_step = row_step;
  for (hc_context* f : c->followers)   // the next prep pass overwrites the integrals: the followers' scans of the last one come first
    if (f->scan_in_flight) { HIPCHK(hipStreamWaitEvent(s, f->ev_done, 0)); f->scan_in_flight = false; }
  {
    const bool fused0 = use_fused_prep(c, nframes, wants_fused_variance(c));
    int rc0;
    if (!c->leader && !P.levels.empty() && (c->taps_enabled || !fused0 || c->semantics == 1) && (rc0 = ensure_pyr(c))) return rc0;
    if (!P.levels.empty() && c->taps_enabled && (rc0 = ensure_taps(c))) return rc0;
  }
  HIPCHK(hipMemsetAsync(c->d_count, 0, sizeof(unsigned int), s));
  const bool timed = c->timing_enabled;
  c->timed_last = timed;
  if (!P.levels.empty()) {
    const size_t pyr_stride = (size_t)P.pyr_bytes, int_stride = (size_t)P.int_words, grid_stride = (size_t)P.grid_elems;
    if (c->taps_enabled) HIPCHK(hipMemsetAsync(c->d_taps, 0xFE, grid_stride * nframes, s));
    int e;
    bool fused_variance = false;
    c->sq_valid = true;
    if (timed) HIPCHK(hipEventRecord(c->ev[0], s));
    // batch size in 720p-frame equivalents decides the small-batch policies (prep kernels, segments, side stream)
    const double equiv = (double)nframes * (double)P.rows * (double)P.cols / (1280.0 * 720.0);
    bool cv24_fused = false;
    if (c->semantics == 1) {
      // OpenCV 2.4 through the fused prep kernel (cvResize arithmetic, integrals and the f64 variance factors in one march,
      // nothing materialised in between) when the geometry fits its ring route and the batch is large enough for one
      // workgroup per (level, frame); HC_CV24_FUSED=0 pins the four separate kernels.
      static const bool fused_ok = [] { const char* e_ = getenv("HC_CV24_FUSED"); return !e_ || atoi(e_) != 0; }();
      int max_w = 0, max_nx = 0;
      for (const hc::DevLevel& L : P.levels) { max_w = std::max(max_w, L.w); max_nx = std::max(max_nx, L.nx); }
      cv24_fused = fused_ok && c->cv.fused_tables && (c->cv_chain || c->cv_tree_tiled) && use_fused_prep(c, nframes, true) &&
                   hc::pyramid_integral_takes_cv24(max_w, max_nx, P.cols, m.win_w - 2, m.win_h - 2);
      if (cv24_fused) {
        if ((e = hc::launch_pyramid_integral(s, nframes, (int)P.levels.size(), max_w, d_frames, frame_stride, row_step, P.rows, P.cols,
                                             c->d_levels, c->d_xtab, c->d_ytab, c->d_sum, c->d_sq, int_stride,
                                             c->taps_enabled ? c->d_pyr : nullptr, pyr_stride, (float*)c->d_vnf64, grid_stride, m.win_w - 2,
                                             m.win_h - 2, max_nx, c->taps_enabled, &c->sq_valid, 1)))
          return set_err(HC_ERR_HIP, std::string("k_pyramid_integral (2.4) launch: ") + hipGetErrorString((hipError_t)e));
        if (timed) { HIPCHK(hipEventRecord(c->ev[1], s)); HIPCHK(hipEventRecord(c->ev[2], s)); }
        fused_variance = true;
      }
    }
    if (c->semantics == 1 && !cv24_fused) {
      // OpenCV 2.4: cvResize pyramid (materialised), the two-pass integrals, f64 variance factors
      if ((e = hc::launch_cv24_resize(s, nframes, (int)P.levels.size(), c->cv.max_pix, d_frames, frame_stride, row_step, P.rows, P.cols,
                                      c->d_cvlevels, c->d_cvtab_ofs, c->d_cvtab_a0, c->d_cvtab_a1, c->d_pyr, pyr_stride)))
        return set_err(HC_ERR_HIP, std::string("k_cv24_resize launch: ") + hipGetErrorString((hipError_t)e));
      if (timed) HIPCHK(hipEventRecord(c->ev[1], s));
      if ((e = hc::launch_integral_rows(s, nframes, (int)P.row_items.size(), c->d_pyr, pyr_stride, c->d_levels, c->d_row_items, c->d_sum,
                                        c->d_sq, int_stride)))
        return set_err(HC_ERR_HIP, std::string("k_integral_rows launch: ") + hipGetErrorString((hipError_t)e));
      if (timed) HIPCHK(hipEventRecord(c->ev[2], s));
      if ((e = hc::launch_integral_cols(s, nframes, (int)P.col_items.size(), c->d_levels, c->d_col_items, c->d_sum, c->d_sq, int_stride)))
        return set_err(HC_ERR_HIP, std::string("k_integral_cols launch: ") + hipGetErrorString((hipError_t)e));
      fused_variance = true;           // the generic k_variance below is the >= 4 gate; 2.4 has its own factor
    } else if (c->semantics == 1) {
      // (done by the fused kernel above)
    } else if (c->leader) {
      // shared prep: the leader's kernels have produced (or are producing) the integrals of this very batch
      HIPCHK(hipStreamWaitEvent(s, c->leader->ev_prep, 0));
      if (c->vnf_shared) HIPCHK(hipStreamWaitEvent(s, c->leader->ev_var, 0));   // the leader's variance factors of this batch
      if (timed) { HIPCHK(hipEventRecord(c->ev[1], s)); HIPCHK(hipEventRecord(c->ev[2], s)); }
    } else if (use_fused_prep(c, nframes, wants_fused_variance(c))) {
      int max_w = 0, max_nx = 0;
      for (const hc::DevLevel& L : P.levels) { max_w = std::max(max_w, L.w); max_nx = std::max(max_nx, L.nx); }
      // the pyramid bytes only exist as a debug tap in this mode
      // a context on its own gets its variance factors from the same kernel (no k_variance launch, no second pass
      // over the integrals); a shared prep pass serves cascades with different windows, which run k_variance each
      fused_variance = wants_fused_variance(c);
      if ((e = hc::launch_pyramid_integral(s, nframes, (int)P.levels.size(), max_w, d_frames, frame_stride, row_step,
                                           P.rows, P.cols, c->d_levels, c->d_xtab, c->d_ytab, c->d_sum, c->d_sq,
                                           int_stride, c->taps_enabled ? c->d_pyr : nullptr, pyr_stride,
                                           fused_variance ? c->d_vnf : nullptr, grid_stride, m.win_w - 2, m.win_h - 2, max_nx,
                                           c->taps_enabled, &c->sq_valid)))
        return set_err(HC_ERR_HIP, std::string("k_pyramid_integral launch: ") + hipGetErrorString((hipError_t)e));
      if (timed) { HIPCHK(hipEventRecord(c->ev[1], s)); HIPCHK(hipEventRecord(c->ev[2], s)); }
    } else {
      if ((e = hc::launch_resize(s, nframes, (int)P.resize_items.size(), d_frames, frame_stride, row_step, P.rows,
                                 P.cols, c->d_levels, c->d_xtab, c->d_ytab, c->d_resize_items, c->d_pyr, pyr_stride)))
        return set_err(HC_ERR_HIP, std::string("k_resize launch: ") + hipGetErrorString((hipError_t)e));
      if (timed) HIPCHK(hipEventRecord(c->ev[1], s));
      if ((e = hc::launch_integral_rows(s, nframes, (int)P.row_items.size(), c->d_pyr, pyr_stride, c->d_levels,
                                        c->d_row_items, c->d_sum, c->d_sq, int_stride)))
        return set_err(HC_ERR_HIP, std::string("k_integral_rows launch: ") + hipGetErrorString((hipError_t)e));
      if (timed) HIPCHK(hipEventRecord(c->ev[2], s));
      if ((e = hc::launch_integral_cols(s, nframes, (int)P.col_items.size(), c->d_levels, c->d_col_items, c->d_sum,
                                        c->d_sq, int_stride)))
        return set_err(HC_ERR_HIP, std::string("k_integral_cols launch: ") + hipGetErrorString((hipError_t)e));
    }
    if (!c->followers.empty()) HIPCHK(hipEventRecord(c->ev_prep, s));
    if (timed) HIPCHK(hipEventRecord(c->ev[3], s));
    if (!fused_variance && !c->vnf_shared &&
        (e = hc::launch_variance(s, nframes, (int)P.var_items.size(), c->d_levels, c->d_var_items, c->d_sum, c->d_sq,
                                 int_stride, m.win_w - 2, m.win_h - 2, c->d_vnf, grid_stride)))
      return set_err(HC_ERR_HIP, std::string("k_variance launch: ") + hipGetErrorString((hipError_t)e));
    if (!c->followers.empty()) HIPCHK(hipEventRecord(c->ev_var, s));
    if (c->semantics == 1 && !cv24_fused &&
        (e = hc::launch_cv24_variance(s, nframes, (int)P.var_items.size(), c->d_levels, c->d_var_items, c->d_sum, c->d_sq, int_stride,
                                      m.win_w - 2, m.win_h - 2, c->cv.inv_area, c->d_vnf64, grid_stride)))
      return set_err(HC_ERR_HIP, std::string("k_cv24_variance launch: ") + hipGetErrorString((hipError_t)e));
    if (timed) HIPCHK(hipEventRecord(c->ev[4], s));
    if (c->semantics == 1 && !c->cv_chain && !c->cv_tree_tiled) {
      // a real stage tree of a cascade the tiled tree kernel does not take: window-at-a-time evaluation that walks parent / next / child
      hc::Cv24Args T{};
      T.levels = c->d_cvlevels; T.sum = c->d_sum; T.sq = c->d_sq; T.int_stride = int_stride; T.grid_stride = grid_stride;
      T.stages = c->d_cvstages; T.trees = c->d_cvtrees; T.nodes = c->d_cvnodes; T.leaves = c->d_leaves; T.inv_area = c->cv.inv_area;
      T.win_w = m.win_w; T.win_h = m.win_h; T.cand = c->d_cand; T.cand_count = c->d_count; T.cand_cap = (unsigned int)c->cand_cap;
      T.taps = c->taps_enabled ? c->d_taps : nullptr;
      if ((e = hc::launch_cv24_cascade(s, nframes, (int)P.levels.size(), c->cv.max_pos, T)))
        return set_err(HC_ERR_HIP, std::string("k_cv24_cascade launch: ") + hipGetErrorString((hipError_t)e));
    }
    // With one or two frames neither cascade launch fills 256 CUs and a band walk is a serial chain of tiles:
    // the two launches (independent: different levels) then run side by side on two streams.
    const bool side = equiv <= c->side_max_batch && !timed && !P.band_items[0].empty() && !P.band_items[1].empty();
    if (side) {
      if (!c->side_stream) {
        // a HIGH-priority stream: the runtime keeps those on hardware queues apart from the normal ones, so the side
        // launch can never land on the queue of `stream` itself (where "side by side" would be one after the other plus
        // two event hops: 0.33 instead of 0.28 ms for a 720p frame, depending on how many streams the process had made)
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (const char* e_ = getenv("HC_SIDE_PRIO")) { if (atoi(e_) == 0) hi = lo; }
        HIPCHK(hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, hi));
      }
      HIPCHK(hipEventRecord(c->ev_fork, s));
      HIPCHK(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
    }
    int launches = 0;
    c->last_compiled = 0;
    for (int k = 1; k >= 0; k--) {      // ystep-2 levels (the largest) first
      if (c->semantics == 1 && !c->cv_chain && !c->cv_tree_tiled) break;           // done by the window-at-a-time kernel above
      hc::CascadeArgs A{};
      A.sum = c->d_sum; A.int_stride = int_stride; A.vnf = c->d_vnf; A.vnf64 = c->d_vnf64; A.grid_stride = grid_stride;
      const bool seg = c->semantics == 0 && equiv <= c->seg_max_batch && c->n_seg_items[k] > 0;
      A.levels = c->d_levels;
      A.bands = seg ? c->d_seg_items[k] : c->d_band_items[k];
      A.nbands = seg ? c->n_seg_items[k] : (int)P.band_items[k].size();
      A.nodes = c->d_nodes[k]; A.nodes_sparse = c->d_nodes_sparse[k]; A.stages = c->d_stages; A.kinds = c->d_kinds; A.nstages = (int)m.stages.size();
      const bool p16 = k == 0 && c->p16 && !seg && c->semantics == 0 && c->debug_max_stages == 0;
      if (p16) { A.nodes = c->d_nodes_p16; A.nodes_sparse = c->d_nodes_sparse_p16; A.kinds = c->d_kinds_p16; }
      if (c->debug_max_stages > 0) A.nstages = std::min(A.nstages, c->debug_max_stages);   // timing experiments only (HC_DEBUG_MAX_STAGES)
      A.trees = c->d_trees; A.leaves = c->d_leaves; A.g = p16 ? c->geom_p16 : c->geom[k]; A.tile_words = p16 ? c->tile_words_p16 : c->tile_words[k];
      A.sparse_max = c->sparse_max;
      A.frame_chunk = c->frame_chunk;
      A.row_stages = c->row_stages;
      A.cand = c->d_cand; A.cand_count = c->d_count;
      A.cand_cap = (unsigned int)c->cand_cap;
      if (A.nstages < (int)m.stages.size()) A.cand_cap = 0;   // truncated cascade (timing experiments): emit nothing
      A.taps = c->taps_enabled ? c->d_taps : nullptr;
      A.int_weights = c->int_weights ? 1 : 0;
      A.spec = p16 ? c->spec_id_p16 : c->spec_id;
      int compiled = 0;
      A.tree_pass = c->d_tree_pass; A.tree_fail = c->d_tree_fail; A.tree_order = c->d_tree_order;
      if (c->semantics == 1 && !c->cv_chain)
        e = hc::launch_cv24_tree(side && k == 1 ? c->side_stream : s, nframes, c->band_waves[k], c->lds_bytes[k], A);
      else if (c->semantics == 1)
        e = hc::launch_cascade_cv24(side && k == 1 ? c->side_stream : s, nframes, !m.is_stump(), c->band_rows[k], c->band_waves[k],
                                    c->lds_bytes[k], A);
      else
      e = hc::launch_cascade(side && k == 1 ? c->side_stream : s, nframes, !m.is_stump(), seg, c->band_rows[k],
                             c->band_waves[k], p16 ? c->lds_bytes_p16 : c->lds_bytes[k], A, &compiled);
      if (A.nbands > 0) c->last_compiled = launches++ == 0 ? compiled : std::min(c->last_compiled, compiled);
      if (e) return set_err(HC_ERR_HIP, std::string("k_cascade launch: ") + hipGetErrorString((hipError_t)e));
    }
    if (side) {
      HIPCHK(hipEventRecord(c->ev_join, c->side_stream));
      HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
    }
    if (timed) HIPCHK(hipEventRecord(c->ev[5], s));
  }
  // Results: the batch's own last kernels store the candidates and {count, sequence number} in page-locked host memory;
  // the host polls the sequence word.  No copy command, no wait on a stream or event of the runtime: those return when
  // the hardware queue they were mapped to has drained, and with several contexts alive that queue also carries another
  // context's pending work -- a software pipeline then runs one batch deep (see ev_copied).
  if (int e = hc::launch_publish(s, c->d_count, c->d_cand, (unsigned int)std::min<size_t>(c->cand_cap, 0xffffffffu), c->h_cand, c->h_pub,
                                 ++c->pub_seq))
    return set_err(HC_ERR_HIP, std::string("k_publish launch: ") + hipGetErrorString((hipError_t)e));
  HIPCHK(hipEventRecord(c->ev_done, s));
  if (c->leader) c->scan_in_flight = true;
  c->pending = nframes;
  return HC_OK;
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

const char* hc_last_error(void) { return g_err.c_str(); }
const char* hc_version(void) { return "hip_cascade 0.1 (gfx950)"; }
const char* hc_kernel_name(int i) { return (i >= 0 && i < HC_NKERNELS) ? kKernelNames[i] : ""; }

int hc_cascade_load(const char* path, hc_cascade** out) {
  if (!path || !out) return set_err(HC_ERR_ARG, "null argument");
  *out = nullptr;
  hc_cascade* c = new hc_cascade;
  std::string err;
  int rc = hc::load_model(path, c->model, err);
  if (rc != HC_OK) { delete c; return set_err(rc, err); }
  *out = c;
  return HC_OK;
}

void hc_cascade_free(hc_cascade* c) {
  if (!c) return;
  for (hc_context* ctx : c->sync_ctxs) hc_context_destroy(ctx);
  delete c;
}

int hc_cascade_info(const hc_cascade* c, int* win_w, int* win_h, int* nstages, int* is_tree) {
  if (!c) return set_err(HC_ERR_ARG, "null cascade");
  if (win_w) *win_w = c->model.win_w;
  if (win_h) *win_h = c->model.win_h;
  if (nstages) *nstages = (int)c->model.stages.size();
  if (is_tree) *is_tree = c->model.is_stump() ? 0 : 1;
  return HC_OK;
}

int hc_cascade_counts(const hc_cascade* c, int* ntrees, int* nnodes, int* nleaves) {
  if (!c) return set_err(HC_ERR_ARG, "null cascade");
  if (ntrees) *ntrees = (int)c->model.trees.size();
  if (nnodes) *nnodes = (int)c->model.nodes.size();
  if (nleaves) *nleaves = (int)c->model.leaves.size();
  return HC_OK;
}

int hc_cascade_compiled_stages(const hc_cascade* c, uint64_t* hash, int* nstages) {
  if (!c) return set_err(HC_ERR_ARG, "null cascade");
  const hc::Model& m = c->model;
  const unsigned long long h = model_hash(m);
  int n = 0;
  ensure_env_plugins();
  if (weights_are_small_integers(m)) (void)hc::spec_lookup(h, m.win_w, m.win_h, &n);
  if (hash) *hash = (uint64_t)h;
  if (nstages) *nstages = n;
  return HC_OK;
}

int hc_cascade_dump(const hc_cascade* c, float* stage_thr, int32_t* stage_ntrees, int32_t* tree_nnodes,
                    int32_t* node_rects, float* node_thr, int32_t* node_left, int32_t* node_right, float* leaves) {
  if (!c) return set_err(HC_ERR_ARG, "null cascade");
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

int hc_context_create(const hc_cascade* c, int device, void* stream, hc_context** out) {
  if (!c || !out) return set_err(HC_ERR_ARG, "null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return set_err(HC_ERR_NO_DEVICE, "no HIP device available (this backend has no CPU fallback)");
  if (device < 0 || device >= ndev) return set_err(HC_ERR_ARG, "device index out of range");
  HIPCHK(hipSetDevice(device));
  hc_context* ctx = new hc_context;
  ctx->casc = c;
  ctx->device = device;
  if (stream) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
  else {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return set_err(HC_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    ctx->own_stream = true;
  }
  for (auto& e : ctx->ev) {
    if (hipEventCreate(&e) != hipSuccess) { hc_context_destroy(ctx); return set_err(HC_ERR_HIP, "hipEventCreate failed"); }
  }
  // (the side stream of the small-batch route is made on first use: every stream alive takes a share of the hardware queues)
  // copy_stream (the skin filter's second round trip: box sums of the grouped rectangles while the next batch already runs)
  // is a HIGH-priority stream: the runtime keeps those on hardware queues of their own, so its few microseconds of work
  // neither queue behind the next batch's kernels nor wait for them to drain
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  if (const char* e_ = getenv("HC_COPY_PRIO")) { if (atoi(e_) == 0) prio_greatest = prio_least; }
  if (hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, prio_greatest) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_copied, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_prep, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_var, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming) != hipSuccess) {
    hc_context_destroy(ctx);
    return set_err(HC_ERR_HIP, "cannot create the result stream / event");
  }
  if (const char* e_ = getenv("HC_SPARSE_MAX")) ctx->sparse_max = std::min(256, std::max(0, atoi(e_)));   // tuning knobs (a wavefront owns at most 64 of them)
  if (const char* e_ = getenv("HC_HOST_THREADS")) ctx->host_threads = std::max(1, atoi(e_));
  if (const char* e_ = getenv("HC_DEBUG_MAX_STAGES")) ctx->debug_max_stages = std::max(0, atoi(e_));
  if (const char* e_ = getenv("HC_ROW_STAGES")) ctx->row_stages = std::max(1, atoi(e_));
  if (const char* e_ = getenv("HC_BAND_ROWS1")) ctx->band_rows[0] = atoi(e_);
  if (const char* e_ = getenv("HC_BAND_ROWS2")) ctx->band_rows[1] = atoi(e_);
  if (const char* e_ = getenv("HC_BAND_WAVES1")) ctx->band_waves[0] = atoi(e_) == 8 ? 8 : 4;
  if (const char* e_ = getenv("HC_BAND_WAVES2")) ctx->band_waves[1] = atoi(e_) == 8 ? 8 : 4;
  if (const char* e_ = getenv("HC_FUSED_PREP")) ctx->fused_prep = atoi(e_) != 0 ? 1 : 0;
  if (const char* e_ = getenv("HC_FRAME_CHUNK")) ctx->frame_chunk = std::max(0, atoi(e_)) & ~7;
  if (const char* e_ = getenv("HC_FUSE_VARIANCE")) { ctx->fuse_variance = atoi(e_) != 0; ctx->fuse_variance_forced = ctx->fuse_variance; }
  if (const char* e_ = getenv("HC_INT_WEIGHTS")) ctx->allow_int_weights = atoi(e_) != 0;
  if (const char* e_ = getenv("HC_SPEC")) ctx->allow_spec = atoi(e_) != 0;
  if (const char* e_ = getenv("HC_PAIR_FORM")) ctx->pair_form = atoi(e_) != 0;
  if (const char* e_ = getenv("HC_P16")) ctx->allow_p16 = atoi(e_) != 0;
  if (const char* e_ = getenv("HC_BANK_ORDER")) ctx->bank_order = atoi(e_) != 0;
  if (const char* e_ = getenv("HC_FUSED_MIN_BATCH")) ctx->fused_min_batch = atof(e_);
  if (const char* e_ = getenv("HC_SIDE_MAX_BATCH")) ctx->side_max_batch = std::max(0, atoi(e_));
  if (const char* e_ = getenv("HC_SEG_MAX_BATCH")) ctx->seg_max_batch = std::max(0, atoi(e_));
  if (const char* e_ = getenv("HC_SEG_TILES")) ctx->seg_tiles = std::min(255, std::max(1, atoi(e_)));
  for (int k = 0; k < 2; k++)
    if (ctx->band_rows[k] != 8 && ctx->band_rows[k] != 16 && ctx->band_rows[k] != 32) {
      delete ctx;
      return set_err(HC_ERR_ARG, "band rows must be 8, 16 or 32");
    }
  int rc = upload_model(ctx);
  if (rc != HC_OK) { hc_context_destroy(ctx); return rc; }
  *out = ctx;
  return HC_OK;
}

void hc_context_destroy(hc_context* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
  if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
  if (ctx->upload_stream) { (void)hipStreamSynchronize(ctx->upload_stream); (void)hipStreamDestroy(ctx->upload_stream); }
  if (ctx->ev_uploaded) (void)hipEventDestroy(ctx->ev_uploaded);
  if (ctx->ev_done) (void)hipEventDestroy(ctx->ev_done);
  if (ctx->ev_copied) (void)hipEventDestroy(ctx->ev_copied);
  if (ctx->ev_prep) (void)hipEventDestroy(ctx->ev_prep);
  if (ctx->ev_var) (void)hipEventDestroy(ctx->ev_var);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  ctx->free_workspace();
  free_model_tables(ctx);
  for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

}  // extern "C"

namespace {
// prep_scales / leader: null for a context on its own; for a group sharing one prep pass, the union ladder and (for
// followers) the context whose workspace holds the integrals
int configure_impl(hc_context* ctx, int rows, int cols, double scale_factor, int min_neighbors, int min_w, int min_h, int max_w,
                   int max_h, int max_batch, int cand_capacity, const std::vector<float>* prep_scales, hc_context* leader) {
  if (!ctx) return set_err(HC_ERR_ARG, "null context");
  if (!(scale_factor > 1.0)) return set_err(HC_ERR_ARG, "scaleFactor must be > 1");   // CV_Assert upstream
  if (rows <= 0 || cols <= 0 || max_batch <= 0 || max_batch > 65535) return set_err(HC_ERR_ARG, "bad geometry / batch");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->free_workspace();
  hc::Plan& P = ctx->plan;
  const bool cv24 = ctx->semantics == 1;
  if (cv24 && (prep_scales || leader)) return set_err(HC_ERR_UNSUPPORTED, "shared prep passes exist for the OpenCV >= 4 semantics only");
  if (cv24 && (ctx->band_rows[0] != 16 || ctx->band_rows[1] != 16)) return set_err(HC_ERR_UNSUPPORTED, "2.4 semantics: 16-row bands only");
  int rc = cv24 ? hc::build_plan_cv24(ctx->casc->model, rows, cols, scale_factor, min_w, min_h, max_w, max_h, ctx->band_rows[0],
                                      ctx->band_rows[1], P, ctx->cv)
                : hc::build_plan(ctx->casc->model, rows, cols, scale_factor, min_w, min_h, max_w, max_h, ctx->band_rows[0],
                                 ctx->band_rows[1], P, prep_scales);
  if (rc != HC_OK) return set_err(rc, "cannot plan this configuration");
  for (const hc::DevLevel& L : P.levels)
    if (L.nx > 65535 || L.ny_all > 65535 || P.levels.size() > 255) return set_err(HC_ERR_UNSUPPORTED, "frame too large");
  ctx->max_batch = max_batch;
  ctx->min_neighbors = min_neighbors;
  ctx->cand_per_frame = cand_capacity > 0 ? cand_capacity : 4096;
  const size_t B = (size_t)max_batch;
  if (leader) {
    if (leader->plan.int_words != P.int_words || leader->max_batch != max_batch || leader->device != ctx->device)
      return set_err(HC_ERR_STATE, "shared prep: integral layouts of leader and follower differ");
    ctx->d_sum = leader->d_sum;
    ctx->d_sq = leader->d_sq;
    ctx->leader = leader;
    leader->followers.push_back(ctx);
    ctx->seen_seq = leader->prep_seq;
    ctx->scan_in_flight = false;
    // Same window and the same grid as the leader's (alt, alt2 and alt_tree are all 20 x 20): the variance factors are the
    // same numbers, so the leader's k_variance serves this member too (HC_SHARE_VNF=0 gives every member its own).
    static const bool share_ok = [] { const char* e = getenv("HC_SHARE_VNF"); return !e || atoi(e) != 0; }();
    const hc::Model &lm = leader->casc->model, &m = ctx->casc->model;
    bool same = share_ok && !cv24 && leader->semantics == 0 && lm.win_w == m.win_w && lm.win_h == m.win_h &&
                leader->plan.grid_elems == P.grid_elems && leader->plan.levels.size() == P.levels.size();
    for (size_t i = 0; same && i < P.levels.size(); i++) {
      const hc::DevLevel &a = leader->plan.levels[i], &b = P.levels[i];
      same = a.nx == b.nx && a.ny == b.ny && a.ny_all == b.ny_all && a.ystep == b.ystep && a.grid_ofs == b.grid_ofs;
    }
    ctx->vnf_shared = same;
  } else {
    if ((rc = dev_alloc(&ctx->d_sum, B * P.int_words))) return rc;
    if ((rc = dev_alloc(&ctx->d_sq, B * P.int_words))) return rc;
    // row 0 / column 0 of every integral stay zero for the lifetime of the workspace
    HIPCHK(hipMemset(ctx->d_sum, 0, std::max<size_t>(B * P.int_words, 1) * 4));
    HIPCHK(hipMemset(ctx->d_sq, 0, std::max<size_t>(B * P.int_words, 1) * 4));
  }
  if (cv24) {
    if ((rc = dev_alloc(&ctx->d_vnf64, B * P.grid_elems))) return rc;
    HIPCHK(hipMemset(ctx->d_vnf64, 0, std::max<size_t>(B * P.grid_elems, 1) * 8));
    if ((rc = dev_upload(&ctx->d_cvlevels, ctx->cv.cl)) || (rc = dev_upload(&ctx->d_cvtab_ofs, ctx->cv.tab_ofs)) ||
        (rc = dev_upload(&ctx->d_cvtab_a0, ctx->cv.tab_a0)) || (rc = dev_upload(&ctx->d_cvtab_a1, ctx->cv.tab_a1)))
      return rc;
  }
  if (ctx->vnf_shared) ctx->d_vnf = leader->d_vnf;
  else if ((rc = dev_alloc(&ctx->d_vnf, cv24 ? 1 : B * P.grid_elems))) return rc;
  ctx->cand_cap = B * ctx->cand_per_frame;
  if ((rc = dev_alloc(&ctx->d_cand, ctx->cand_cap))) return rc;
  if ((rc = dev_alloc(&ctx->d_count, 1))) return rc;
  if (!cv24 && !ctx->vnf_shared) HIPCHK(hipMemset(ctx->d_vnf, 0, std::max<size_t>(B * P.grid_elems, 1) * 4));
  if ((rc = dev_upload(&ctx->d_levels, P.levels))) return rc;
  if ((rc = dev_upload(&ctx->d_xtab, P.xtab))) return rc;
  if ((rc = dev_upload(&ctx->d_ytab, P.ytab))) return rc;
  if ((rc = dev_upload(&ctx->d_resize_items, P.resize_items))) return rc;
  if ((rc = dev_upload(&ctx->d_row_items, P.row_items))) return rc;
  if ((rc = dev_upload(&ctx->d_col_items, P.col_items))) return rc;
  if ((rc = dev_upload(&ctx->d_var_items, P.var_items))) return rc;
  if ((rc = dev_upload(&ctx->d_band_items[0], P.band_items[0]))) return rc;
  if ((rc = dev_upload(&ctx->d_band_items[1], P.band_items[1]))) return rc;
  for (int k = 0; k < 2; k++) {   // the same bands cut into x-segments (k_cascade re-derives the x-skip state per segment)
    std::vector<hc::WorkItem> seg;
    for (const hc::WorkItem& it : P.band_items[k]) {
      const int ntiles = (P.levels[it.level].nx + hc::TILE_POS - 1) / hc::TILE_POS;
      for (int t0 = 0; t0 < ntiles; t0 += ctx->seg_tiles)
        seg.push_back(hc::WorkItem{it.level, it.a, (t0 * hc::TILE_POS) | (ctx->seg_tiles << 16)});
    }
    ctx->n_seg_items[k] = (int)seg.size();
    if ((rc = dev_upload(&ctx->d_seg_items[k], seg))) return rc;
  }
  g_alloc_count += 2;
  HIPCHK(hipHostMalloc((void**)&ctx->h_cand, ctx->cand_cap * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped));
  HIPCHK(hipHostMalloc((void**)&ctx->h_pub, 2 * sizeof(unsigned int), hipHostMallocCoherent | hipHostMallocMapped));
  ctx->h_pub[0] = ctx->h_pub[1] = 0;
  ctx->pub_seq = 0;
  ctx->pending = 0;
  ctx->last_frames = 0;
  ctx->configured = true;
  return HC_OK;
}
}  // namespace

extern "C" {

int hc_context_configure(hc_context* ctx, int rows, int cols, double scale_factor, int min_neighbors, int min_w,
                         int min_h, int max_w, int max_h, int max_batch, int cand_capacity) {
  return configure_impl(ctx, rows, cols, scale_factor, min_neighbors, min_w, min_h, max_w, max_h, max_batch, cand_capacity,
                        nullptr, nullptr);
}

int hc_context_configure_shared(hc_context* const* ctxs, int n, int rows, int cols, double scale_factor, int min_neighbors,
                                int min_w, int min_h, int max_w, int max_h, int max_batch, int cand_capacity) {
  if (!ctxs || n < 1) return set_err(HC_ERR_ARG, "hc_context_configure_shared: no contexts");
  for (int i = 0; i < n; i++) {
    if (!ctxs[i]) return set_err(HC_ERR_ARG, "null context");
    if (ctxs[i]->device != ctxs[0]->device) return set_err(HC_ERR_ARG, "shared prep: contexts must live on one device");
    for (int j = 0; j < i; j++)
      if (ctxs[i] == ctxs[j]) return set_err(HC_ERR_ARG, "shared prep: the same context twice");
  }
  // union of the ladders: one pyramid + integral pass serves every cascade (levels depend on the scale only)
  std::vector<float> prep;
  for (int i = 0; i < n; i++)
    for (float sc : hc::scale_ladder(ctxs[i]->casc->model, rows, cols, scale_factor, min_w, min_h, max_w, max_h))
      if (std::find(prep.begin(), prep.end(), sc) == prep.end()) prep.push_back(sc);
  std::sort(prep.begin(), prep.end());
  int rc = configure_impl(ctxs[0], rows, cols, scale_factor, min_neighbors, min_w, min_h, max_w, max_h, max_batch, cand_capacity,
                          &prep, nullptr);
  for (int i = 1; i < n && rc == HC_OK; i++)
    rc = configure_impl(ctxs[i], rows, cols, scale_factor, min_neighbors, min_w, min_h, max_w, max_h, max_batch, cand_capacity,
                        &prep, ctxs[0]);
  return rc;
}

int hc_detect_batch_enqueue(hc_context* ctx, const uint8_t* d_frames, size_t frame_stride_bytes, size_t row_step_bytes,
                            int nframes) {
  if (!ctx || !ctx->configured) return set_err(HC_ERR_STATE, "context not configured");
  if (!d_frames || nframes <= 0 || nframes > ctx->max_batch) return set_err(HC_ERR_ARG, "bad frame pointer / batch size");
  if (row_step_bytes < (size_t)ctx->plan.cols) return set_err(HC_ERR_ARG, "row step smaller than the row");
  if (ctx->pending) return set_err(HC_ERR_STATE, "previous batch not waited");
  HIPCHK(hipSetDevice(ctx->device));
  if (int rc = shared_enqueue_check(ctx, d_frames, nframes)) return rc;
  return enqueue_device(ctx, d_frames, frame_stride_bytes, row_step_bytes, nframes);
}

int hc_detect_batch_enqueue_host(hc_context* ctx, const uint8_t* h_frames, size_t frame_stride_bytes,
                                 size_t row_step_bytes, int nframes) {
  if (!ctx || !ctx->configured) return set_err(HC_ERR_STATE, "context not configured");
  if (!h_frames || nframes <= 0 || nframes > ctx->max_batch) return set_err(HC_ERR_ARG, "bad frame pointer / batch size");
  if (row_step_bytes < (size_t)ctx->plan.cols) return set_err(HC_ERR_ARG, "row step smaller than the row");
  if (ctx->pending) return set_err(HC_ERR_STATE, "previous batch not waited");
  HIPCHK(hipSetDevice(ctx->device));
  if (int rc = shared_enqueue_check(ctx, h_frames, nframes)) return rc;
  if (ctx->leader) return enqueue_device(ctx, h_frames, frame_stride_bytes, row_step_bytes, nframes);   // frames went up with the leader
  if (int rc = ensure_frames(ctx)) return rc;
  const size_t rows = ctx->plan.rows, cols = ctx->plan.cols;
  // d_frames is free: the context's previous batch has been waited for (checked above), so the copy may start at once
  hipStream_t up = ctx->stream;
  const bool separate = (size_t)nframes * rows * cols >= ((size_t)4 << 20);
  if (separate) {
    if (!ctx->upload_stream) {
      HIPCHK(hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&ctx->ev_uploaded, hipEventDisableTiming));
    }
    up = ctx->upload_stream;
  }
  if (row_step_bytes == cols && (nframes == 1 || frame_stride_bytes == rows * cols)) {
    HIPCHK(hipMemcpyAsync(ctx->d_frames, h_frames, (size_t)nframes * rows * cols, hipMemcpyHostToDevice, up));
  } else {
    for (int f = 0; f < nframes; f++)
      HIPCHK(hipMemcpy2DAsync(ctx->d_frames + (size_t)f * rows * cols, cols, h_frames + (size_t)f * frame_stride_bytes,
                              row_step_bytes, cols, rows, hipMemcpyHostToDevice, up));
  }
  if (separate) {
    HIPCHK(hipEventRecord(ctx->ev_uploaded, up));
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_uploaded, 0));
  }
  return enqueue_device(ctx, ctx->d_frames, rows * cols, cols, nframes);
}

void* hc_context_stream(const hc_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int hc_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return set_err(HC_ERR_ARG, "hc_host_alloc: bad arguments");
  *out = nullptr;
  HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return HC_OK;
}

void hc_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int hc_host_register(void* p, size_t bytes) {
  if (!p || bytes == 0) return set_err(HC_ERR_ARG, "hc_host_register: bad arguments");
  HIPCHK(hipHostRegister(p, bytes, hipHostRegisterDefault));
  return HC_OK;
}

int hc_host_unregister(void* p) {
  if (!p) return set_err(HC_ERR_ARG, "hc_host_unregister: null pointer");
  HIPCHK(hipHostUnregister(p));
  return HC_OK;
}

}  // extern "C"

namespace {
// Host wait for an event WITHOUT the runtime's blocking wait: hipEventSynchronize / hipStreamSynchronize on a stream whose
// hardware queue also carries another context's pending copy return only when THAT copy is done (one batch late).
int wait_event_polling(hipEvent_t ev) {
  for (unsigned spins = 0;; spins++) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return HC_OK;
    if (e != hipErrorNotReady) return set_err(HC_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
    if (spins > 64) std::this_thread::yield();
  }
}

// The batch's last kernel has stored sequence number pub_seq (k_publish_flag): poll for it.  Every millisecond or so the
// batch's completion event is asked for errors, so a faulted kernel ends the wait with an error instead of a hang.
int wait_published(hc_context* ctx) {
  const unsigned int want = ctx->pub_seq;
  const auto t0 = std::chrono::steady_clock::now();
  bool napping = false;
  unsigned int naps = 0;
  int done_polls = 0;
  for (unsigned int spins = 1;; spins++) {
    if (__atomic_load_n(&ctx->h_pub[0], __ATOMIC_ACQUIRE) == want) return HC_OK;
    if (!napping) {
      // The first half millisecond is spun through (a single frame is back in 0.2-0.4 ms); a batch that takes longer is
      // slept for in 50 us naps: a waiting thread must not eat a core of a CPU quota shared with seven other ranks.
      if (spins & 63u) { __builtin_ia32_pause(); continue; }
      if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 500.0) continue;
      napping = true;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((++naps & 63u) == 0) {
      const hipError_t e = hipEventQuery(ctx->ev_done);
      if (e != hipSuccess && e != hipErrorNotReady) return set_err(HC_ERR_HIP, std::string("batch failed: ") + hipGetErrorString(e));
      if (e == hipSuccess && ++done_polls > 200)
        return set_err(HC_ERR_HIP, "the batch completed but its result flag was never published");
    }
  }
}

// waits for the enqueued batch and groups its candidates: grouped[f] = boxes of frame f
int wait_core(hc_context* ctx, std::vector<std::vector<hc_rect>>& grouped) {
  HIPCHK(hipSetDevice(ctx->device));
  const int nframes = ctx->pending;
  ctx->pending = 0;
  unsigned int count = 0;
  for (int attempt = 0;; attempt++) {
    if (int rcw = wait_published(ctx)) return rcw;      // this batch's kernels are done, count and candidates are in host memory
    count = ctx->h_pub[1];
    if (ctx->debug_max_stages > 0 && count > ctx->cand_cap) count = 0;   // timing experiments only
    if (count <= ctx->cand_cap) break;
    // More raw candidates than the workspace holds (the counter keeps counting past the capacity, so the need is
    // known exactly).  OpenCV would simply return more rectangles: grow the buffers and run the batch again.
    if (attempt >= 2 || !ctx->inflight_src || (ctx->leader && ctx->leader->prep_seq != ctx->seen_seq))
      return set_err(HC_ERR_CAND_OVERFLOW, "raw candidate workspace overflow and the batch cannot be re-run "
                                           "(shared prep: the leader has already moved on); raise cand_capacity");
    const size_t want = (size_t)count + (size_t)count / 4 + 1024;
    // the frames of a follower are the leader's integrals, still in place (checked above)
    const bool own_frames = ctx->leader || ctx->inflight_src == ctx->d_frames || ctx->inflight_src == ctx->d_gray;
    const bool may_rerun = own_frames || ctx->rerun_caller_frames;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->d_cand); ctx->d_cand = nullptr;
    (void)hipHostFree(ctx->h_cand); ctx->h_cand = nullptr;
    ctx->cand_cap = 0;
    int rc = dev_alloc(&ctx->d_cand, want);
    if (rc) return set_err(HC_ERR_CAND_OVERFLOW, "raw candidate workspace overflow; growing it failed: " + g_err);
    g_alloc_count++;
    if (hipHostMalloc((void**)&ctx->h_cand, want * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess)
      return set_err(HC_ERR_CAND_OVERFLOW, "raw candidate workspace overflow; growing the host buffer failed");
    ctx->cand_cap = want;
    if (!may_rerun)
      return set_err(HC_ERR_CAND_OVERFLOW, "raw candidate workspace overflow: " + std::to_string(count) + " candidates; the workspace "
                                           "now holds " + std::to_string(want) + " -- enqueue the batch again (caller-owned device "
                                           "frames are only re-read by the library after hc_context_set_rerun_on_overflow(ctx, 1))");
    if ((rc = enqueue_device(ctx, ctx->inflight_src, ctx->inflight_frame_stride, ctx->inflight_row_step, nframes))) return rc;
    ctx->pending = 0;
  }
  if (ctx->timed_last && !ctx->plan.levels.empty())
    for (int i = 0; i < HC_NKERNELS; i++) HIPCHK(hipEventElapsedTime(&ctx->last_ms[i], ctx->ev[i], ctx->ev[i + 1]));
  ctx->last_frames = nframes;
  ctx->last_raw.clear();
  ctx->last_raw_ofs.assign(nframes + 1, 0);
  // canonical single-thread order: frame -> scale -> y -> x  == ascending key
  std::sort(ctx->h_cand, ctx->h_cand + count);
  const hc::Plan& P = ctx->plan;
  // frame boundaries inside the sorted key list
  std::vector<size_t> first(nframes + 1, count);
  {
    size_t i = 0;
    for (int f = 0; f < nframes; f++) {
      first[f] = i;
      while (i < count && (int)(ctx->h_cand[i] >> 40) == f) i++;
    }
    first[nframes] = i;
  }
  ctx->last_raw.resize(count);
  grouped.assign(nframes, {});
  auto do_frame = [&](int f) {
    std::vector<hc_rect>& grp = grouped[f];
    grp.reserve(first[f + 1] - first[f]);
    for (size_t i = first[f]; i < first[f + 1]; i++) {
      const unsigned long long k = ctx->h_cand[i];
      const hc::DevLevel& L = P.levels[(k >> 32) & 0xFF];
      const int gy = (int)((k >> 16) & 0xFFFF), gx = (int)(k & 0xFFFF);
      hc_rect r;
      if (ctx->semantics == 1) {
        // 2.4 (B.9): cvRound(x * factor) with the f64 factor of the scale loop
        const double factor = ctx->cv.lv[(k >> 32) & 0xFF].factor;
        r = hc_rect{(int)std::nearbyint(gx * L.ystep * factor), (int)std::nearbyint(gy * L.ystep * factor), L.win_w, L.win_h};
      } else {
        // Rect(cvRound(x*scale), cvRound(y*scale), winSize): int -> f32 times f32 scale
        r = hc_rect{(int)std::nearbyintf((float)(gx * L.ystep) * L.scale), (int)std::nearbyintf((float)(gy * L.ystep) * L.scale),
                    L.win_w, L.win_h};
      }
      grp.push_back(r);
      ctx->last_raw[i] = r;
    }
    if (ctx->semantics == 1) {
      if (ctx->min_neighbors != 0) hc::group_rectangles(grp, std::max(ctx->min_neighbors, 1), 0.2);   // 2.4: minNeighbors 0 = raw
    } else {
      hc::group_rectangles(grp, ctx->min_neighbors, 0.2);   // GROUP_EPS upstream
    }
  };
  if (nframes >= 4 && count >= 512 && ctx->host_threads > 1) {
    if (!ctx->pool) ctx->pool.reset(new WorkerPool(ctx->host_threads - 1));
    ctx->pool->parallel_for(nframes, do_frame);
  } else {
    for (int f = 0; f < nframes; f++) do_frame(f);
  }
  for (int f = 0; f < nframes; f++) ctx->last_raw_ofs[f + 1] = (int32_t)first[f + 1];
  return HC_OK;
}

int emit(const std::vector<std::vector<hc_rect>>& grouped, hc_rect* out, int out_capacity, int32_t* offsets) {
  int total = 0;
  const int nframes = (int)grouped.size();
  for (int f = 0; f < nframes; f++) {
    offsets[f] = total;
    if (total + (int)grouped[f].size() > out_capacity) return set_err(HC_ERR_CAPACITY, "output rectangle buffer too small");
    for (const hc_rect& r : grouped[f]) out[total++] = r;
  }
  offsets[nframes] = total;
  return HC_OK;
}
}  // namespace

extern "C" {

int hc_detect_batch_wait(hc_context* ctx, hc_rect* out, int out_capacity, int32_t* offsets) {
  if (!ctx || !ctx->configured) return set_err(HC_ERR_STATE, "context not configured");
  if (!ctx->pending) return set_err(HC_ERR_STATE, "no batch enqueued");
  if (ctx->faces_pending) return set_err(HC_ERR_STATE, "a detectFaces batch is pending: call hc_faces_batch_wait");
  if (!offsets || (!out && out_capacity > 0)) return set_err(HC_ERR_ARG, "null output");
  std::vector<std::vector<hc_rect>> grouped;
  int rc = wait_core(ctx, grouped);
  if (rc != HC_OK) return rc;
  return emit(grouped, out, out_capacity, offsets);
}

// ---- FaceDetector::detectFaces (reference src/tracker/detector.cc:10-43) ------------------------------
int hc_context_enable_faces(hc_context* ctx) {
  if (!ctx || !ctx->configured) return set_err(HC_ERR_STATE, "context not configured");
  if (ctx->faces_enabled) return HC_OK;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t B = (size_t)ctx->max_batch, rows = ctx->plan.rows, cols = ctx->plan.cols, npix = rows * cols;
  ctx->mask_pitch = (int)((cols + 1 + 15) / 16 * 16);
  ctx->mask_int_words = (size_t)ctx->mask_pitch * (rows + 1);
  ctx->box_cap = 256 * ctx->max_batch;
  int rc;
  if ((rc = dev_alloc(&ctx->d_bgr, B * npix * 3))) return rc;
  if ((rc = dev_alloc(&ctx->d_gray, B * npix))) return rc;
  if ((rc = dev_alloc(&ctx->d_mask, B * npix))) return rc;
  if ((rc = dev_alloc(&ctx->d_lut, B * 256))) return rc;
  if ((rc = dev_alloc(&ctx->d_hist, B * 256))) return rc;
  if ((rc = dev_alloc(&ctx->d_msum, B * ctx->mask_int_words))) return rc;
  if ((rc = dev_alloc(&ctx->d_boxsums, (size_t)ctx->box_cap))) return rc;
  if ((rc = dev_alloc(&ctx->d_boxes, (size_t)ctx->box_cap * 5))) return rc;
  HIPCHK(hipMemset(ctx->d_msum, 0, B * ctx->mask_int_words * 4));          // row 0 / column 0 stay zero
  // identity "pyramid level": the mask integral reuses k_pyramid_integral at scale 1
  hc::DevLevel L{};
  L.scale = 1.f; L.w = (int)cols; L.h = (int)rows; L.ystep = 1; L.int_pitch = ctx->mask_pitch;
  std::vector<hc::DevLevel> lv(1, L);
  std::vector<uint32_t> idx, idy;
  hc::build_resize_table((int)cols, (int)cols, idx);
  hc::build_resize_table((int)rows, (int)rows, idy);
  if ((rc = dev_upload(&ctx->d_idlevel, lv))) return rc;
  if ((rc = dev_upload(&ctx->d_idx, idx))) return rc;
  if ((rc = dev_upload(&ctx->d_idy, idy))) return rc;
  // work lists of the parallel row / column scan kernels for the same plane (small batches)
  std::vector<hc::WorkItem> ri, ci;
  for (int y0 = 0; y0 < (int)rows; y0 += 4) ri.push_back(hc::WorkItem{0, y0, 0});
  for (int x0 = 0; x0 < (int)cols; x0 += hc::WG_THREADS) ci.push_back(hc::WorkItem{0, x0, 0});
  ctx->n_mask_items[0] = (int)ri.size(); ctx->n_mask_items[1] = (int)ci.size();
  if ((rc = dev_upload(&ctx->d_mask_items[0], ri))) return rc;
  if ((rc = dev_upload(&ctx->d_mask_items[1], ci))) return rc;
  g_alloc_count += 2;
  HIPCHK(hipHostMalloc((void**)&ctx->h_boxes, (size_t)ctx->box_cap * 5 * sizeof(int), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&ctx->h_boxsums, (size_t)ctx->box_cap * sizeof(uint32_t), hipHostMallocDefault));
  ctx->faces_enabled = true;
  return HC_OK;
}

int hc_context_enable_scan_resize(hc_context* ctx, int src_rows, int src_cols) {
  if (!ctx || !ctx->configured || !ctx->faces_enabled) return set_err(HC_ERR_STATE, "call hc_context_enable_faces first");
  if (src_rows < ctx->plan.rows || src_cols < ctx->plan.cols)
    return set_err(HC_ERR_UNSUPPORTED, "INTER_AREA scan resize: the scan size must not exceed the source size in either axis");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  void* old[] = {ctx->d_src_bgr, ctx->d_area_xofs, ctx->d_area_xsi, ctx->d_area_xal, ctx->d_area_yofs, ctx->d_area_ysi, ctx->d_area_yal};
  for (void* p : old) if (p) (void)hipFree(p);
  ctx->d_src_bgr = nullptr;
  ctx->d_area_xofs = ctx->d_area_xsi = ctx->d_area_yofs = ctx->d_area_ysi = nullptr; ctx->d_area_xal = ctx->d_area_yal = nullptr;
  int rc = dev_alloc(&ctx->d_src_bgr, (size_t)ctx->max_batch * src_rows * src_cols * 3);
  if (rc) return rc;
  // cv::resize: the box kernel when both ratios are integers (resizeAreaFast_), tap tables otherwise (resizeArea_)
  ctx->area_general = (src_rows % ctx->plan.rows) != 0 || (src_cols % ctx->plan.cols) != 0;
  if (ctx->area_general) {
    std::vector<int> xo, xs, yo, ysv;
    std::vector<float> xa, ya;
    hc::build_area_table(src_cols, ctx->plan.cols, xo, xs, xa);
    hc::build_area_table(src_rows, ctx->plan.rows, yo, ysv, ya);
    if ((rc = dev_upload(&ctx->d_area_xofs, xo)) || (rc = dev_upload(&ctx->d_area_xsi, xs)) || (rc = dev_upload(&ctx->d_area_xal, xa)) ||
        (rc = dev_upload(&ctx->d_area_yofs, yo)) || (rc = dev_upload(&ctx->d_area_ysi, ysv)) || (rc = dev_upload(&ctx->d_area_yal, ya)))
      return rc;
  }
  ctx->src_rows = src_rows;
  ctx->src_cols = src_cols;
  return HC_OK;
}

static int faces_enqueue(hc_context* ctx, const uint8_t* d_bgr, size_t frame_stride, size_t row_step, int nframes,
                         const int lower[3], const int upper[3]) {
  if (ctx->src_rows) {   // cv::resize(orig, curr, scan.frame_size, 0, 0, INTER_AREA) first
    const size_t drow = (size_t)ctx->plan.cols * 3, dframe = drow * ctx->plan.rows;
    int e = ctx->area_general
                ? hc::launch_area_resize(ctx->stream, nframes, d_bgr, frame_stride, row_step, 3, ctx->d_area_xofs, ctx->d_area_xsi,
                                         ctx->d_area_xal, ctx->d_area_yofs, ctx->d_area_ysi, ctx->d_area_yal, ctx->d_bgr, dframe,
                                         ctx->plan.rows, ctx->plan.cols)
                : hc::launch_area_decimate(ctx->stream, nframes, d_bgr, frame_stride, row_step, 3, ctx->src_cols / ctx->plan.cols,
                                           ctx->src_rows / ctx->plan.rows, ctx->d_bgr, dframe, ctx->plan.rows, ctx->plan.cols);
    if (e) return set_err(HC_ERR_HIP, std::string("INTER_AREA launch: ") + hipGetErrorString((hipError_t)e));
    d_bgr = ctx->d_bgr; frame_stride = dframe; row_step = drow;
  }
  hc::SkinBounds sb{};
  for (int k = 0; k < 3; k++) { sb.lo[k] = lower[k]; sb.hi[k] = upper[k]; }
  const size_t npix = (size_t)ctx->plan.rows * ctx->plan.cols;
  int e = hc::launch_face_prep(ctx->stream, nframes, d_bgr, frame_stride, row_step, ctx->plan.rows, ctx->plan.cols, sb,
                               ctx->d_gray, ctx->d_mask, npix, ctx->d_hist, ctx->d_lut);
  if (e) return set_err(HC_ERR_HIP, std::string("face prep launch: ") + hipGetErrorString((hipError_t)e));
  // integral of the mask (sum only) for the skin-ratio post-filter
  const double equiv = (double)nframes * (double)npix / (1280.0 * 720.0);
  (void)equiv;
  if (use_fused_prep(ctx, nframes, false)) {
    e = hc::launch_pyramid_integral(ctx->stream, nframes, 1, ctx->plan.cols, ctx->d_mask, npix, ctx->plan.cols, ctx->plan.rows,
                                    ctx->plan.cols, ctx->d_idlevel, ctx->d_idx, ctx->d_idy, ctx->d_msum, nullptr,
                                    ctx->mask_int_words, nullptr, 0);
  } else {   // small batch: row scans, then column scans, each fully parallel (same policy as the pyramid prep)
    e = hc::launch_integral_rows(ctx->stream, nframes, ctx->n_mask_items[0], ctx->d_mask, npix, ctx->d_idlevel,
                                 ctx->d_mask_items[0], ctx->d_msum, nullptr, ctx->mask_int_words);
    if (!e) e = hc::launch_integral_cols(ctx->stream, nframes, ctx->n_mask_items[1], ctx->d_idlevel, ctx->d_mask_items[1],
                                         ctx->d_msum, nullptr, ctx->mask_int_words);
  }
  if (e) return set_err(HC_ERR_HIP, std::string("mask integral launch: ") + hipGetErrorString((hipError_t)e));
  int rc = enqueue_device(ctx, ctx->d_gray, npix, ctx->plan.cols, nframes);
  if (rc == HC_OK) ctx->faces_pending = true;
  return rc;
}

int hc_faces_batch_enqueue(hc_context* ctx, const uint8_t* d_bgr, size_t frame_stride_bytes, size_t row_step_bytes,
                           int nframes, const int lower[3], const int upper[3]) {
  if (!ctx || !ctx->configured || !ctx->faces_enabled) return set_err(HC_ERR_STATE, "call hc_context_enable_faces first");
  if (!d_bgr || !lower || !upper || nframes <= 0 || nframes > ctx->max_batch) return set_err(HC_ERR_ARG, "bad argument");
  if (row_step_bytes < (size_t)(ctx->src_cols ? ctx->src_cols : ctx->plan.cols) * 3) return set_err(HC_ERR_ARG, "row step smaller than the row");
  if (ctx->pending) return set_err(HC_ERR_STATE, "previous batch not waited");
  HIPCHK(hipSetDevice(ctx->device));
  return faces_enqueue(ctx, d_bgr, frame_stride_bytes, row_step_bytes, nframes, lower, upper);
}

int hc_faces_batch_enqueue_host(hc_context* ctx, const uint8_t* h_bgr, size_t frame_stride_bytes, size_t row_step_bytes,
                                int nframes, const int lower[3], const int upper[3]) {
  if (!ctx || !ctx->configured || !ctx->faces_enabled) return set_err(HC_ERR_STATE, "call hc_context_enable_faces first");
  if (!h_bgr || !lower || !upper || nframes <= 0 || nframes > ctx->max_batch) return set_err(HC_ERR_ARG, "bad argument");
  const size_t rows = ctx->src_rows ? ctx->src_rows : ctx->plan.rows;
  const size_t rowb = (size_t)(ctx->src_cols ? ctx->src_cols : ctx->plan.cols) * 3;
  if (row_step_bytes < rowb) return set_err(HC_ERR_ARG, "row step smaller than the row");
  if (ctx->pending) return set_err(HC_ERR_STATE, "previous batch not waited");
  HIPCHK(hipSetDevice(ctx->device));
  uint8_t* stage = ctx->src_rows ? ctx->d_src_bgr : ctx->d_bgr;
  for (int f = 0; f < nframes; f++)
    HIPCHK(hipMemcpy2DAsync(stage + (size_t)f * rows * rowb, rowb, h_bgr + (size_t)f * frame_stride_bytes,
                            row_step_bytes, rowb, rows, hipMemcpyHostToDevice, ctx->stream));
  return faces_enqueue(ctx, stage, rows * rowb, rowb, nframes, lower, upper);
}

static int faces_wait_core(hc_context* ctx, std::vector<std::vector<hc_rect>>& grouped) {
  if (!ctx->pending || !ctx->faces_pending) return set_err(HC_ERR_STATE, "no detectFaces batch enqueued");
  ctx->faces_pending = false;
  int rc = wait_core(ctx, grouped);
  if (rc != HC_OK) return rc;
  // skin post-filter (detector.cc:33-42): m = norm(mean(mask(roi))); keep iff !(m/256 < 0.8)
  int nb = 0;
  for (size_t f = 0; f < grouped.size(); f++)
    for (const hc_rect& r : grouped[f]) {
      if (nb >= ctx->box_cap) return set_err(HC_ERR_CAPACITY, "too many rectangles for the skin filter workspace");
      int* b = ctx->h_boxes + 5 * nb++;
      b[0] = (int)f; b[1] = r.x; b[2] = r.y; b[3] = r.width; b[4] = r.height;
    }
  if (nb) {
    HIPCHK(hipMemcpyAsync(ctx->d_boxes, ctx->h_boxes, (size_t)nb * 5 * sizeof(int), hipMemcpyHostToDevice, ctx->copy_stream));
    int e = hc::launch_box_mask_sums(ctx->copy_stream, ctx->d_msum, ctx->mask_int_words, ctx->mask_pitch, ctx->d_boxes, nb,
                                     ctx->d_boxsums);
    if (e) return set_err(HC_ERR_HIP, std::string("k_box_mask_sums launch: ") + hipGetErrorString((hipError_t)e));
    HIPCHK(hipMemcpyAsync(ctx->h_boxsums, ctx->d_boxsums, (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->copy_stream));
    HIPCHK(hipEventRecord(ctx->ev_copied, ctx->copy_stream));
    if (int rcw = wait_event_polling(ctx->ev_copied)) return rcw;
  }
  int k = 0;
  for (auto& g : grouped) {
    std::vector<hc_rect> keep;
    for (const hc_rect& r : g) {
      const double mean0 = (double)ctx->h_boxsums[k++] * (1. / ((double)r.width * r.height));   // cv::mean
      const double m = std::sqrt(mean0 * mean0);                                               // cv::norm(Scalar)
      if (m / 256 < 0.8) continue;
      keep.push_back(r);
    }
    g.swap(keep);
  }
  return HC_OK;
}

int hc_faces_batch_wait(hc_context* ctx, hc_rect* out, int out_capacity, int32_t* offsets) {
  if (!ctx || !ctx->configured) return set_err(HC_ERR_STATE, "context not configured");
  if (!offsets || (!out && out_capacity > 0)) return set_err(HC_ERR_ARG, "null output");
  std::vector<std::vector<hc_rect>> grouped;
  int rc = faces_wait_core(ctx, grouped);
  if (rc != HC_OK) return rc;
  return emit(grouped, out, out_capacity, offsets);
}

}  // extern "C"
namespace {
int sync_context(hc_cascade* c, int rows, int cols, double scale_factor, int min_neighbors, int min_w, int min_h, int max_w,
                 int max_h, hc_context** out, int semantics = 0);
}
extern "C" {

int hc_detect_faces(hc_cascade* c, const uint8_t* bgr, int rows, int cols, size_t step_bytes, const int lower[3],
                    const int upper[3], double scale_factor, hc_rect* out, int out_capacity, int* out_count) {
  if (!c || !bgr || !lower || !upper || !out_count) return set_err(HC_ERR_ARG, "null argument");
  if (!(scale_factor > 1.0)) return set_err(HC_ERR_ARG, "scaleFactor must be > 1");
  *out_count = 0;
  std::lock_guard<std::mutex> lock(c->mu);
  hc_context* ctx = nullptr;
  // the reference's literals: minNeighbors 2, minSize 30x30, maxSize unset (detector.cc:28-31)
  int rc = sync_context(c, rows, cols, scale_factor, 2, 30, 30, 0, 0, &ctx);
  if (rc) return rc;
  if ((rc = hc_context_enable_faces(ctx))) return rc;
  if ((rc = hc_faces_batch_enqueue_host(ctx, bgr, 0, step_bytes, 1, lower, upper))) return rc;
  std::vector<std::vector<hc_rect>> grouped;
  if ((rc = faces_wait_core(ctx, grouped))) return rc;
  return hc::deliver_sync_result(c, grouped[0], out, out_capacity, out_count);
}

int hc_batch_raw_candidates(hc_context* ctx, hc_rect* out, int out_capacity, int32_t* offsets) {
  if (!ctx || !offsets) return set_err(HC_ERR_ARG, "null argument");
  if ((int)ctx->last_raw.size() > out_capacity) return set_err(HC_ERR_CAPACITY, "output rectangle buffer too small");
  for (size_t i = 0; i < ctx->last_raw.size(); i++) out[i] = ctx->last_raw[i];
  for (size_t i = 0; i < ctx->last_raw_ofs.size(); i++) offsets[i] = ctx->last_raw_ofs[i];
  return HC_OK;
}

int hc_context_levels(const hc_context* ctx, hc_level* levels, int cap, int* nlevels) {
  if (!ctx || !ctx->configured || !nlevels) return set_err(HC_ERR_STATE, "context not configured");
  const auto& Ls = ctx->plan.levels;
  *nlevels = (int)Ls.size();
  if ((int)Ls.size() > cap) return set_err(HC_ERR_CAPACITY, "level buffer too small");
  for (size_t i = 0; i < Ls.size(); i++) levels[i] = to_public(Ls[i]);
  return HC_OK;
}

int hc_plan_query(const hc_cascade* c, int rows, int cols, double scale_factor, int min_w, int min_h, int max_w,
                  int max_h, hc_level* levels, int cap, int* nlevels, int64_t* algorithmic_bytes) {
  if (!c || !nlevels) return set_err(HC_ERR_ARG, "null argument");
  hc::Plan P;
  int rc = hc::build_plan(c->model, rows, cols, scale_factor, min_w, min_h, max_w, max_h, 16, 16, P);
  if (rc != HC_OK) return set_err(rc, "cannot plan this configuration");
  *nlevels = (int)P.levels.size();
  if (algorithmic_bytes) *algorithmic_bytes = P.algorithmic_bytes;
  if ((int)P.levels.size() > cap) return set_err(HC_ERR_CAPACITY, "level buffer too small");
  for (size_t i = 0; i < P.levels.size(); i++) levels[i] = to_public(P.levels[i]);
  return HC_OK;
}

int hc_group_rectangles(const hc_rect* in, int n, int group_threshold, double eps, hc_rect* out, int cap, int* nout) {
  if ((!in && n > 0) || !nout || n < 0) return set_err(HC_ERR_ARG, "null argument");
  std::vector<hc_rect> v(in, in + n);
  hc::group_rectangles(v, group_threshold, eps);
  if ((int)v.size() > cap) return set_err(HC_ERR_CAPACITY, "output rectangle buffer too small");
  for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
  *nout = (int)v.size();
  return HC_OK;
}

int64_t hc_context_algorithmic_bytes(const hc_context* ctx) {
  return (ctx && ctx->configured) ? ctx->plan.algorithmic_bytes : 0;
}

int hc_context_enable_stage_taps(hc_context* ctx, int enable) {
  if (!ctx) return set_err(HC_ERR_ARG, "null context");
  ctx->taps_enabled = enable != 0;
  return HC_OK;
}

int hc_context_read_tap(hc_context* ctx, int frame, int level, int what, void* dst, size_t dst_bytes) {
  if (!ctx || !ctx->configured || !dst) return set_err(HC_ERR_STATE, "context not configured");
  if (ctx->pending) return set_err(HC_ERR_STATE, "batch still in flight");
  const bool frame_tap = what == HC_TAP_FACE_MASK || what == HC_TAP_FACE_GRAY || what == HC_TAP_FACE_BGR;
  if (frame < 0 || frame >= ctx->last_frames || (!frame_tap && (level < 0 || level >= (int)ctx->plan.levels.size())))
    return set_err(HC_ERR_ARG, "frame / level out of range");
  HIPCHK(hipSetDevice(ctx->device));
  const hc::Plan& P = ctx->plan;
  const hc::DevLevel& L = P.levels[frame_tap ? 0 : level];
  switch (what) {
    case HC_TAP_PYRAMID: {
      const size_t n = (size_t)L.w * L.h;
      if (dst_bytes < n) return set_err(HC_ERR_CAPACITY, "tap buffer too small");
      if (!ctx->d_pyr) return set_err(HC_ERR_STATE, "no pyramid was materialised: enable stage taps before the enqueue");
      HIPCHK(hipMemcpy(dst, ctx->d_pyr + (size_t)frame * P.pyr_bytes + L.pyr_ofs, n, hipMemcpyDeviceToHost));
      return HC_OK;
    }
    case HC_TAP_SUM:
    case HC_TAP_SQSUM: {
      if (what == HC_TAP_SQSUM && !ctx->sq_valid)
        return set_err(HC_ERR_STATE, "the squares' integral was not materialised: enable stage taps before the enqueue");
      const size_t wbytes = (size_t)(L.w + 1) * 4;
      if (dst_bytes < wbytes * (L.h + 1)) return set_err(HC_ERR_CAPACITY, "tap buffer too small");
      const uint32_t* src = (what == HC_TAP_SUM ? ctx->d_sum : ctx->d_sq) + (size_t)frame * P.int_words + L.int_ofs;
      HIPCHK(hipMemcpy2D(dst, wbytes, src, (size_t)L.int_pitch * 4, wbytes, L.h + 1, hipMemcpyDeviceToHost));
      return HC_OK;
    }
    case HC_TAP_VNF: {
      const size_t n = (size_t)L.nx * L.ny * 4;
      if (dst_bytes < n) return set_err(HC_ERR_CAPACITY, "tap buffer too small");
      if (n) HIPCHK(hipMemcpy(dst, ctx->d_vnf + (size_t)frame * P.grid_elems + L.grid_ofs, n, hipMemcpyDeviceToHost));
      return HC_OK;
    }
    case HC_TAP_STAGES: {
      if (!ctx->taps_enabled || !ctx->d_taps) return set_err(HC_ERR_STATE, "stage taps were not enabled");
      const size_t n = (size_t)L.nx * L.ny_all;
      if (dst_bytes < n) return set_err(HC_ERR_CAPACITY, "tap buffer too small");
      if (n) HIPCHK(hipMemcpy(dst, ctx->d_taps + (size_t)frame * P.grid_elems + L.grid_ofs, n, hipMemcpyDeviceToHost));
      if (ctx->semantics == 1 && ctx->cv_chain) {
        // 2.4 semantics report "stages passed along the window's path, | 64 when accepted"; the tiled kernel records the
        // index of the failed stage of a chain (= stages passed) and nstages for an accepted window
        const int nst = (int)ctx->casc->model.stages.size();
        int8_t* t = (int8_t*)dst;
        for (size_t i = 0; i < n; i++)
          if (t[i] == (int8_t)nst) t[i] = (int8_t)(std::min(nst, 63) | 64);
      }
      return HC_OK;
    }
    case HC_TAP_FACE_BGR: {
      if (!ctx->faces_enabled) return set_err(HC_ERR_STATE, "detectFaces path not enabled");
      const size_t n = (size_t)P.rows * P.cols * 3;
      if (dst_bytes < n) return set_err(HC_ERR_CAPACITY, "tap buffer too small");
      HIPCHK(hipMemcpy(dst, ctx->d_bgr + (size_t)frame * n, n, hipMemcpyDeviceToHost));
      return HC_OK;
    }
    case HC_TAP_FACE_MASK:
    case HC_TAP_FACE_GRAY: {
      if (!ctx->faces_enabled) return set_err(HC_ERR_STATE, "detectFaces path not enabled");
      const size_t n = (size_t)P.rows * P.cols;
      if (dst_bytes < n) return set_err(HC_ERR_CAPACITY, "tap buffer too small");
      HIPCHK(hipMemcpy(dst, (what == HC_TAP_FACE_MASK ? ctx->d_mask : ctx->d_gray) + (size_t)frame * n, n, hipMemcpyDeviceToHost));
      return HC_OK;
    }
    default:
      return set_err(HC_ERR_ARG, "unknown tap");
  }
}

int hc_context_enable_timing(hc_context* ctx, int enable) {
  if (!ctx) return set_err(HC_ERR_ARG, "null context");
  ctx->timing_enabled = enable != 0;
  return HC_OK;
}

int hc_context_compiled_stages(hc_context* ctx, int* nstages) {
  if (!ctx || !nstages) return set_err(HC_ERR_ARG, "null argument");
  *nstages = ctx->last_compiled;
  return HC_OK;
}

int hc_context_kernel_ms(hc_context* ctx, float ms[HC_NKERNELS]) {
  if (!ctx || !ms) return set_err(HC_ERR_ARG, "null argument");
  if (!ctx->timed_last) return set_err(HC_ERR_STATE, "timing was not enabled for the last batch");
  for (int i = 0; i < HC_NKERNELS; i++) ms[i] = ctx->last_ms[i];
  return HC_OK;
}

// ---- synchronous single-frame call: the literal replacement of
//      cascade.detectMultiScale(gray, faces, scale, 2, 0|CASCADE_SCALE_IMAGE, Size(30,30))
}  // extern "C"

namespace {
// the per-cascade context of the synchronous calls, (re)configured for this geometry and these call parameters
int sync_context(hc_cascade* c, int rows, int cols, double scale_factor, int min_neighbors, int min_w, int min_h, int max_w,
                 int max_h, hc_context** out, int semantics) {
  int dev = 0, ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return set_err(HC_ERR_NO_DEVICE, "no HIP device available (this backend has no CPU fallback)");
  HIPCHK(hipGetDevice(&dev));
  c->last_result_valid = false;
  const int emw = (max_w == 0 || max_h == 0) ? cols : max_w, emh = (max_w == 0 || max_h == 0) ? rows : max_h;
  auto& lru = c->sync_ctxs;                    // most recently used first
  for (size_t i = 0; i < lru.size(); i++) {
    hc_context* x = lru[i];
    const hc::Plan& P = x->plan;
    if (x->device == dev && x->semantics == semantics && x->configured && P.rows == rows && P.cols == cols && P.scale_factor == scale_factor &&
        P.min_w == min_w && P.min_h == min_h && P.max_w == emw && P.max_h == emh && x->min_neighbors == min_neighbors) {
      std::rotate(lru.begin(), lru.begin() + i, lru.begin() + i + 1);
      *out = x;
      return HC_OK;
    }
  }
  // a combination not seen lately: a new context while the cache has room, else the least recently used one re-configured
  static const size_t cache_size = [] {
    const char* e = getenv("HC_SYNC_CONTEXTS");
    return (size_t)std::min(64, std::max(1, e ? atoi(e) : 4));
  }();
  hc_context* ctx = nullptr;
  int rc;
  if (lru.size() >= cache_size) {
    ctx = lru.back();
    lru.pop_back();
    if (ctx->device != dev) { hc_context_destroy(ctx); ctx = nullptr; }
  }
  if (!ctx && (rc = hc_context_create(c, dev, nullptr, &ctx))) return rc;
  if (ctx->semantics != semantics && (rc = hc_context_set_semantics(ctx, semantics))) { hc_context_destroy(ctx); return rc; }
  if ((rc = hc_context_configure(ctx, rows, cols, scale_factor, min_neighbors, min_w, min_h, max_w, max_h, 1, 1 << 16))) {
    hc_context_destroy(ctx);
    return rc;
  }
  lru.insert(lru.begin(), ctx);
  *out = ctx;
  return HC_OK;
}

// H2D copy of one interleaved 8-bit image into the context's colour staging buffer + cvtColor(.., BGR2GRAY) into the
// gray staging plane (what upstream's detectMultiScale does first with a multi-channel image)
int stage_color_as_gray(hc_context* ctx, const uint8_t* img, int rows, int cols, int channels, size_t step_bytes) {
  const size_t rowb = (size_t)cols * channels, need = rowb * rows;
  if (ctx->color_bytes < need) {
    if (ctx->d_color) (void)hipFree(ctx->d_color);
    ctx->d_color = nullptr; ctx->color_bytes = 0;
    int rc = dev_alloc(&ctx->d_color, need);
    if (rc) return rc;
    ctx->color_bytes = need;
  }
  if (int rc = ensure_frames(ctx)) return rc;
  HIPCHK(hipMemcpy2DAsync(ctx->d_color, rowb, img, step_bytes, rowb, rows, hipMemcpyHostToDevice, ctx->stream));
  int e = hc::launch_bgr2gray(ctx->stream, 1, ctx->d_color, need, rowb, channels, rows, cols, ctx->d_frames, (size_t)rows * cols);
  if (e) return set_err(HC_ERR_HIP, std::string("k_bgr2gray launch: ") + hipGetErrorString((hipError_t)e));
  return HC_OK;
}
}  // namespace

extern "C" {

int hc_detect_multiscale(hc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step_bytes, double scale_factor,
                         int min_neighbors, int flags, int min_w, int min_h, int max_w, int max_h, hc_rect* out,
                         int out_capacity, int* out_count) {
  return hc_detect_multiscale_image(c, gray, rows, cols, 1, step_bytes, scale_factor, min_neighbors, flags, min_w, min_h,
                                    max_w, max_h, out, out_capacity, out_count);
}

int hc_detect_multiscale_image(hc_cascade* c, const uint8_t* image, int rows, int cols, int channels, size_t step_bytes,
                               double scale_factor, int min_neighbors, int flags, int min_w, int min_h, int max_w, int max_h,
                               hc_rect* out, int out_capacity, int* out_count) {
  (void)flags;   // ignored by the OpenCV >= 4 evaluation path for this cascade format
  if (!c || !image || !out_count) return set_err(HC_ERR_ARG, "null argument");
  if (!(scale_factor > 1.0)) return set_err(HC_ERR_ARG, "scaleFactor must be > 1");
  if (channels != 1 && channels != 3 && channels != 4) return set_err(HC_ERR_ARG, "8-bit image with 1, 3 or 4 channels required");
  if (rows <= 0 || cols <= 0 || step_bytes < (size_t)cols * channels) return set_err(HC_ERR_ARG, "bad image geometry");
  *out_count = 0;
  std::lock_guard<std::mutex> lock(c->mu);
  hc_context* ctx = nullptr;
  int rc = sync_context(c, rows, cols, scale_factor, min_neighbors, min_w, min_h, max_w, max_h, &ctx);
  if (rc) return rc;
  if (channels == 1) {
    if ((rc = hc_detect_batch_enqueue_host(ctx, image, 0, step_bytes, 1))) return rc;
  } else {
    if (ctx->pending) return set_err(HC_ERR_STATE, "previous batch not waited");
    HIPCHK(hipSetDevice(ctx->device));
    if ((rc = stage_color_as_gray(ctx, image, rows, cols, channels, step_bytes))) return rc;
    if ((rc = enqueue_device(ctx, ctx->d_frames, (size_t)rows * cols, cols, 1))) return rc;
  }
  std::vector<std::vector<hc_rect>> grouped;
  if ((rc = wait_core(ctx, grouped))) return rc;
  return hc::deliver_sync_result(c, grouped[0], out, out_capacity, out_count);
}

int hc_detect_multiscale_reread(hc_cascade* c, hc_rect* out, int out_capacity, int* out_count) {
  if (!c || !out_count) return set_err(HC_ERR_ARG, "null argument");
  std::lock_guard<std::mutex> lock(c->mu);
  if (!c->last_result_valid) return set_err(HC_ERR_STATE, "the last synchronous call on this cascade did not end in HC_ERR_CAPACITY");
  *out_count = (int)c->last_result.size();
  if ((int)c->last_result.size() > out_capacity || !out) return set_err(HC_ERR_CAPACITY, "output rectangle buffer too small");
  for (size_t i = 0; i < c->last_result.size(); i++) out[i] = c->last_result[i];
  c->last_result_valid = false;
  c->last_result.clear();
  return HC_OK;
}

long long hc_debug_alloc_count(void) { return g_alloc_count.load(); }

int hc_spec_plugin_load(const char* path) {
  if (!path) return set_err(HC_ERR_ARG, "null path");
  return plugin_load(path);
}

int hc_context_set_rerun_on_overflow(hc_context* ctx, int allow) {
  if (!ctx) return set_err(HC_ERR_ARG, "null context");
  ctx->rerun_caller_frames = allow != 0;
  return HC_OK;
}

int hc_context_set_semantics(hc_context* ctx, int semantics) {
  if (!ctx) return set_err(HC_ERR_ARG, "null context");
  if (semantics != 0 && semantics != 1) return set_err(HC_ERR_ARG, "semantics: 0 (OpenCV >= 4) or 1 (OpenCV 2.4)");
  if (ctx->pending) return set_err(HC_ERR_STATE, "a batch is in flight");
  if (semantics == ctx->semantics) return HC_OK;
  const hc::Model& m = ctx->casc->model;
  if (semantics == 1) {
    if (m.has_tilted) return set_err(HC_ERR_UNSUPPORTED, "tilted features");
    for (const hc::ModelNode& n : m.nodes)
      for (int k = 0; k < 3; k++)
        for (int j = 0; j < 4; j++)
          if (n.rect[k][j] < 0) return set_err(HC_ERR_UNSUPPORTED, "rect coordinate out of range");
  }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->free_workspace();               // geometry, buffers and node tables all depend on the generation
  ctx->semantics = semantics;
  return upload_model(ctx);
}

// ---- legacy mode: the synchronous call with OpenCV 2.4.x semantics through a cached context
int hc_detect_multiscale_cv24_taps(hc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step_bytes, double scale_factor,
                                   int min_neighbors, int flags, int min_w, int min_h, int max_w, int max_h, hc_rect* out,
                                   int out_capacity, int* out_count, int8_t* taps, size_t taps_capacity) {
  if (!c || !gray || !out_count) return set_err(HC_ERR_ARG, "null argument");
  *out_count = 0;
  if (!(scale_factor > 1.0)) return set_err(HC_ERR_ARG, "scaleFactor must be > 1");
  if (rows <= 0 || cols <= 0 || step_bytes < (size_t)cols) return set_err(HC_ERR_ARG, "bad image geometry");
  if (!(flags & 2))   // CV_HAAR_SCALE_IMAGE: without it 2.4 scales the FEATURES instead of the image (not restated)
    return set_err(HC_ERR_UNSUPPORTED, "legacy mode is the CASCADE_SCALE_IMAGE branch (the flag the reference passes)");
  if (cols > 65535 || rows > 65535) return set_err(HC_ERR_UNSUPPORTED, "frame too large");
  std::lock_guard<std::mutex> lock(c->mu);
  hc_context* ctx = nullptr;
  int rc = sync_context(c, rows, cols, scale_factor, min_neighbors, min_w, min_h, max_w, max_h, &ctx, 1);
  if (rc) return rc;
  const hc::Plan& P = ctx->plan;
  if (taps) {
    size_t need = 0;
    for (const hc::DevLevel& L : P.levels) need += (size_t)L.nx * L.ny;
    if (need > taps_capacity) return set_err(HC_ERR_CAPACITY, "tap buffer too small");
  }
  ctx->taps_enabled = taps != nullptr;
  if ((rc = hc_detect_batch_enqueue_host(ctx, gray, 0, step_bytes, 1))) { ctx->taps_enabled = false; return rc; }
  std::vector<std::vector<hc_rect>> grouped;
  rc = wait_core(ctx, grouped);
  if (rc == HC_OK && taps) {
    size_t at = 0;
    for (size_t li = 0; li < P.levels.size() && rc == HC_OK; li++) {
      const size_t n = (size_t)P.levels[li].nx * P.levels[li].ny;
      if (n) rc = hc_context_read_tap(ctx, 0, (int)li, HC_TAP_STAGES, taps + at, n);
      at += n;
    }
  }
  ctx->taps_enabled = false;
  if (rc) return rc;
  return hc::deliver_sync_result(c, grouped[0], out, out_capacity, out_count);
}

int hc_detect_multiscale_cv24(hc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step_bytes, double scale_factor,
                              int min_neighbors, int flags, int min_w, int min_h, int max_w, int max_h, hc_rect* out,
                              int out_capacity, int* out_count) {
  return hc_detect_multiscale_cv24_taps(c, gray, rows, cols, step_bytes, scale_factor, min_neighbors, flags, min_w, min_h, max_w,
                                        max_h, out, out_capacity, out_count, nullptr, 0);
}

int hc_convert_bgr2gray(const uint8_t* image, int rows, int cols, int channels, size_t step_bytes, uint8_t* gray,
                        size_t gray_step_bytes) {
  if (!image || !gray) return set_err(HC_ERR_ARG, "null argument");
  if ((channels != 3 && channels != 4) || rows <= 0 || cols <= 0 || step_bytes < (size_t)cols * channels ||
      gray_step_bytes < (size_t)cols)
    return set_err(HC_ERR_ARG, "8-bit image with 3 or 4 channels required");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return set_err(HC_ERR_NO_DEVICE, "no HIP device available (this backend has no CPU fallback)");
  const size_t rowb = (size_t)cols * channels;
  uint8_t *d_src = nullptr, *d_dst = nullptr;
  int rc = dev_alloc(&d_src, rowb * rows);
  if (!rc) rc = dev_alloc(&d_dst, (size_t)rows * cols);
  hipError_t e = hipSuccess;
  if (!rc) {
    e = hipMemcpy2D(d_src, rowb, image, step_bytes, rowb, rows, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = (hipError_t)hc::launch_bgr2gray(nullptr, 1, d_src, rowb * rows, rowb, channels, rows, cols, d_dst, (size_t)rows * cols);
    if (e == hipSuccess) e = hipMemcpy2D(gray, gray_step_bytes, d_dst, cols, cols, rows, hipMemcpyDeviceToHost);
  }
  if (d_src) (void)hipFree(d_src);
  if (d_dst) (void)hipFree(d_dst);
  if (rc) return rc;
  if (e != hipSuccess) return set_err(HC_ERR_HIP, std::string("hc_convert_bgr2gray: ") + hipGetErrorString(e));
  return HC_OK;
}

// ---- box files: frame id -> rect list, the hand-over format between a batched / multi-GPU detection
// run and the sequential tracker.  Layout in include/hip_cascade.h; identical to ugproj-detector_amd/shard.py.
int hc_boxes_save(const char* path, int nframes, const int32_t* frame_ids, const int32_t* offsets, const hc_rect* rects) {
  if (!path || nframes < 0 || (nframes > 0 && (!frame_ids || !offsets)) || (nframes > 0 && offsets[nframes] > 0 && !rects))
    return set_err(HC_ERR_ARG, "hc_boxes_save: null argument");
  std::vector<int> order(nframes);
  for (int i = 0; i < nframes; i++) {
    order[i] = i;
    if (offsets[i + 1] < offsets[i]) return set_err(HC_ERR_ARG, "hc_boxes_save: offsets must not decrease");
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return frame_ids[a] < frame_ids[b]; });   // ascending id
  for (int i = 1; i < nframes; i++)
    if (frame_ids[order[i]] == frame_ids[order[i - 1]]) return set_err(HC_ERR_ARG, "hc_boxes_save: duplicate frame id");
  FILE* f = fopen(path, "wb");
  if (!f) return set_err(HC_ERR_IO, std::string("cannot create ") + path);
  const int32_t hdr[2] = {1, nframes};
  bool ok = fwrite("HCBX", 1, 4, f) == 4 && fwrite(hdr, 4, 2, f) == 2;
  for (int k = 0; ok && k < nframes; k++) {
    const int i = order[k];
    const int32_t rec[2] = {frame_ids[i], offsets[i + 1] - offsets[i]};
    ok = fwrite(rec, 4, 2, f) == 2;
    if (ok && rec[1] > 0) ok = fwrite(rects + offsets[i], sizeof(hc_rect), (size_t)rec[1], f) == (size_t)rec[1];
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? HC_OK : set_err(HC_ERR_IO, std::string("short write to ") + path);
}

int hc_boxes_load(const char* path, int32_t* frame_ids, int32_t* offsets, int frame_capacity, hc_rect* rects, int rect_capacity,
                  int* nframes, int* nrects) {
  if (!path || !nframes || !nrects) return set_err(HC_ERR_ARG, "hc_boxes_load: null argument");
  *nframes = *nrects = 0;
  FILE* f = fopen(path, "rb");
  if (!f) return set_err(HC_ERR_IO, std::string("cannot open ") + path);
  std::string buf;
  char tmp[1 << 16];
  size_t n;
  while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, n);
  fclose(f);
  if (buf.size() < 12 || memcmp(buf.data(), "HCBX", 4) != 0) return set_err(HC_ERR_FORMAT, "not an HCBX file");
  int32_t hdr[2];
  memcpy(hdr, buf.data() + 4, 8);
  if (hdr[0] != 1 || hdr[1] < 0) return set_err(HC_ERR_FORMAT, "unsupported HCBX version / frame count");
  // first pass: validate and count, so that a too-small buffer learns the sizes it needs
  size_t p = 12;
  long long total = 0;
  for (int i = 0; i < hdr[1]; i++) {
    if (p + 8 > buf.size()) return set_err(HC_ERR_FORMAT, "truncated HCBX file");
    int32_t rec[2];
    memcpy(rec, buf.data() + p, 8);
    p += 8;
    if (rec[1] < 0 || p + (size_t)rec[1] * 16 > buf.size()) return set_err(HC_ERR_FORMAT, "truncated HCBX file");
    p += (size_t)rec[1] * 16;
    total += rec[1];
  }
  *nframes = hdr[1];
  *nrects = (int)total;
  if (!frame_ids && !offsets && !rects) return HC_OK;                       // size query
  if (hdr[1] > frame_capacity || total > rect_capacity || !frame_ids || !offsets || (total > 0 && !rects))
    return set_err(HC_ERR_CAPACITY, "hc_boxes_load: output buffers too small (sizes returned)");
  p = 12;
  int at = 0;
  for (int i = 0; i < hdr[1]; i++) {
    int32_t rec[2];
    memcpy(rec, buf.data() + p, 8);
    p += 8;
    frame_ids[i] = rec[0];
    offsets[i] = at;
    if (rec[1] > 0) memcpy(rects + at, buf.data() + p, (size_t)rec[1] * 16);
    p += (size_t)rec[1] * 16;
    at += rec[1];
  }
  offsets[hdr[1]] = at;
  return HC_OK;
}

}  // extern "C"
