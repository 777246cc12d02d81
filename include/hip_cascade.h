/*
 * hip_cascade.h -- C ABI of the MI355X-native Haar-cascade detection backend.
 *
 * Drop-in boundary.  The reference reaches the hot path through exactly
 * two OpenCV member calls on one cv::CascadeClassifier object:
 *
 *   cascade_.load(cfg.detection.cascade_filepath.native())        reference src/file_io.cc:11
 *   cascade.detectMultiScale(gray, facesInGray, cfg.scale, 2,
 *                            0|CASCADE_SCALE_IMAGE, Size(30,30))   reference src/tracker/detector.cc:25-31
 *
 * Each entry point below names the reference interface it replaces.  The C++ shim
 * ugproj-detector_amd/hip_cascade/cascade_classifier.hpp wraps this ABI behind the
 * cv::CascadeClassifier spelling so the tracker's call-site is the only line that
 * changes.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  Never throws or aborts
 * across the ABI: every function returns an hc_status (0 = OK, the reference's own
 * convention at src/tracker/main.cc:14-41 and src/file_io.cc:7-15).
 * There is NO CPU fallback: without a HIP device every compute entry point fails
 * with HC_ERR_NO_DEVICE.
 */
#ifndef HIP_CASCADE_H
#define HIP_CASCADE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HC_API __attribute__((visibility("default")))

typedef enum hc_status {
  HC_OK = 0,
  HC_ERR_IO = -1,            /* file missing / unreadable                          */
  HC_ERR_FORMAT = -2,        /* not an opencv-haar-classifier XML / HCC1 container */
  HC_ERR_ARG = -3,           /* bad argument (scaleFactor <= 1, null pointer, ...)  */
  HC_ERR_UNSUPPORTED = -4,   /* tilted features, malformed trees                   */
  HC_ERR_NO_DEVICE = -5,     /* no HIP device / HIP runtime error at init          */
  HC_ERR_HIP = -6,           /* HIP runtime error (hc_last_error() has the text)   */
  HC_ERR_CAPACITY = -7,      /* caller's output buffer too small                   */
  HC_ERR_CAND_OVERFLOW = -8, /* raw-candidate workspace overflow that could not be grown (normally it grows and the batch re-runs) */
  HC_ERR_STATE = -9          /* call order violated (e.g. wait without enqueue)    */
} hc_status;

/* cv::Rect layout: x, y, width, height (reference uses std::vector<cv::Rect>,
 * src/tracker/detector.cc:24). */
typedef struct hc_rect { int32_t x, y, width, height; } hc_rect;

typedef struct hc_cascade hc_cascade;   /* immutable model; replaces cv::CascadeClassifier's loaded state */
typedef struct hc_context hc_context;   /* device workspace + stream; single caller at a time          */

/* ---- model: replaces cv::CascadeClassifier::load (reference src/file_io.cc:11) -------- */
/* Accepts the old-format XML the reference ships (data/haarcascade_frontalface_*.xml,
 * root type_id="opencv-haar-classifier") and the packed HCC1 container
 * (tools/pack_cascade.py).  Pure host code, needs no device. */
HC_API int  hc_cascade_load(const char* path, hc_cascade** out);
HC_API void hc_cascade_free(hc_cascade* c);
/* cv::CascadeClassifier::empty / getOriginalWindowSize analogues. */
HC_API int  hc_cascade_info(const hc_cascade* c, int* win_w, int* win_h, int* nstages, int* is_tree);
/* Flat model dump for loader-parity tests (same layout as oracle's orc_cascade_dump). */
HC_API int  hc_cascade_counts(const hc_cascade* c, int* ntrees, int* nnodes, int* nleaves);
HC_API int  hc_cascade_dump(const hc_cascade* c, float* stage_thr, int32_t* stage_ntrees,
                            int32_t* tree_nnodes, int32_t* node_rects, float* node_thr,
                            int32_t* node_left, int32_t* node_right, float* leaves);

/* Which cascade this is, and whether the library carries compiled early stages for it: *hash = FNV-1a over window,
 * stage thresholds, tree sizes, rects, weights, node thresholds, child links and leaves (the fields that decide every result); *nstages = number
 * of leading stages for which the library holds straight-line device code generated from exactly this cascade
 * (csrc/spec/, tools/gen_spec.py; all four stock cascades), 0 = the generic kernel evaluates every
 * stage.  Results are identical either way (tests/test_gpu_parity.py, HC_SPEC=0 switches the compiled stages off).
 * No counterpart in the reference: cv::CascadeClassifier has one evaluation path.  Pure host code. */
HC_API int  hc_cascade_compiled_stages(const hc_cascade* c, uint64_t* hash, int* nstages);
/* Compiled early stages for a cascade the library was not built with (a retrained haarcascade_frontalface_alt2, say --
 * the file the shipped configuration selects, reference conf/apink.mrchu.720p.cfg:10):
 *   tools/build_spec_plugin.sh <cascade.xml|.hcc> <out.so> [stages]
 * generates the stage code (tools/gen_spec.py) and compiles it, with the very same kernel text, into a small shared
 * object; this call (or HC_SPEC_PLUGINS=a.so:b.so in the environment) registers it for the rest of the process.
 * Contexts created afterwards for a cascade with that exact hash run the compiled stages; every other cascade is
 * unaffected.  HC_ERR_UNSUPPORTED when the object was built from another version of the kernel sources. */
HC_API int  hc_spec_plugin_load(const char* path);

/* ---- the call: replaces cv::CascadeClassifier::detectMultiScale
 *      (reference src/tracker/detector.cc:25-31) ------------------------------------------
 * Synchronous, host image in, host rectangles out (the reference's calling convention:
 * one frame at a time from one thread, src/tracker/face_tracker.cc:93-152).
 * `flags` is accepted and ignored exactly as OpenCV >= 4 ignores it for this path.
 * max_w/max_h == 0 means "image size" (cv::Size()).  `out` is overwritten.
 * Uses internal per-cascade contexts on the current HIP device, one per (frame geometry, call parameters) and a few
 * of them cached (HC_SYNC_CONTEXTS, default 4, least recently used one re-configured): a caller alternating between
 * a few image sizes allocates nothing after the first call of each.
 * When the rectangles do not fit `out`: HC_ERR_CAPACITY, *out_count = the number found, and
 * hc_detect_multiscale_reread hands them over without running the detection again (all synchronous calls on a
 * cascade share this: hc_detect_multiscale[_image], hc_detect_multiscale_cv24[_taps], hc_detect_faces). */
HC_API int hc_detect_multiscale(hc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step_bytes,
                                double scale_factor, int min_neighbors, int flags,
                                int min_w, int min_h, int max_w, int max_h,
                                hc_rect* out, int out_capacity, int* out_count);

/* The same call on an image with `channels` interleaved 8-bit channels: 1 (gray), or 3 / 4 (BGR / BGRA), which
 * upstream's detectMultiScale first converts with cvtColor(image, gray, COLOR_BGR2GRAY) (SURVEY 8b "Semantics").
 * The reference passes one channel (src/tracker/detector.cc:20-22); a caller that hands the colour frame
 * straight to the classifier gets upstream's behaviour instead of an error. */
HC_API int hc_detect_multiscale_image(hc_cascade* c, const uint8_t* image, int rows, int cols, int channels,
                                      size_t step_bytes, double scale_factor, int min_neighbors, int flags,
                                      int min_w, int min_h, int max_w, int max_h,
                                      hc_rect* out, int out_capacity, int* out_count);
HC_API int hc_detect_multiscale_reread(hc_cascade* c, hc_rect* out, int out_capacity, int* out_count);
/* cvtColor(image, gray, COLOR_BGR2GRAY) alone (reference src/tracker/detector.cc:20), host in / host out, on the
 * current device: (3735 B + 19235 G + 9798 R + 16384) >> 15.  channels = 3 or 4. */
HC_API int hc_convert_bgr2gray(const uint8_t* image, int rows, int cols, int channels, size_t step_bytes,
                               uint8_t* gray, size_t gray_step_bytes);

/* ---- legacy mode: OpenCV 2.4.x semantics of the same call ("T2") -------------------------
 * The reference was written against OpenCV 2.4 (src/tracker/associator.cc:3 includes opencv2/nonfree, removed in 3.0) and
 * passes CASCADE_SCALE_IMAGE (src/tracker/detector.cc:30), which only that generation honours.  A 2.4 build and a >= 4
 * build return different rectangles for the same frame; everything else in this header follows >= 4, and this
 * entry point is the explicit switch to the older behaviour: cvHaarDetectObjectsForROC, scale-image branch, scalar C
 * path -- f64 scale ladder, cvResize(CV_INTER_LINEAR), f64 variance without the low-variance gate, no x-skip, the last
 * row / column of window origins not scanned, area-normalised rect weights, stage bias 0.0001f, and the cascade file's
 * STAGE TREE (parent / next; data/haarcascade_frontalface_alt_tree.xml:1303-1304) instead of a linear chain.
 * `flags` must contain CASCADE_SCALE_IMAGE (2), as the reference's call does; minNeighbors 0 returns the raw candidates.
 * Synchronous, host in / host out, through a cached internal context like hc_detect_multiscale (the batched form:
 * hc_context_set_semantics).  Parity unpinned, like the rest.
 * _taps: also returns, per scanned position (levels of hc_plan_query_cv24 concatenated, [gy][gx]), the number of stages
 * passed along the window's path through the stage tree, | 64 when the window was accepted. */
HC_API int hc_detect_multiscale_cv24(hc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step_bytes,
                                     double scale_factor, int min_neighbors, int flags,
                                     int min_w, int min_h, int max_w, int max_h,
                                     hc_rect* out, int out_capacity, int* out_count);
HC_API int hc_detect_multiscale_cv24_taps(hc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step_bytes,
                                          double scale_factor, int min_neighbors, int flags,
                                          int min_w, int min_h, int max_w, int max_h,
                                          hc_rect* out, int out_capacity, int* out_count,
                                          int8_t* taps, size_t taps_capacity);
/* levels of that call (pure host): factors[i] (f64 scale) and geometry[7*i ..] = w, h, ystep, nx, ny, win_w, win_h */
HC_API int hc_plan_query_cv24(const hc_cascade* c, int rows, int cols, double scale_factor, int min_w, int min_h,
                              int max_w, int max_h, double* factors, int32_t* geometry, int cap, int* nlevels);
/* cvResize(src, dst, CV_INTER_LINEAR) of 2.4 for 8UC1 on the device (test tap of the legacy mode's first step) */
HC_API int hc_resize_linear_cv24(const uint8_t* src, int rows, int cols, size_t step_bytes,
                                 uint8_t* dst, int drows, int dcols);

/* ---- page-locked host memory for the frame pool of the throughput path ------------------ */
HC_API int  hc_host_alloc(size_t bytes, void** out);       /* hipHostMalloc; free with hc_host_free */
HC_API void hc_host_free(void* p);
HC_API int  hc_host_register(void* p, size_t bytes);       /* pin caller-owned memory in place */
HC_API int  hc_host_unregister(void* p);

/* ---- throughput path: batched, device-resident, asynchronous -------------------------- */
/* stream: a hipStream_t to enqueue on, or NULL for a stream owned by the context. */
HC_API int  hc_context_create(const hc_cascade* c, int device, void* stream, hc_context** out);
HC_API void hc_context_destroy(hc_context* ctx);
/* The hipStream_t the context enqueues on (its own, or the one it was created with).  A software pipeline of two or
 * more contexts is fastest with ALL of them on one stream -- hc_context_create(c, dev, hc_context_stream(first), &next):
 * kernels of consecutive batches then run back to back instead of side by side (two cascade launches sharing the CUs
 * cost 9 % at 720p), a large host batch (>= 4 MiB) still travels on an upload stream of the context's own while the
 * previous batch computes, and results come back through host memory, ordered after their own batch only.  The stream
 * of a context must outlive every context created on it. */
HC_API void* hc_context_stream(const hc_context* ctx);
/* Which generation of the call the context evaluates; call before hc_context_configure (it drops the workspace).
 * 0 (default): OpenCV >= 4, everything in this section as described.
 * 1: OpenCV 2.4.x (Appendix B; see hc_detect_multiscale_cv24 above) as a batched, device-resident path: its own scale
 *    loop and cvResize pyramid, f64 variance factors, then the same LDS-tiled band kernel with 2.4's node arithmetic
 *    for cascades whose stages form a chain (alt, alt2, default) and a preorder sweep of the stage tree on the same
 *    tiles for alt_tree.  Host or device frames, stage taps (HC_TAP_STAGES: stages passed, | 64 when accepted), raw candidates
 *    and grouping (minNeighbors 0 = raw candidates) as 2.4 defines them.  Not available: shared prep passes, the
 *    detectFaces path, pyramid / integral taps of the >= 4 layout keep their meaning (HC_TAP_VNF is empty). */
HC_API int  hc_context_set_semantics(hc_context* ctx, int semantics);
/* Fix geometry + call parameters and allocate the persistent workspace (no hipMalloc
 * happens on the detect path afterwards).  cand_capacity = raw candidates per frame
 * the workspace can hold (0 -> default 4096). */
HC_API int  hc_context_configure(hc_context* ctx, int rows, int cols, double scale_factor,
                                 int min_neighbors, int min_w, int min_h, int max_w, int max_h,
                                 int max_batch, int cand_capacity);
/* Several cascades over the SAME frames in one pass (BASELINE.json configs #3: alt2 + default, #5: all four):
 * ctxs[0..n) are contexts of different cascades on one device.  Pyramid level geometry depends on the scale
 * alone, so one pyramid + integral pass over the union of the cascades' scale ladders serves them all: ctxs[0]
 * (the leader) owns those integrals, the other contexts (followers) only run their own variance and cascade
 * kernels on them.  Per batch: enqueue the leader first, then each follower once with the same batch (a
 * follower takes the frames from the leader's pass; its frame pointer argument names the batch and must equal the
 * one the leader got -- HC_ERR_STATE otherwise, e.g. when the leader was enqueued again before this follower had
 * scanned the previous batch), then wait on each context for its cascade's rectangles.  Cross-stream ordering (followers behind the leader's prep kernels, the
 * leader's next batch behind the followers' scans) is handled with events, so the contexts may own their
 * streams or share one.  Reconfiguring or destroying the leader leaves the followers unconfigured.
 * hc_context_algorithmic_bytes stays each cascade's own full figure (sharing is an optimisation, not a
 * discount); hc_context_levels of any member lists the union ladder, levels outside that cascade's own ladder
 * with an empty grid (nx = ny = 0). */
HC_API int  hc_context_configure_shared(hc_context* const* ctxs, int n, int rows, int cols, double scale_factor,
                                        int min_neighbors, int min_w, int min_h, int max_w, int max_h,
                                        int max_batch, int cand_capacity);
/* Enqueue detection of nframes 8-bit gray frames that already live in device memory
 * (frame f at d_frames + f*frame_stride_bytes, rows row_step_bytes apart). Asynchronous.
 * Frame lifetime: the kernels read d_frames in stream order, so the buffer may be recycled behind the batch on the
 * same stream (a double-buffered decoder).  The one exception is a batch that yields more raw candidates than the
 * workspace holds (cand_capacity): hc_detect_batch_wait then grows the workspace and
 *   - by default returns HC_ERR_CAND_OVERFLOW -- nothing was lost, enqueue the same batch again;
 *   - after hc_context_set_rerun_on_overflow(ctx, 1) re-reads d_frames and runs the batch again by itself: the
 *     caller thereby promises to keep the frames valid and unmodified until the wait returns.
 * Host-input batches (hc_detect_batch_enqueue_host, the synchronous calls, the detectFaces path) are staged in the
 * context's own memory and always re-run by themselves.  Same rule for hc_faces_batch_enqueue. */
HC_API int  hc_detect_batch_enqueue(hc_context* ctx, const uint8_t* d_frames, size_t frame_stride_bytes,
                                    size_t row_step_bytes, int nframes);
HC_API int  hc_context_set_rerun_on_overflow(hc_context* ctx, int allow);
/* Rectangle grouping ON the device (cv::groupRectangles inside the call at reference
 * src/tracker/detector.cc:25-31): with on != 0 every following batch of at least 8 frames (HC_GROUP_MIN_BATCH; smaller
 * batches and the synchronous calls are latency-bound and grouped faster by the host) sorts, decodes and groups its raw
 * candidates in its own last kernels (one workgroup per frame) and publishes the grouped boxes; hc_detect_batch_wait then
 * copies them out and the host neither sorts nor groups.  Same boxes in the same order as the host grouping (tested
 * against it and against the oracle).  A frame with more than 1024 raw candidates or more than 128 grouped boxes is
 * grouped on the host as before; hc_context_group_host_frames says how many frames of the last waited batch that
 * were (all of them while the switch is off or minNeighbors asks for the raw rectangles).  Default: the value of
 * HC_DEVICE_GROUP when the context was created (1 unless set to 0). */
HC_API int  hc_context_set_device_grouping(hc_context* ctx, int on);
HC_API int  hc_context_group_host_frames(const hc_context* ctx);
/* Same, from host memory (adds the H2D copy on the context stream; one copy for the whole batch when the
 * frames are tight and contiguous).  The copy is truly asynchronous -- and about 4x faster -- when the
 * frames live in page-locked memory: either allocate the frame pool with hc_host_alloc, or pin a pool the
 * caller already owns (e.g. the video decoder's) with hc_host_register.  The reference reads frames with
 * cv::VideoCapture into pageable cv::Mat storage (src/tracker/face_tracker.cc:93-111); that works too. */
HC_API int  hc_detect_batch_enqueue_host(hc_context* ctx, const uint8_t* h_frames, size_t frame_stride_bytes,
                                         size_t row_step_bytes, int nframes);
/* Wait for the enqueued batch, group rectangles, return them frame by frame:
 * frame f's boxes are out[offsets[f] .. offsets[f+1]) ; offsets has nframes+1 entries. */
HC_API int  hc_detect_batch_wait(hc_context* ctx, hc_rect* out, int out_capacity, int32_t* offsets);
/* Raw (ungrouped) candidates of the last waited batch, canonical order (scale,y,x) per frame. */
HC_API int  hc_batch_raw_candidates(hc_context* ctx, hc_rect* out, int out_capacity, int32_t* offsets);

/* ---- the caller of the path: FaceDetector::detectFaces (reference src/tracker/detector.cc:10-43) ----
 * BGR 8UC3 frame in, skin-filtered face rectangles out.  On device, in the reference's order:
 * cvtColor(BGR2YCrCb) + inRange(lower, upper) -> mask (:17-18); cvtColor(BGR2GRAY) + equalizeHist +
 * gray &= mask (:20-22); detectMultiScale(gray, faces, scale, 2, 0|CASCADE_SCALE_IMAGE, Size(30,30))
 * (:25-31); keep a rect iff norm(mean(mask(roi)))/256 >= 0.8 (:33-42).  lower/upper are the YCrCb
 * bounds cfg.detection.skin_lower/upper (reference src/structure.cc:75-82).
 * hc_detect_faces is the synchronous single-frame form (configures minNeighbors 2, minSize 30x30
 * itself).  The batched form needs a context configured with those values and
 * hc_context_enable_faces (allocates the BGR/mask workspace once). */
HC_API int hc_detect_faces(hc_cascade* c, const uint8_t* bgr, int rows, int cols, size_t step_bytes,
                           const int lower[3], const int upper[3], double scale_factor,
                           hc_rect* out, int out_capacity, int* out_count);
HC_API int hc_context_enable_faces(hc_context* ctx);
/* Scan-size resize ahead of detectFaces: cv::resize(orig, curr, cfg.scan.frame_size, 0, 0, INTER_AREA)
 * (reference src/tracker/face_tracker.cc:112-114).  After this call the faces enqueue functions take
 * src_rows x src_cols BGR frames and shrink them on device to the configured rows x cols: the box average
 * for integer ratios (1280x720 -> 640x360 under the shipped cfg), upstream's weighted cell coverage for any
 * other ratio.  A scan size larger than the source in either axis is HC_ERR_UNSUPPORTED. */
HC_API int hc_context_enable_scan_resize(hc_context* ctx, int src_rows, int src_cols);
HC_API int hc_faces_batch_enqueue(hc_context* ctx, const uint8_t* d_bgr, size_t frame_stride_bytes,
                                  size_t row_step_bytes, int nframes, const int lower[3], const int upper[3]);
HC_API int hc_faces_batch_enqueue_host(hc_context* ctx, const uint8_t* h_bgr, size_t frame_stride_bytes,
                                       size_t row_step_bytes, int nframes, const int lower[3], const int upper[3]);
HC_API int hc_faces_batch_wait(hc_context* ctx, hc_rect* out, int out_capacity, int32_t* offsets);

/* ---- box files: frame id -> rect list, the hand-over from a batched / multi-GPU detection run
 * to the sequential tracker, which consumes every box of every frame in frame order (reference
 * src/tracker/face_tracker.cc:123-148, :294-302; the reference only has a TODO for tracker metadata files,
 * src/file_io.h:41-48).  Little endian: "HCBX", int32 version = 1, int32 nframes, then per frame in ascending
 * frame id: int32 frame_id, int32 count, count x {int32 x, y, width, height}.  Same bytes as
 * ugproj-detector_amd/shard.py save_boxes / load_boxes.  Pure host code, needs no device.
 * save: frame i has id frame_ids[i] and rects[offsets[i] .. offsets[i+1]) (the layout hc_detect_batch_wait returns);
 *       frames may come in any order, ids must be distinct.
 * load: *nframes / *nrects always receive the file's sizes; with null output pointers it is a size query;
 *       HC_ERR_CAPACITY when the buffers are too small. */
HC_API int hc_boxes_save(const char* path, int nframes, const int32_t* frame_ids, const int32_t* offsets,
                         const hc_rect* rects);
HC_API int hc_boxes_load(const char* path, int32_t* frame_ids, int32_t* offsets, int frame_capacity,
                         hc_rect* rects, int rect_capacity, int* nframes, int* nrects);

/* ---- plan introspection (tests, bench accounting) -------------------------------------- */
typedef struct hc_level {
  float   scale;
  int32_t w, h, ystep;
  int32_t szw_w, szw_h;   /* valid origins x < szw_w, y < szw_h                     */
  int32_t y_end;          /* rows actually scanned: y < y_end (stripe rule)         */
  int32_t nx, ny;         /* scanned grid: ceil(szw_w/ystep) x ceil(y_end/ystep)    */
  int32_t win_w, win_h;   /* emitted rectangle size                                 */
  int32_t int_pitch;      /* integral row pitch in words                            */
  int64_t pyr_ofs;        /* byte offset of the level in the per-frame pyramid      */
  int64_t int_ofs;        /* word offset of the level in the per-frame integral     */
  int64_t grid_ofs;       /* element offset of the level in per-frame grid arrays   */
} hc_level;
HC_API int hc_context_levels(const hc_context* ctx, hc_level* levels, int cap, int* nlevels);
/* Same plan without a device (pure host): the scale ladder / level geometry that
 * detectMultiScale would use for a rows x cols image.  *nlevels = 0 when the image is
 * smaller than the window. */
HC_API int hc_plan_query(const hc_cascade* c, int rows, int cols, double scale_factor, int min_w, int min_h,
                         int max_w, int max_h, hc_level* levels, int cap, int* nlevels, int64_t* algorithmic_bytes);
/* Host rectangle grouping used inside the call: cv::groupRectangles(rects, group_threshold, eps)
 * semantics (GROUP_EPS = 0.2 in detectMultiScale).  In and out may alias. */
HC_API int hc_group_rectangles(const hc_rect* in, int n, int group_threshold, double eps, hc_rect* out, int cap,
                               int* nout);
/* Algorithmic bytes per frame of the configured plan: W*H + sum 16*(w+1)*(h+1). */
HC_API int64_t hc_context_algorithmic_bytes(const hc_context* ctx);

/* ---- debug taps (tests only; copy device intermediates of the LAST batch to host) ------ */
typedef enum hc_tap {
  HC_TAP_PYRAMID = 0,   /* u8  [h][w] tight                                          */
  HC_TAP_SUM = 1,       /* i32 [(h+1)][(w+1)] tight                                  */
  HC_TAP_SQSUM = 2,     /* i32 [(h+1)][(w+1)] tight                                  */
  HC_TAP_VNF = 3,       /* f32 [ny][nx], 0 = not evaluable                           */
  HC_TAP_STAGES = 4,    /* i8  [ceil(szw_h/ystep)][nx]: -2 unvisited, -1 not evaluable,
                           k = stages passed (nstages = detection)                   */
  HC_TAP_FACE_MASK = 5, /* u8  [rows][cols] skin mask of the detectFaces path (level ignored) */
  HC_TAP_FACE_GRAY = 6, /* u8  [rows][cols] equalised gray & mask (level ignored)            */
  HC_TAP_FACE_BGR = 7   /* u8  [rows][cols][3] scan-size BGR frame after the INTER_AREA resize */
} hc_tap;
/* Taps of kind HC_TAP_STAGES are only recorded when enabled before the enqueue. */
HC_API int hc_context_enable_stage_taps(hc_context* ctx, int enable);
HC_API int hc_context_read_tap(hc_context* ctx, int frame, int level, int what, void* dst, size_t dst_bytes);

/* ---- profiling: per-kernel device time of the last waited batch (HIP events) ----------- */
/* Slots: 0 k_pyramid_integral (fused exact resize + 2-D integral; in the unfused debug mode
 * HC_FUSED_PREP=0 this slot times k_resize), 1 k_integral_rows and 2 k_integral_cols (0 ms when
 * fused), 3 k_variance, 4 k_cascade (all cascade launches of the batch). */
#define HC_NKERNELS 5
HC_API int hc_context_enable_timing(hc_context* ctx, int enable);
HC_API int hc_context_kernel_ms(hc_context* ctx, float ms[HC_NKERNELS]);
HC_API const char* hc_kernel_name(int i);
/* Number of leading stages that ran as compiled code (hc_cascade_compiled_stages) in every cascade launch of the last
 * enqueued batch; 0 when any of them ran the generic kernel (other cascade, HC_SPEC=0, non-default band shape). */
HC_API int hc_context_compiled_stages(hc_context* ctx, int* nstages);

/* What the LAST enqueue of a context actually launched -- which kernel build, which prep / variance route, and the
 * frame-chunk values both workgroup remaps ran with.  Tests and bench.py read this back instead of trusting that a tuning
 * variable reached the launcher (every HC_* variable is parsed once, in hc_context_create, into the context).
 * No counterpart in the reference: cv::CascadeClassifier has one evaluation path. */
typedef enum hc_variant {
  HC_VARIANT_NONE = 0,        /* no launch for this scan-step class (no levels)                              */
  HC_VARIANT_F32 = 1,         /* generic kernel, f32 rect weights                                            */
  HC_VARIANT_INT = 2,         /* generic kernel, exact integer rect weights                                  */
  HC_VARIANT_SPEC = 3,        /* compiled early stages (csrc/spec) + generic later stages, 32-bit tile       */
  HC_VARIANT_P16 = 4,         /* compiled early stages on the 16-bit tile, two windows per lane in row form  */
  HC_VARIANT_CV24 = 5,        /* 2.4 semantics on the tiled band kernel (stage chain)                        */
  HC_VARIANT_CV24_TREE = 6,   /* 2.4 semantics, stage tree swept in preorder on the tiles                    */
  HC_VARIANT_CV24_WINDOW = 7, /* 2.4 semantics, window-at-a-time kernel                                      */
  HC_VARIANT_CV24_CHAINS = 8  /* 2.4 semantics, stage tree as alternative chains on the tiled band kernel    */
} hc_variant;
typedef enum hc_prep_route {
  HC_PREP_NONE = 0, HC_PREP_FUSED = 1, HC_PREP_PARALLEL = 2, HC_PREP_SHARED_FOLLOWER = 3,
  HC_PREP_CV24_FUSED = 4, HC_PREP_CV24_PARALLEL = 5
} hc_prep_route;
typedef enum hc_variance_route {
  HC_VAR_NONE = 0,
  HC_VAR_KERNEL = 1,          /* separate k_variance launch                                                  */
  HC_VAR_RING = 2,            /* fused: register rings (and LDS rings for the grid columns beyond them)      */
  HC_VAR_READBACK = 3,        /* fused: integrals read back from global memory                               */
  HC_VAR_LEADER = 4,          /* shared pass: the leader's factors (same window, or emitted for this member) */
  HC_VAR_RING_STRIPS = 5,     /* fused, level cut into column strips, rings per strip                        */
  HC_VAR_CV24_KERNEL = 6      /* 2.4: separate f64 variance kernel                                           */
} hc_variance_route;
typedef struct hc_launch_info {
  int32_t prep_route, variance_route;
  int32_t pi_frame_chunk, pi_remapped, pi_tail_frames;       /* k_pyramid_integral: chunk size given, 1 if the remap ran,
                                                                 frames in the short last chunk                          */
  int32_t casc_frame_chunk, casc_remapped, casc_tail_frames; /* k_cascade, same meaning                                  */
  int32_t variant[2];                                        /* hc_variant of the scan-step-1 / scan-step-2 launch       */
  int32_t compiled_stages[2];
  int32_t segments;                                          /* 1: bands were walked in segments (small batches)         */
  int32_t side_stream;                                       /* 1: the two cascade launches ran on two streams           */
  int32_t sq_stored;                                         /* 1: the squares' integral went to global memory           */
  int32_t nframes;
  int32_t band_step[2];                                      /* grid rows per band of the two launches: the kernel's 16 (32), or one
                                                                 less where that fits one more workgroup per CU (HC_BAND_TRIM)          */
  int32_t lds_bytes[2];                                      /* LDS per workgroup of the two launches                    */
  int32_t device_grouping;                                   /* 1: the batch grouped its rectangles on the device (K6)   */
} hc_launch_info;
HC_API int hc_context_last_launch(const hc_context* ctx, hc_launch_info* out);

/* The column strips the fused prep kernel (k_pyramid_strips) would run for this call on a rows x cols image: pure host code, for
 * the CPU tests of the planner.  strips[12 * i ..] = level, c0, c1, ce (stored pixel columns [c0, c1), computed [c0, ce)), g0, g1
 * (grid columns whose variance factors the strip emits), sx0, sxn (staged source bytes of a row), left (index of the left neighbour,
 * -1: none), has_right, carry_ofs, 0.  comp_max: pixel columns a strip computes at most, 0 = the library's choice (the widest of
 * 832 / 640 / 512 / 384 that leaves two workgroups per CU).  *lds_bytes = LDS per workgroup, 0 when the geometry has no strip plan
 * (*nstrips = 0 then).  semantics as hc_context_set_semantics. */
HC_API int hc_debug_plan_strips(const hc_cascade* c, int rows, int cols, double scale_factor, int min_w, int min_h, int max_w,
                                int max_h, int semantics, int comp_max, int32_t* strips, int cap, int* nstrips, int64_t* lds_bytes);

/* Test entry of the result side alone (K6 and the host grouping): nkeys raw candidate keys
 * (frame << 40 | level << 32 | grid_y << 16 | grid_x, in any order; levels as hc_context_levels lists them) take the
 * place of a batch's candidates; hc_detect_batch_wait / hc_batch_raw_candidates then return what nframes frames with
 * those candidates would have returned. */
HC_API int hc_debug_enqueue_keys(hc_context* ctx, const unsigned long long* keys, int nkeys, int nframes);

/* Device and pinned-host allocations the library has made in this process so far (hipMalloc / hipHostMalloc
 * calls): tests use it to show that a steady state allocates nothing. */
HC_API long long hc_debug_alloc_count(void);

HC_API const char* hc_last_error(void);
HC_API const char* hc_version(void);

#ifdef __cplusplus
}
#endif
#endif
