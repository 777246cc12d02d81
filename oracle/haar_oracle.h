/*
 * haar_oracle.h -- C interface of the CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * This library restates, on the CPU and in canonical single-thread order, the
 * algorithm that the reference obtains from
 *     cv::CascadeClassifier::load            (reference src/file_io.cc:11)
 *     cv::CascadeClassifier::detectMultiScale(reference src/tracker/detector.cc:25-31)
 * i.e. OpenCV >= 4.0 semantics for an old-format Haar cascade.
 *
 * PARITY UNPINNED: OpenCV is an un-vendored dependency of the reference, is absent
 * from the build container, and the reference holds no golden vectors or tests for
 * this path.  The restatement follows the published upstream
 * algorithm; it is pinned only by hand-derived known answers, an independent numpy
 * restatement (oracle/np_oracle.py) and the committed fixtures under tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
 * call this library.  The product (ugproj-detector_amd/) never does.
 */
#ifndef HAAR_ORACLE_H
#define HAAR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_cascade orc_cascade;

typedef struct orc_rect { int32_t x, y, w, h; } orc_rect;

/* One pyramid level. */
typedef struct orc_level {
  float   scale;      /* stored f32 scale factor */
  int32_t w, h;       /* level image size (sz) */
  int32_t ystep;      /* 1 or 2 */
  int32_t szw_w, szw_h; /* working size: valid window origins are x<szw_w, y<szw_h */
  int32_t stripe;     /* stripe size in pixels (multiple of ystep) */
  int32_t y_end;      /* y loop runs y=0,ystep,.. < y_end  (= min(nstripes*stripe, szw_h)) */
  int32_t nx, ny;     /* number of grid positions visited candidates: ceil(szw_w/ystep), ceil(y_end/ystep) */
  int32_t win_w, win_h; /* cvRound(orig_win * scale): size of emitted rectangles */
} orc_level;

/* ---- model ------------------------------------------------------------ */
int  orc_cascade_load_xml(const char* path, orc_cascade** out);  /* old opencv-haar-classifier XML */
int  orc_cascade_load_hcc(const char* path, orc_cascade** out);  /* packed fixture format (tools/pack_cascade.py) */
void orc_cascade_free(orc_cascade* c);
/* info[0..7] = win_w, win_h, nstages, ntrees, nnodes, nleaves, max_nodes_per_tree, has_tilted */
int  orc_cascade_info(const orc_cascade* c, int32_t info[8]);
/* Flat dump for loader-parity tests.  Any pointer may be NULL.
 *  stage_thr[nstages] (f32, already minus THRESHOLD_EPS), stage_ntrees[nstages],
 *  tree_nnodes[ntrees],
 *  node_rects[nnodes*15] = 3 x (x,y,w,h,weight-as-int), node_thr[nnodes],
 *  node_left[nnodes], node_right[nnodes], leaves[nleaves]                       */
int  orc_cascade_dump(const orc_cascade* c, float* stage_thr, int32_t* stage_ntrees,
                      int32_t* tree_nnodes, int32_t* node_rects, float* node_thr,
                      int32_t* node_left, int32_t* node_right, float* leaves);

/* ---- stages of the path, individually (taps) -------------------------- */
int orc_plan(const orc_cascade* c, int rows, int cols, double scale_factor,
             int min_w, int min_h, int max_w, int max_h,
             orc_level* levels, int cap, int* nlevels, int* nstripes);

/* cv::resize(..., INTER_LINEAR_EXACT) for 8UC1. dst is drows x dcols, tight. */
int orc_resize_linear_exact(const uint8_t* src, int rows, int cols, size_t step,
                            uint8_t* dst, int drows, int dcols);

/* cv::integral(img, sum, sqsum, CV_32S, CV_32S): (rows+1)x(cols+1), tight. */
int orc_integral(const uint8_t* img, int rows, int cols, size_t step,
                 int32_t* sum, int32_t* sqsum);

/* cv::groupRectangles(rects, group_threshold, eps). Returns count in *nout. */
int orc_group_rectangles(const orc_rect* in, int n, int group_threshold, double eps,
                         orc_rect* out, int cap, int* nout);

/* ---- the whole call ---------------------------------------------------- */
/* detectMultiScale(gray, objects, scale_factor, min_neighbors, flags(ignored), minSize, maxSize).
 *  out/out_count   : grouped rectangles in single-thread OpenCV order (scale, y, x of first member)
 *  raw/raw_count   : optional raw candidates before grouping, canonical order scale->y->x
 *  taps            : optional int8 per grid position, levels concatenated, each level
 *                    laid out [gy][gx] with gx<ceil(szw_w/ystep), gy<ceil(szw_h/ystep):
 *                      -2 not visited (x-skip or stripe y-limit), -1 not evaluable
 *                      (setWindow false), k>=0 number of stages passed (nstages = detection)
 *  taps_cap        : capacity in bytes of taps.
 * Returns 0 on success, <0 on error (-3 capacity). */
int orc_detect(const orc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step,
               double scale_factor, int min_neighbors, int min_w, int min_h, int max_w, int max_h,
               orc_rect* out, int cap, int* out_count,
               orc_rect* raw, int raw_cap, int* raw_count,
               int8_t* taps, size_t taps_cap);

/* CPU baseline leg: run orc_detect on nframes tight frames with nthreads host threads
 * (frame-parallel), grouped counts per frame into counts[nframes]. */
int orc_detect_many(const orc_cascade* c, const uint8_t* frames, int nframes, int rows, int cols,
                    double scale_factor, int min_neighbors, int min_w, int min_h, int max_w, int max_h,
                    int nthreads, int32_t* counts);

/* ---- the caller of the path: FaceDetector::detectFaces (reference src/tracker/detector.cc:10-43)
 * frame: 8UC3 BGR, rows x cols, `step` bytes per row.  lower/upper: YCrCb skin bounds
 * (cfg.detection.skin_lower/upper, defaults (0,133,77)-(255,173,127) at src/structure.cc:75-82).
 * Pre-steps: cvtColor BGR->YCrCb + inRange -> mask; cvtColor BGR->GRAY; equalizeHist;
 * gray &= mask; then detectMultiScale(gray, faces, scale, 2, flags, Size(30,30)); then keep a rect
 * iff norm(mean(mask(roi)))/256 >= 0.8.  Optional taps: mask and the masked equalised gray (tight). */
int orc_detect_faces(const orc_cascade* c, const uint8_t* bgr, int rows, int cols, size_t step,
                     const int lower[3], const int upper[3], double scale_factor,
                     orc_rect* out, int cap, int* out_count, uint8_t* mask_tap, uint8_t* gray_tap);

/* cv::resize(src, dst, dsize, 0, 0, INTER_AREA) for 8-bit images with 1 or 3 interleaved channels, both
 * axes shrinking -- the scan-size resize of the tracker (reference src/tracker/face_tracker.cc:112-114;
 * 1280x720 -> 640x360 under conf/apink.mrchu.720p.cfg:4, but scan.frame_size accepts any size).
 * Integer factors: 2x2 (a+b+c+d+2)>>2 (ResizeAreaFastVec), otherwise cvRound(sum * (1.f/area))
 * (resizeAreaFast_).  Other ratios: computeResizeAreaTab + resizeArea_<uchar,float> (f32 accumulation in
 * table order, cvRound at the end).  Enlarging in either axis returns -4.  [UPSTREAM-RECALL, unpinned] */
int orc_resize_area(const uint8_t* src, int rows, int cols, int channels, size_t step,
                    uint8_t* dst, int drows, int dcols);

/* cvtColor(src, gray, COLOR_BGR2GRAY) for 8UC3 / 8UC4: (3735 B + 19235 G + 9798 R + 16384) >> 15 -- what
 * detectMultiScale applies to a multi-channel image and FaceDetector::detectFaces at detector.cc:20. */
int orc_bgr2gray(const uint8_t* src, int rows, int cols, int channels, size_t step, uint8_t* gray);

/* ---- T2: OpenCV 2.4.x legacy semantics (rules B.1-B.9 at the top of that section in haar_oracle.cpp):
 * cvHaarDetectObjectsForROC with CV_HAAR_SCALE_IMAGE, scalar C path, stage tree honoured (alt_tree).  Unpinned. */
int orc_resize_linear_cv24(const uint8_t* src, int rows, int cols, size_t step, uint8_t* dst, int drows, int dcols);
/* levels of the call: factors[i] (f64) and geom[7*i ..] = w, h, ystep, nx, ny, win_w, win_h */
int orc_plan_cv24(const orc_cascade* c, int rows, int cols, double scale_factor, int min_w, int min_h, int max_w, int max_h,
                  double* factors, int32_t* geom, int cap, int* nlevels);
/* taps: int8 per scanned position (levels concatenated, [gy][gx]): stages passed along the window's path through the
 * stage tree, | 64 when accepted.  min_neighbors 0 returns the raw candidates ungrouped, as upstream does. */
int orc_detect_cv24(const orc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step, double scale_factor,
                    int min_neighbors, int min_w, int min_h, int max_w, int max_h, orc_rect* out, int cap, int* out_count,
                    orc_rect* raw, int raw_cap, int* raw_count, int8_t* taps, size_t taps_cap);

/* Evaluate one window-sized patch (win_h x win_w, tight) at scale 1: returns number of
 * stages passed (nstages = accepted), -1 if not evaluable. margins (optional, nstages
 * doubles) receive stage_sum - stage_threshold for every stage evaluated regardless of
 * failure (used by the synthetic-face template search in tools/). */
int orc_eval_patch(const orc_cascade* c, const uint8_t* patch, double* margins);

#ifdef __cplusplus
}
#endif
#endif
