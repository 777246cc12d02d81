// api_faces.cpp -- the caller of the path on the device: FaceDetector::detectFaces (reference src/tracker/detector.cc:10-43).
// Skin mask, equalised gray & mask, the detectMultiScale call, the skin-ratio post-filter; optionally the tracker's
// scan-size INTER_AREA resize ahead of it (src/tracker/face_tracker.cc:112-114).

#include <cmath>

#include "hc_context.h"

using hc::dev_alloc;
using hc::dev_upload;

extern "C" {

// ---- FaceDetector::detectFaces (reference src/tracker/detector.cc:10-43) ------------------------------
int hc_context_enable_faces(hc_context* ctx) {
  if (!ctx || !ctx->configured) return hc::set_err(HC_ERR_STATE, "context not configured");
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
  hc::count_alloc(); hc::count_alloc();
  HIPCHK(hipHostMalloc((void**)&ctx->h_boxes, (size_t)ctx->box_cap * 5 * sizeof(int), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&ctx->h_boxsums, (size_t)ctx->box_cap * sizeof(uint32_t), hipHostMallocDefault));
  ctx->faces_enabled = true;
  return HC_OK;
}

int hc_context_enable_scan_resize(hc_context* ctx, int src_rows, int src_cols) {
  if (!ctx || !ctx->configured || !ctx->faces_enabled) return hc::set_err(HC_ERR_STATE, "call hc_context_enable_faces first");
  if (src_rows < ctx->plan.rows || src_cols < ctx->plan.cols)
    return hc::set_err(HC_ERR_UNSUPPORTED, "INTER_AREA scan resize: the scan size must not exceed the source size in either axis");
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
    if (e) return hc::set_err(HC_ERR_HIP, std::string("INTER_AREA launch: ") + hipGetErrorString((hipError_t)e));
    d_bgr = ctx->d_bgr; frame_stride = dframe; row_step = drow;
  }
  hc::SkinBounds sb{};
  for (int k = 0; k < 3; k++) { sb.lo[k] = lower[k]; sb.hi[k] = upper[k]; }
  const size_t npix = (size_t)ctx->plan.rows * ctx->plan.cols;
  int e = hc::launch_face_prep(ctx->stream, nframes, d_bgr, frame_stride, row_step, ctx->plan.rows, ctx->plan.cols, sb,
                               ctx->d_gray, ctx->d_mask, npix, ctx->d_hist, ctx->d_lut);
  if (e) return hc::set_err(HC_ERR_HIP, std::string("face prep launch: ") + hipGetErrorString((hipError_t)e));
  // integral of the mask (sum only) for the skin-ratio post-filter
  const double equiv = (double)nframes * (double)npix / (1280.0 * 720.0);
  (void)equiv;
  if (hc::use_fused_prep(ctx, nframes, false)) {
    e = hc::launch_pyramid_integral(ctx->stream, nframes, 1, ctx->plan.cols, ctx->d_mask, npix, ctx->plan.cols, ctx->plan.rows,
                                    ctx->plan.cols, ctx->d_idlevel, ctx->d_idx, ctx->d_idy, ctx->d_msum, nullptr,
                                    ctx->mask_int_words, nullptr, 0, hc::prep_knobs(ctx));
  } else {   // small batch: row scans, then column scans, each fully parallel (same policy as the pyramid prep)
    e = hc::launch_integral_rows(ctx->stream, nframes, ctx->n_mask_items[0], ctx->d_mask, npix, ctx->d_idlevel,
                                 ctx->d_mask_items[0], ctx->d_msum, nullptr, ctx->mask_int_words);
    if (!e) e = hc::launch_integral_cols(ctx->stream, nframes, ctx->n_mask_items[1], ctx->d_idlevel, ctx->d_mask_items[1],
                                         ctx->d_msum, nullptr, ctx->mask_int_words);
  }
  if (e) return hc::set_err(HC_ERR_HIP, std::string("mask integral launch: ") + hipGetErrorString((hipError_t)e));
  int rc = hc::enqueue_device(ctx, ctx->d_gray, npix, ctx->plan.cols, nframes);
  if (rc == HC_OK) ctx->faces_pending = true;
  return rc;
}

int hc_faces_batch_enqueue(hc_context* ctx, const uint8_t* d_bgr, size_t frame_stride_bytes, size_t row_step_bytes,
                           int nframes, const int lower[3], const int upper[3]) {
  if (!ctx || !ctx->configured || !ctx->faces_enabled) return hc::set_err(HC_ERR_STATE, "call hc_context_enable_faces first");
  if (!d_bgr || !lower || !upper || nframes <= 0 || nframes > ctx->max_batch) return hc::set_err(HC_ERR_ARG, "bad argument");
  if (row_step_bytes < (size_t)(ctx->src_cols ? ctx->src_cols : ctx->plan.cols) * 3) return hc::set_err(HC_ERR_ARG, "row step smaller than the row");
  if (ctx->pending) return hc::set_err(HC_ERR_STATE, "previous batch not waited");
  HIPCHK(hipSetDevice(ctx->device));
  return faces_enqueue(ctx, d_bgr, frame_stride_bytes, row_step_bytes, nframes, lower, upper);
}

int hc_faces_batch_enqueue_host(hc_context* ctx, const uint8_t* h_bgr, size_t frame_stride_bytes, size_t row_step_bytes,
                                int nframes, const int lower[3], const int upper[3]) {
  if (!ctx || !ctx->configured || !ctx->faces_enabled) return hc::set_err(HC_ERR_STATE, "call hc_context_enable_faces first");
  if (!h_bgr || !lower || !upper || nframes <= 0 || nframes > ctx->max_batch) return hc::set_err(HC_ERR_ARG, "bad argument");
  const size_t rows = ctx->src_rows ? ctx->src_rows : ctx->plan.rows;
  const size_t rowb = (size_t)(ctx->src_cols ? ctx->src_cols : ctx->plan.cols) * 3;
  if (row_step_bytes < rowb) return hc::set_err(HC_ERR_ARG, "row step smaller than the row");
  if (ctx->pending) return hc::set_err(HC_ERR_STATE, "previous batch not waited");
  HIPCHK(hipSetDevice(ctx->device));
  uint8_t* stage = ctx->src_rows ? ctx->d_src_bgr : ctx->d_bgr;
  for (int f = 0; f < nframes; f++)
    HIPCHK(hipMemcpy2DAsync(stage + (size_t)f * rows * rowb, rowb, h_bgr + (size_t)f * frame_stride_bytes,
                            row_step_bytes, rowb, rows, hipMemcpyHostToDevice, ctx->stream));
  return faces_enqueue(ctx, stage, rows * rowb, rowb, nframes, lower, upper);
}

}  // extern "C"

int hc::faces_wait_core(hc_context* ctx, std::vector<std::vector<hc_rect>>& grouped) {
  if (!ctx->pending || !ctx->faces_pending) return hc::set_err(HC_ERR_STATE, "no detectFaces batch enqueued");
  ctx->faces_pending = false;
  int rc = hc::wait_core(ctx, grouped);
  if (rc != HC_OK) return rc;
  // skin post-filter (detector.cc:33-42): m = norm(mean(mask(roi))); keep iff !(m/256 < 0.8)
  int nb = 0;
  for (size_t f = 0; f < grouped.size(); f++)
    for (const hc_rect& r : grouped[f]) {
      if (nb >= ctx->box_cap) return hc::set_err(HC_ERR_CAPACITY, "too many rectangles for the skin filter workspace");
      int* b = ctx->h_boxes + 5 * nb++;
      b[0] = (int)f; b[1] = r.x; b[2] = r.y; b[3] = r.width; b[4] = r.height;
    }
  if (nb) {
    HIPCHK(hipMemcpyAsync(ctx->d_boxes, ctx->h_boxes, (size_t)nb * 5 * sizeof(int), hipMemcpyHostToDevice, ctx->copy_stream));
    int e = hc::launch_box_mask_sums(ctx->copy_stream, ctx->d_msum, ctx->mask_int_words, ctx->mask_pitch, ctx->d_boxes, nb,
                                     ctx->d_boxsums);
    if (e) return hc::set_err(HC_ERR_HIP, std::string("k_box_mask_sums launch: ") + hipGetErrorString((hipError_t)e));
    HIPCHK(hipMemcpyAsync(ctx->h_boxsums, ctx->d_boxsums, (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->copy_stream));
    HIPCHK(hipEventRecord(ctx->ev_copied, ctx->copy_stream));
    if (int rcw = hc::wait_event_polling(ctx->ev_copied)) return rcw;
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

extern "C" {

int hc_faces_batch_wait(hc_context* ctx, hc_rect* out, int out_capacity, int32_t* offsets) {
  if (!ctx || !ctx->configured) return hc::set_err(HC_ERR_STATE, "context not configured");
  if (!offsets || (!out && out_capacity > 0)) return hc::set_err(HC_ERR_ARG, "null output");
  std::vector<std::vector<hc_rect>> grouped;
  int rc = hc::faces_wait_core(ctx, grouped);
  if (rc != HC_OK) return rc;
  return hc::emit_rects(grouped, out, out_capacity, offsets);
}

int hc_detect_faces(hc_cascade* c, const uint8_t* bgr, int rows, int cols, size_t step_bytes, const int lower[3],
                    const int upper[3], double scale_factor, hc_rect* out, int out_capacity, int* out_count) {
  if (!c || !bgr || !lower || !upper || !out_count) return hc::set_err(HC_ERR_ARG, "null argument");
  if (!(scale_factor > 1.0)) return hc::set_err(HC_ERR_ARG, "scaleFactor must be > 1");
  *out_count = 0;
  std::lock_guard<std::mutex> lock(c->mu);
  hc_context* ctx = nullptr;
  // the reference's literals: minNeighbors 2, minSize 30x30, maxSize unset (detector.cc:28-31)
  int rc = hc::sync_context(c, rows, cols, scale_factor, 2, 30, 30, 0, 0, &ctx);
  if (rc) return rc;
  if ((rc = hc_context_enable_faces(ctx))) return rc;
  if ((rc = hc_faces_batch_enqueue_host(ctx, bgr, 0, step_bytes, 1, lower, upper))) return rc;
  std::vector<std::vector<hc_rect>> grouped;
  if ((rc = hc::faces_wait_core(ctx, grouped))) return rc;
  return hc::deliver_sync_result(c, grouped[0], out, out_capacity, out_count);
}

}  // extern "C"
