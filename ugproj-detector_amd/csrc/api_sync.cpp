// api_sync.cpp -- the synchronous drop-in calls: the literal replacement of
//   cascade.detectMultiScale(gray, faces, scale, 2, 0|CASCADE_SCALE_IMAGE, Size(30,30))     reference src/tracker/detector.cc:25-31
// one frame at a time from host memory, through per-cascade contexts cached by (device, semantics, geometry, call parameters).

#include <algorithm>

#include "hc_context.h"

namespace hc {
// the per-cascade context of the synchronous calls, (re)configured for this geometry and these call parameters
int sync_context(hc_cascade* c, int rows, int cols, double scale_factor, int min_neighbors, int min_w, int min_h, int max_w,
                 int max_h, hc_context** out, int semantics) {
  int dev = 0, ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return hc::set_err(HC_ERR_NO_DEVICE, "no HIP device available (this backend has no CPU fallback)");
  HIPCHK(hipGetDevice(&dev));
  c->pending_results.erase(std::this_thread::get_id());   // this thread's unread rectangles of an earlier call are dropped
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
  const size_t cache_size = (size_t)std::max(1, c->sync_cache_size);   // HC_SYNC_CONTEXTS, read when the cascade was loaded
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

}  // namespace hc

namespace {
using hc::dev_alloc;
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
  if (int rc = hc::ensure_frames(ctx)) return rc;
  HIPCHK(hipMemcpy2DAsync(ctx->d_color, rowb, img, step_bytes, rowb, rows, hipMemcpyHostToDevice, ctx->stream));
  int e = hc::launch_bgr2gray(ctx->stream, 1, ctx->d_color, need, rowb, channels, rows, cols, ctx->d_frames, (size_t)rows * cols);
  if (e) return hc::set_err(HC_ERR_HIP, std::string("k_bgr2gray launch: ") + hipGetErrorString((hipError_t)e));
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
  if (!c || !image || !out_count) return hc::set_err(HC_ERR_ARG, "null argument");
  if (!(scale_factor > 1.0)) return hc::set_err(HC_ERR_ARG, "scaleFactor must be > 1");
  if (channels != 1 && channels != 3 && channels != 4) return hc::set_err(HC_ERR_ARG, "8-bit image with 1, 3 or 4 channels required");
  if (rows <= 0 || cols <= 0 || step_bytes < (size_t)cols * channels) return hc::set_err(HC_ERR_ARG, "bad image geometry");
  *out_count = 0;
  std::lock_guard<std::mutex> lock(c->mu);
  hc_context* ctx = nullptr;
  int rc = hc::sync_context(c, rows, cols, scale_factor, min_neighbors, min_w, min_h, max_w, max_h, &ctx);
  if (rc) return rc;
  if (channels == 1) {
    if ((rc = hc_detect_batch_enqueue_host(ctx, image, 0, step_bytes, 1))) return rc;
  } else {
    if (ctx->pending) return hc::set_err(HC_ERR_STATE, "previous batch not waited");
    HIPCHK(hipSetDevice(ctx->device));
    if ((rc = stage_color_as_gray(ctx, image, rows, cols, channels, step_bytes))) return rc;
    if ((rc = hc::enqueue_device(ctx, ctx->d_frames, (size_t)rows * cols, cols, 1))) return rc;
  }
  std::vector<std::vector<hc_rect>> grouped;
  if ((rc = hc::wait_core(ctx, grouped))) return rc;
  return hc::deliver_sync_result(c, grouped[0], out, out_capacity, out_count);
}

int hc_detect_multiscale_reread(hc_cascade* c, hc_rect* out, int out_capacity, int* out_count) {
  if (!c || !out_count) return hc::set_err(HC_ERR_ARG, "null argument");
  std::lock_guard<std::mutex> lock(c->mu);
  const auto it = c->pending_results.find(std::this_thread::get_id());
  if (it == c->pending_results.end())
    return hc::set_err(HC_ERR_STATE, "this thread's last synchronous call on this cascade did not end in HC_ERR_CAPACITY");
  *out_count = (int)it->second.size();
  if ((int)it->second.size() > out_capacity || (!out && !it->second.empty())) return hc::set_err(HC_ERR_CAPACITY, "output rectangle buffer too small");
  for (size_t i = 0; i < it->second.size(); i++) out[i] = it->second[i];
  c->pending_results.erase(it);
  return HC_OK;
}
int hc_detect_multiscale_cv24_taps(hc_cascade* c, const uint8_t* gray, int rows, int cols, size_t step_bytes, double scale_factor,
                                   int min_neighbors, int flags, int min_w, int min_h, int max_w, int max_h, hc_rect* out,
                                   int out_capacity, int* out_count, int8_t* taps, size_t taps_capacity) {
  if (!c || !gray || !out_count) return hc::set_err(HC_ERR_ARG, "null argument");
  *out_count = 0;
  if (!(scale_factor > 1.0)) return hc::set_err(HC_ERR_ARG, "scaleFactor must be > 1");
  if (rows <= 0 || cols <= 0 || step_bytes < (size_t)cols) return hc::set_err(HC_ERR_ARG, "bad image geometry");
  if (!(flags & 2))   // CV_HAAR_SCALE_IMAGE: without it 2.4 scales the FEATURES instead of the image (not restated)
    return hc::set_err(HC_ERR_UNSUPPORTED, "legacy mode is the CASCADE_SCALE_IMAGE branch (the flag the reference passes)");
  if (cols > 65535 || rows > 65535) return hc::set_err(HC_ERR_UNSUPPORTED, "frame too large");
  std::lock_guard<std::mutex> lock(c->mu);
  hc_context* ctx = nullptr;
  int rc = hc::sync_context(c, rows, cols, scale_factor, min_neighbors, min_w, min_h, max_w, max_h, &ctx, 1);
  if (rc) return rc;
  const hc::Plan& P = ctx->plan;
  if (taps) {
    size_t need = 0;
    for (const hc::DevLevel& L : P.levels) need += (size_t)L.nx * L.ny;
    if (need > taps_capacity) return hc::set_err(HC_ERR_CAPACITY, "tap buffer too small");
  }
  ctx->taps_enabled = taps != nullptr;
  if ((rc = hc_detect_batch_enqueue_host(ctx, gray, 0, step_bytes, 1))) { ctx->taps_enabled = false; return rc; }
  std::vector<std::vector<hc_rect>> grouped;
  rc = hc::wait_core(ctx, grouped);
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
  if (!image || !gray) return hc::set_err(HC_ERR_ARG, "null argument");
  if ((channels != 3 && channels != 4) || rows <= 0 || cols <= 0 || step_bytes < (size_t)cols * channels ||
      gray_step_bytes < (size_t)cols)
    return hc::set_err(HC_ERR_ARG, "8-bit image with 3 or 4 channels required");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return hc::set_err(HC_ERR_NO_DEVICE, "no HIP device available (this backend has no CPU fallback)");
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
  if (e != hipSuccess) return hc::set_err(HC_ERR_HIP, std::string("hc_convert_bgr2gray: ") + hipGetErrorString(e));
  return HC_OK;
}

}  // extern "C"
