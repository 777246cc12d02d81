// api_wait.cpp -- the result side of a batch: wait for the kernels' published flag, decode the raw candidate keys into
// rectangles (Rect(cvRound(x * scale), ...)), group them per frame (cv::groupRectangles, A.10) and hand them
// over frame by frame -- what the reference's caller consumes at src/tracker/face_tracker.cc:290-304.

#include <algorithm>
#include <chrono>
#include <cmath>

#include "group_rects.h"
#include "hc_context.h"

namespace hc {
// Host wait for an event WITHOUT the runtime's blocking wait: hipEventSynchronize / hipStreamSynchronize on a stream whose
// hardware queue also carries another context's pending copy return only when THAT copy is done (one batch late).
int wait_event_polling(hipEvent_t ev) {
  for (unsigned spins = 0;; spins++) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return HC_OK;
    if (e != hipErrorNotReady) return hc::set_err(HC_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
    if (spins > 64) std::this_thread::yield();
  }
}

// The batch's last kernel has stored sequence number pub_seq (k_publish_flag): poll for it.  Every millisecond or so the
// batch's completion event is asked for errors, so a faulted kernel ends the wait with an error instead of a hang.
static int wait_published(hc_context* ctx) {
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
      if (e != hipSuccess && e != hipErrorNotReady) return hc::set_err(HC_ERR_HIP, std::string("batch failed: ") + hipGetErrorString(e));
      if (e == hipSuccess && ++done_polls > 200)
        return hc::set_err(HC_ERR_HIP, "the batch completed but its result flag was never published");
    }
  }
}

// ---- raw candidates: keys -> upstream's order -> rectangles --------------------------------------------------------
static hc_rect decode_key(const hc_context* ctx, unsigned long long k) {
  const hc::DevLevel& L = ctx->plan.levels[(k >> 32) & 0xFF];
  const int gy = (int)((k >> 16) & 0xFFFF), gx = (int)(k & 0xFFFF);
  if (ctx->semantics == 1) {
    // 2.4 (B.9): cvRound(x * factor) with the f64 factor of the scale loop
    const double factor = ctx->cv.lv[(k >> 32) & 0xFF].factor;
    return hc_rect{(int)std::nearbyint(gx * L.ystep * factor), (int)std::nearbyint(gy * L.ystep * factor), L.win_w, L.win_h};
  }
  // Rect(cvRound(x*scale), cvRound(y*scale), winSize): int -> f32 times f32 scale
  return hc_rect{(int)std::nearbyintf((float)(gx * L.ystep) * L.scale), (int)std::nearbyintf((float)(gy * L.ystep) * L.scale),
                 L.win_w, L.win_h};
}

// canonical single-thread order: frame -> scale -> y -> x  == ascending key; last_raw_ofs = frame boundaries in it
static void sort_raw_keys(hc_context* ctx) {
  std::vector<unsigned long long>& K = ctx->raw_keys;
  std::sort(K.begin(), K.end());
  const int nframes = ctx->last_frames;
  ctx->last_raw_ofs.assign(nframes + 1, 0);
  size_t i = 0;
  for (int f = 0; f < nframes; f++) {
    ctx->last_raw_ofs[f] = (int32_t)i;
    while (i < K.size() && (int)(K[i] >> 40) == f) i++;
  }
  ctx->last_raw_ofs[nframes] = (int32_t)i;
}

// the raw rectangles of the last waited batch (hc_batch_raw_candidates, frames the device left to the host)
static void decode_raw(hc_context* ctx) {
  if (ctx->raw_decoded) return;
  sort_raw_keys(ctx);
  ctx->last_raw.resize(ctx->raw_keys.size());
  for (size_t i = 0; i < ctx->raw_keys.size(); i++) ctx->last_raw[i] = decode_key(ctx, ctx->raw_keys[i]);
  ctx->raw_decoded = true;
}

// the grouping threshold a context's batches run with: 0 = the raw rectangles are the result
int group_threshold(const hc_context* ctx) {
  if (ctx->semantics == 1) return ctx->min_neighbors != 0 ? std::max(ctx->min_neighbors, 1) : 0;   // 2.4: minNeighbors 0 = raw
  return std::max(ctx->min_neighbors, 0);
}

static void group_frame(const hc_context* ctx, std::vector<hc_rect>& grp) {
  const int thr = group_threshold(ctx);
  if (thr > 0) hc::group_rectangles(grp, thr, 0.2);   // GROUP_EPS upstream
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
    if (ctx->h_pub[2])
      return set_err(HC_ERR_HIP, "k_pyramid_strips: a column strip gave up waiting for its left neighbour's carries (workgroups were "
                                 "not started in id order?); HC_VAR_SPLIT=0 avoids the strip form");
    if (ctx->k.debug_max_stages > 0 && count > ctx->cand_cap) count = 0;   // timing experiments only
    if (count <= ctx->cand_cap) break;
    // More raw candidates than the workspace holds (the counter keeps counting past the capacity, so the need is
    // known exactly).  OpenCV would simply return more rectangles: grow the buffers and run the batch again.
    if (attempt >= 2 || !ctx->inflight_src || (ctx->leader && ctx->leader->prep_seq != ctx->seen_seq))
      return hc::set_err(HC_ERR_CAND_OVERFLOW, "raw candidate workspace overflow and the batch cannot be re-run "
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
    if (rc) return hc::set_err(HC_ERR_CAND_OVERFLOW, "raw candidate workspace overflow; growing it failed: " + hc::last_err());
    if (host_alloc_coherent((void**)&ctx->h_cand, want * sizeof(unsigned long long)) != HC_OK)
      return hc::set_err(HC_ERR_CAND_OVERFLOW, "raw candidate workspace overflow; growing the host buffer failed");
    ctx->cand_cap = want;
    if (!may_rerun)
      return hc::set_err(HC_ERR_CAND_OVERFLOW, "raw candidate workspace overflow: " + std::to_string(count) + " candidates; the workspace "
                                           "now holds " + std::to_string(want) + " -- enqueue the batch again (caller-owned device "
                                           "frames are only re-read by the library after hc_context_set_rerun_on_overflow(ctx, 1))");
    if ((rc = enqueue_device(ctx, ctx->inflight_src, ctx->inflight_frame_stride, ctx->inflight_row_step, nframes))) return rc;
    ctx->pending = 0;
  }
  if (ctx->timed_last && !ctx->plan.levels.empty())
    for (int i = 0; i < HC_NKERNELS; i++) HIPCHK(hipEventElapsedTime(&ctx->last_ms[i], ctx->ev[i], ctx->ev[i + 1]));
  ctx->last_frames = nframes;
  // the raw keys leave the published buffer now (the next batch's kernels overwrite it); sorted and decoded when needed
  ctx->raw_keys.assign(ctx->h_cand, ctx->h_cand + count);
  ctx->raw_decoded = false;
  ctx->last_raw.clear();
  ctx->last_raw_ofs.assign(nframes + 1, 0);
  grouped.assign(nframes, {});
  ctx->last_group_host_frames = nframes;
  if (ctx->group_ran) {
    // K6: the batch grouped its own candidates (group_kernels.hip); the boxes sit in page-locked memory, frame by frame.
    // A frame whose candidates overflowed its bin (or its boxes their slot) carries 0xFFFFFFFF and is grouped here.
    static_assert(sizeof(hc_rect) == sizeof(GroupBox), "hc_rect and GroupBox share a layout");
    std::vector<int> left;
    for (int f = 0; f < nframes; f++) {
      const unsigned int nb = ctx->h_gcount[f];
      if (nb == 0xFFFFFFFFu) { left.push_back(f); continue; }
      const hc_rect* src = reinterpret_cast<const hc_rect*>(ctx->h_gboxes) + (size_t)f * HC_GROUP_OUT;
      grouped[f].assign(src, src + nb);
    }
    ctx->last_group_host_frames = (int)left.size();
    if (left.empty()) return HC_OK;
    decode_raw(ctx);
    for (int f : left) {
      grouped[f].assign(ctx->last_raw.begin() + ctx->last_raw_ofs[f], ctx->last_raw.begin() + ctx->last_raw_ofs[f + 1]);
      group_frame(ctx, grouped[f]);
    }
    return HC_OK;
  }
  sort_raw_keys(ctx);
  ctx->last_raw.resize(count);
  auto do_frame = [&](int f) {
    std::vector<hc_rect>& grp = grouped[f];
    const size_t a = (size_t)ctx->last_raw_ofs[f], b = (size_t)ctx->last_raw_ofs[f + 1];
    grp.reserve(b - a);
    for (size_t i = a; i < b; i++) {
      const hc_rect r = decode_key(ctx, ctx->raw_keys[i]);
      grp.push_back(r);
      ctx->last_raw[i] = r;
    }
    group_frame(ctx, grp);
  };
  if (nframes >= 4 && count >= 512 && ctx->k.host_threads > 1) {
    if (!ctx->pool) ctx->pool.reset(new WorkerPool(ctx->k.host_threads - 1));
    ctx->pool->parallel_for(nframes, do_frame);
  } else {
    for (int f = 0; f < nframes; f++) do_frame(f);
  }
  ctx->raw_decoded = true;
  return HC_OK;
}

void decode_raw_public(hc_context* ctx) { decode_raw(ctx); }

int emit_rects(const std::vector<std::vector<hc_rect>>& grouped, hc_rect* out, int out_capacity, int32_t* offsets) {
  int total = 0;
  const int nframes = (int)grouped.size();
  for (int f = 0; f < nframes; f++) {
    offsets[f] = total;
    if (total + (int)grouped[f].size() > out_capacity) return hc::set_err(HC_ERR_CAPACITY, "output rectangle buffer too small");
    for (const hc_rect& r : grouped[f]) out[total++] = r;
  }
  offsets[nframes] = total;
  return HC_OK;
}
}  // namespace hc

extern "C" {

int hc_detect_batch_wait(hc_context* ctx, hc_rect* out, int out_capacity, int32_t* offsets) {
  if (!ctx || !ctx->configured) return hc::set_err(HC_ERR_STATE, "context not configured");
  if (!ctx->pending) return hc::set_err(HC_ERR_STATE, "no batch enqueued");
  if (ctx->faces_pending) return hc::set_err(HC_ERR_STATE, "a detectFaces batch is pending: call hc_faces_batch_wait");
  if (!offsets || (!out && out_capacity > 0)) return hc::set_err(HC_ERR_ARG, "null output");
  std::vector<std::vector<hc_rect>> grouped;
  int rc = hc::wait_core(ctx, grouped);
  if (rc != HC_OK) return rc;
  return hc::emit_rects(grouped, out, out_capacity, offsets);
}

int hc_batch_raw_candidates(hc_context* ctx, hc_rect* out, int out_capacity, int32_t* offsets) {
  if (!ctx || !offsets) return hc::set_err(HC_ERR_ARG, "null argument");
  hc::decode_raw_public(ctx);
  if ((int)ctx->last_raw.size() > out_capacity) return hc::set_err(HC_ERR_CAPACITY, "output rectangle buffer too small");
  for (size_t i = 0; i < ctx->last_raw.size(); i++) out[i] = ctx->last_raw[i];
  for (size_t i = 0; i < ctx->last_raw_ofs.size(); i++) offsets[i] = ctx->last_raw_ofs[i];
  return HC_OK;
}

}  // extern "C"
