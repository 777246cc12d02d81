// api_inspect.cpp -- looking into a context: the plan (scale ladder, level geometry, algorithmic bytes), debug taps of the
// last batch's intermediates, per-kernel timing, and the report of what the last enqueue launched.  Tests and bench.py
// only; nothing here is on the detection path.

#include <algorithm>

#include "hc_context.h"

namespace {
hc_level to_public(const hc::DevLevel& L) {
  hc_level o{};
  o.scale = L.scale; o.w = L.w; o.h = L.h; o.ystep = L.ystep; o.szw_w = L.szw_w; o.szw_h = L.szw_h;
  o.y_end = L.y_end; o.nx = L.nx; o.ny = L.ny; o.win_w = L.win_w; o.win_h = L.win_h; o.int_pitch = L.int_pitch;
  o.pyr_ofs = L.pyr_ofs; o.int_ofs = L.int_ofs; o.grid_ofs = L.grid_ofs;
  return o;
}
const char* const kKernelNames[HC_NKERNELS] = {"k_pyramid_integral", "k_integral_rows", "k_integral_cols", "k_variance", "k_cascade"};
}  // namespace

extern "C" {

const char* hc_kernel_name(int i) { return (i >= 0 && i < HC_NKERNELS) ? kKernelNames[i] : ""; }

int hc_context_levels(const hc_context* ctx, hc_level* levels, int cap, int* nlevels) {
  if (!ctx || !ctx->configured || !nlevels) return hc::set_err(HC_ERR_STATE, "context not configured");
  const auto& Ls = ctx->plan.levels;
  *nlevels = (int)Ls.size();
  if ((int)Ls.size() > cap) return hc::set_err(HC_ERR_CAPACITY, "level buffer too small");
  for (size_t i = 0; i < Ls.size(); i++) levels[i] = to_public(Ls[i]);
  return HC_OK;
}

int hc_plan_query(const hc_cascade* c, int rows, int cols, double scale_factor, int min_w, int min_h, int max_w,
                  int max_h, hc_level* levels, int cap, int* nlevels, int64_t* algorithmic_bytes) {
  if (!c || !nlevels) return hc::set_err(HC_ERR_ARG, "null argument");
  hc::Plan P;
  int rc = hc::build_plan(c->model, rows, cols, scale_factor, min_w, min_h, max_w, max_h, 16, 16, P);
  if (rc != HC_OK) return hc::set_err(rc, "cannot plan this configuration");
  *nlevels = (int)P.levels.size();
  if (algorithmic_bytes) *algorithmic_bytes = P.algorithmic_bytes;
  if ((int)P.levels.size() > cap) return hc::set_err(HC_ERR_CAPACITY, "level buffer too small");
  for (size_t i = 0; i < P.levels.size(); i++) levels[i] = to_public(P.levels[i]);
  return HC_OK;
}
int64_t hc_context_algorithmic_bytes(const hc_context* ctx) {
  return (ctx && ctx->configured) ? ctx->plan.algorithmic_bytes : 0;
}

int hc_context_enable_stage_taps(hc_context* ctx, int enable) {
  if (!ctx) return hc::set_err(HC_ERR_ARG, "null context");
  ctx->taps_enabled = enable != 0;
  return HC_OK;
}

int hc_context_read_tap(hc_context* ctx, int frame, int level, int what, void* dst, size_t dst_bytes) {
  if (!ctx || !ctx->configured || !dst) return hc::set_err(HC_ERR_STATE, "context not configured");
  if (ctx->pending) return hc::set_err(HC_ERR_STATE, "batch still in flight");
  const bool frame_tap = what == HC_TAP_FACE_MASK || what == HC_TAP_FACE_GRAY || what == HC_TAP_FACE_BGR;
  if (frame < 0 || frame >= ctx->last_frames || (!frame_tap && (level < 0 || level >= (int)ctx->plan.levels.size())))
    return hc::set_err(HC_ERR_ARG, "frame / level out of range");
  HIPCHK(hipSetDevice(ctx->device));
  const hc::Plan& P = ctx->plan;
  const hc::DevLevel& L = P.levels[frame_tap ? 0 : level];
  switch (what) {
    case HC_TAP_PYRAMID: {
      const size_t n = (size_t)L.w * L.h;
      if (dst_bytes < n) return hc::set_err(HC_ERR_CAPACITY, "tap buffer too small");
      if (!ctx->d_pyr) return hc::set_err(HC_ERR_STATE, "no pyramid was materialised: enable stage taps before the enqueue");
      HIPCHK(hipMemcpy(dst, ctx->d_pyr + (size_t)frame * P.pyr_bytes + L.pyr_ofs, n, hipMemcpyDeviceToHost));
      return HC_OK;
    }
    case HC_TAP_SUM:
    case HC_TAP_SQSUM: {
      if (what == HC_TAP_SQSUM && !ctx->sq_valid)
        return hc::set_err(HC_ERR_STATE, "the squares' integral was not materialised: enable stage taps before the enqueue");
      const size_t wbytes = (size_t)(L.w + 1) * 4;
      if (dst_bytes < wbytes * (L.h + 1)) return hc::set_err(HC_ERR_CAPACITY, "tap buffer too small");
      const uint32_t* src = (what == HC_TAP_SUM ? ctx->d_sum : ctx->d_sq) + (size_t)frame * P.int_words + L.int_ofs;
      HIPCHK(hipMemcpy2D(dst, wbytes, src, (size_t)L.int_pitch * 4, wbytes, L.h + 1, hipMemcpyDeviceToHost));
      return HC_OK;
    }
    case HC_TAP_VNF: {
      // 2.4 semantics keep f64 factors in another buffer (d_vnf is a one-element stub there): the tap does not exist
      if (ctx->semantics == 1) return hc::set_err(HC_ERR_UNSUPPORTED, "HC_TAP_VNF does not exist under the 2.4 semantics (f64 factors, no gate)");
      const size_t n = (size_t)L.nx * L.ny * 4;
      if (dst_bytes < n) return hc::set_err(HC_ERR_CAPACITY, "tap buffer too small");
      if (n) HIPCHK(hipMemcpy(dst, ctx->d_vnf + (size_t)frame * P.grid_elems + L.grid_ofs, n, hipMemcpyDeviceToHost));
      return HC_OK;
    }
    case HC_TAP_STAGES: {
      if (!ctx->taps_enabled || !ctx->d_taps) return hc::set_err(HC_ERR_STATE, "stage taps were not enabled");
      const size_t n = (size_t)L.nx * L.ny_all;
      if (dst_bytes < n) return hc::set_err(HC_ERR_CAPACITY, "tap buffer too small");
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
      if (!ctx->faces_enabled) return hc::set_err(HC_ERR_STATE, "detectFaces path not enabled");
      const size_t n = (size_t)P.rows * P.cols * 3;
      if (dst_bytes < n) return hc::set_err(HC_ERR_CAPACITY, "tap buffer too small");
      HIPCHK(hipMemcpy(dst, ctx->d_bgr + (size_t)frame * n, n, hipMemcpyDeviceToHost));
      return HC_OK;
    }
    case HC_TAP_FACE_MASK:
    case HC_TAP_FACE_GRAY: {
      if (!ctx->faces_enabled) return hc::set_err(HC_ERR_STATE, "detectFaces path not enabled");
      const size_t n = (size_t)P.rows * P.cols;
      if (dst_bytes < n) return hc::set_err(HC_ERR_CAPACITY, "tap buffer too small");
      HIPCHK(hipMemcpy(dst, (what == HC_TAP_FACE_MASK ? ctx->d_mask : ctx->d_gray) + (size_t)frame * n, n, hipMemcpyDeviceToHost));
      return HC_OK;
    }
    default:
      return hc::set_err(HC_ERR_ARG, "unknown tap");
  }
}

int hc_context_enable_timing(hc_context* ctx, int enable) {
  if (!ctx) return hc::set_err(HC_ERR_ARG, "null context");
  ctx->timing_enabled = enable != 0;
  return HC_OK;
}

int hc_context_compiled_stages(hc_context* ctx, int* nstages) {
  if (!ctx || !nstages) return hc::set_err(HC_ERR_ARG, "null argument");
  *nstages = ctx->last_compiled;
  return HC_OK;
}

int hc_context_kernel_ms(hc_context* ctx, float ms[HC_NKERNELS]) {
  if (!ctx || !ms) return hc::set_err(HC_ERR_ARG, "null argument");
  if (!ctx->timed_last) return hc::set_err(HC_ERR_STATE, "timing was not enabled for the last batch");
  for (int i = 0; i < HC_NKERNELS; i++) ms[i] = ctx->last_ms[i];
  return HC_OK;
}

int hc_debug_plan_strips(const hc_cascade* c, int rows, int cols, double scale_factor, int min_w, int min_h, int max_w, int max_h,
                         int semantics, int comp_max, int32_t* strips, int cap, int* nstrips, int64_t* lds_bytes) {
  if (!c || !nstrips) return hc::set_err(HC_ERR_ARG, "null argument");
  *nstrips = 0;
  if (lds_bytes) *lds_bytes = 0;
  hc::Plan P;
  hc::Cv24Host cv;
  const int rc = semantics == 1 ? hc::build_plan_cv24(c->model, rows, cols, scale_factor, min_w, min_h, max_w, max_h, 16, 16, P, cv)
                                : hc::build_plan(c->model, rows, cols, scale_factor, min_w, min_h, max_w, max_h, 16, 16, P);
  if (rc != HC_OK) return hc::set_err(rc, "cannot plan this configuration");
  if (P.levels.empty() || (semantics == 1 && !cv.fused_tables)) return HC_OK;
  hc::PrepKnobs k;
  k.strip_comp_max = comp_max;
  hc::StripPlan sp;
  hc::plan_pyramid_strips(P.levels, P.xtab, cols, c->model.win_w - 2, c->model.win_h - 2, semantics, k, sp);
  if (!sp.ok) return HC_OK;
  *nstrips = (int)sp.strips.size();
  if (lds_bytes) *lds_bytes = (int64_t)sp.lds_bytes;
  if ((int)sp.strips.size() > cap || !strips) return hc::set_err(HC_ERR_CAPACITY, "strip buffer too small");
  for (size_t i = 0; i < sp.strips.size(); i++) {
    const hc::DevStrip& T = sp.strips[i];
    const int32_t v[12] = {T.level, T.c0, T.c1, T.ce, T.g0, T.g1, T.sx0, T.sxn, T.left, T.has_right, T.carry_ofs, 0};
    for (int j = 0; j < 12; j++) strips[12 * i + j] = v[j];
  }
  return HC_OK;
}

int hc_context_last_launch(const hc_context* ctx, hc_launch_info* out) {
  if (!ctx || !out) return hc::set_err(HC_ERR_ARG, "null argument");
  if (!ctx->configured) return hc::set_err(HC_ERR_STATE, "context not configured");
  *out = ctx->last_launch;
  return HC_OK;
}

}  // extern "C"
