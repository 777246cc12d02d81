// api_context.cpp -- life of a context: the knobs it is created with, its workspace for one frame geometry and set of call
// parameters (the arguments of the reference's call, src/tracker/detector.cc:25-31), shared prep passes over several
// cascades (BASELINE configs #3 / #5), the switch between the two generations of the call, and the route policy.

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "hc_context.h"

using hc::dev_alloc;
using hc::dev_upload;

namespace hc {

Knobs knobs_from_env() {
  Knobs k;
  const auto geti = [](const char* name, int& v, int lo, int hi) { if (const char* e = getenv(name)) v = std::min(hi, std::max(lo, atoi(e))); };
  const auto getb = [](const char* name, bool& v) { if (const char* e = getenv(name)) v = atoi(e) != 0; };
  geti("HC_SPARSE_MAX", k.sparse_max, 0, 256);        // a wavefront owns at most 64 of them
  geti("HC_HOST_THREADS", k.host_threads, 1, 1 << 20);
  geti("HC_DEBUG_MAX_STAGES", k.debug_max_stages, 0, 1 << 20);
  geti("HC_DEBUG_LDS_PAD1", k.debug_lds_pad[0], 0, 1 << 17);
  geti("HC_DEBUG_LDS_PAD2", k.debug_lds_pad[1], 0, 1 << 17);
  geti("HC_ROW_STAGES", k.row_stages, 1, 1 << 20);
  if (const char* e = getenv("HC_BAND_ROWS1")) k.band_rows[0] = atoi(e);
  if (const char* e = getenv("HC_BAND_ROWS2")) k.band_rows[1] = atoi(e);
  if (const char* e = getenv("HC_BAND_WAVES1")) k.band_waves[0] = atoi(e) == 8 ? 8 : 4;
  if (const char* e = getenv("HC_BAND_WAVES2")) k.band_waves[1] = atoi(e) == 8 ? 8 : 4;
  if (const char* e = getenv("HC_FUSED_PREP")) k.fused_prep = atoi(e) != 0 ? 1 : 0;
  if (const char* e = getenv("HC_FRAME_CHUNK")) k.frame_chunk = std::max(0, atoi(e)) & ~7;
  if (const char* e = getenv("HC_PI_FRAME_CHUNK")) k.pi_frame_chunk = std::max(0, atoi(e)) & ~7;
  if (const char* e = getenv("HC_FUSE_VARIANCE")) { k.fuse_variance = atoi(e) != 0; k.fuse_variance_forced = k.fuse_variance; }
  getb("HC_INT_WEIGHTS", k.allow_int_weights);
  getb("HC_SPEC", k.allow_spec);
  getb("HC_PAIR_FORM", k.pair_form);
  getb("HC_P16", k.allow_p16);
  getb("HC_BANK_ORDER", k.bank_order);
  getb("HC_VAR_RING", k.var_ring);
  getb("HC_VAR_XRING", k.var_xring);
  getb("HC_VAR_SPLIT", k.var_split);
  getb("HC_CV24_FUSED", k.cv24_fused);
  getb("HC_CV24_SEG", k.cv24_seg);
  getb("HC_SHARE_VNF", k.share_vnf);
  getb("HC_BAND_TRIM", k.band_trim);
  getb("HC_SIDE_PRIO", k.side_prio);
  getb("HC_COPY_PRIO", k.copy_prio);
  getb("HC_DEVICE_GROUP", k.device_group);
  geti("HC_GROUP_MIN_BATCH", k.group_min_batch, 1, 1 << 20);
  geti("HC_GROUP_BIN_CAP", k.group_bin_cap, 1, HC_GROUP_BIN);
  geti("HC_GROUP_OUT_CAP", k.group_out_cap, 1, HC_GROUP_OUT);
  if (const char* e = getenv("HC_FUSED_MIN_BATCH")) k.fused_min_batch = atof(e);
  geti("HC_SIDE_MAX_BATCH", k.side_max_batch, 0, 1 << 20);
  geti("HC_SEG_MAX_BATCH", k.seg_max_batch, 0, 1 << 20);
  geti("HC_SEG_TILES", k.seg_tiles, 1, 255);
  geti("HC_STRIP_COMP_MAX", k.strip_comp_max, 0, 1024);
  return k;
}

int sync_cache_size_from_env() {
  const char* e = getenv("HC_SYNC_CONTEXTS");
  return std::min(64, std::max(1, e ? atoi(e) : 4));
}

PrepKnobs prep_knobs(const hc_context* c) {
  PrepKnobs p;
  p.frame_chunk = c->k.pi_frame_chunk; p.var_ring = c->k.var_ring; p.var_xring = c->k.var_xring; p.var_split = c->k.var_split;
  p.strip_comp_max = c->k.strip_comp_max;
  return p;
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
bool fused_kernel_has_rings(const hc_context* c) {
  if (c->use_strips) return true;
  int max_w = 0, max_nx = 0;
  for (const hc::DevLevel& L : c->plan.levels) { max_w = std::max(max_w, L.w); max_nx = std::max(max_nx, L.nx); }
  const hc::Model& m = c->casc->model;
  if (c->semantics == 1) return hc::pyramid_integral_takes_cv24(prep_knobs(c), max_w, max_nx, c->plan.cols, m.win_w - 2, m.win_h - 2);
  return hc::pyramid_integral_has_ring(prep_knobs(c), max_w, c->plan.cols, m.win_w - 2, m.win_h - 2);
}

// Variance factors inside the fused prep kernel wherever it has a ring route for them -- the one-workgroup-per-level kernel for
// levels of at most 1024 columns, the strip kernel for any width.  Also for the leader of a shared pass: members with the
// leader's window read these factors, members with another window run k_variance on the stored integrals.  Where neither
// kernel has rings (an inner rect other than 18 or 22 rows) a separate k_variance launch beats the read-back route
// (HC_FUSE_VARIANCE=1 forces that route, =0 never fuses).
bool wants_fused_variance(const hc_context* c) {
  if (!c->k.fuse_variance || c->leader) return false;
  if (c->k.fuse_variance_forced) return true;
  return fused_kernel_has_rings(c);
}

bool use_fused_prep(const hc_context* c, int nframes, bool fused_variance) {
  if (c->k.fused_prep >= 0) return c->k.fused_prep > 0;                         // HC_FUSED_PREP pins the route
  const hc::Plan& P = c->plan;
  const double equiv = (double)nframes * (double)P.rows * (double)P.cols / (1280.0 * 720.0);
  if (c->k.fused_min_batch >= 0) return equiv > c->k.fused_min_batch;           // HC_FUSED_MIN_BATCH: plain threshold
  if (P.levels.empty()) return false;
  // (the strip form marches down a strip of at most 640 columns, not down the whole level)
  const double A = (double)(c->use_strips ? std::min(P.levels[0].w, c->strips.lds_pitch) : P.levels[0].w) * (double)P.levels[0].h / (795.0 * 447.0);
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
// K6 workspace: bins, level table, output slots
int ensure_group(hc_context* c) {
  if (c->d_gbins) return HC_OK;
  const hc::Plan& P = c->plan;
  std::vector<hc::GroupLevel> gl(P.levels.size());
  for (size_t i = 0; i < gl.size(); i++) {
    const hc::DevLevel& L = P.levels[i];
    gl[i] = hc::GroupLevel{L.scale, L.ystep, L.win_w, L.win_h, c->semantics == 1 ? c->cv.lv[i].factor : (double)L.scale};
  }
  int rc;
  if ((rc = dev_upload(&c->d_glevels, gl))) return rc;
  if ((rc = host_alloc_coherent((void**)&c->h_gboxes, (size_t)c->max_batch * HC_GROUP_OUT * sizeof(hc::GroupBox)))) return rc;
  if ((rc = host_alloc_coherent((void**)&c->h_gcount, (size_t)c->max_batch * sizeof(unsigned int)))) return rc;
  return dev_alloc(&c->d_gbins, (size_t)c->max_batch * HC_GROUP_BIN);
}

}  // namespace hc

void hc_context::free_workspace() {
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
                  d_count, d_ticket, d_strips, d_carry, d_strip_progress, d_area_xofs, d_area_xsi, d_area_xal, d_area_yofs, d_area_ysi, d_area_yal};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  d_frames = d_color = d_pyr = nullptr; color_bytes = 0;
  d_area_xofs = d_area_xsi = d_area_yofs = d_area_ysi = nullptr; d_area_xal = d_area_yal = nullptr; area_general = false; d_sum = d_sq = nullptr; d_vnf = nullptr; d_taps = nullptr; d_levels = nullptr;
  d_xtab = d_ytab = nullptr;
  d_resize_items = d_row_items = d_col_items = d_var_items = d_band_items[0] = d_band_items[1] = nullptr;
  d_seg_items[0] = d_seg_items[1] = nullptr;
  d_cand = nullptr; d_count = nullptr; d_ticket = nullptr;
  if (d_gbins) (void)hipFree(d_gbins);
  if (d_glevels) (void)hipFree(d_glevels);
  if (h_gboxes) (void)hipHostFree(h_gboxes);
  if (h_gcount) (void)hipHostFree(h_gcount);
  d_gbins = nullptr; d_glevels = nullptr; h_gboxes = nullptr; h_gcount = nullptr; group_ran = false;
  d_strips = nullptr; d_carry = nullptr; d_strip_progress = nullptr; use_strips = false; strips = hc::StripPlan{};
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

extern "C" {

int hc_context_create(const hc_cascade* c, int device, void* stream, hc_context** out) {
  if (!c || !out) return hc::set_err(HC_ERR_ARG, "null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return hc::set_err(HC_ERR_NO_DEVICE, "no HIP device available (this backend has no CPU fallback)");
  if (device < 0 || device >= ndev) return hc::set_err(HC_ERR_ARG, "device index out of range");
  HIPCHK(hipSetDevice(device));
  hc_context* ctx = new hc_context;
  ctx->casc = c;
  ctx->device = device;
  ctx->k = hc::knobs_from_env();       // every tuning variable, once: nothing below reads the environment again
  ctx->device_group = ctx->k.device_group;
  if (stream) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
  else {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return hc::set_err(HC_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    ctx->own_stream = true;
  }
  for (auto& e : ctx->ev) {
    if (hipEventCreate(&e) != hipSuccess) { hc_context_destroy(ctx); return hc::set_err(HC_ERR_HIP, "hipEventCreate failed"); }
  }
  // (the side stream of the small-batch route is made on first use: every stream alive takes a share of the hardware queues)
  // copy_stream (the skin filter's second round trip: box sums of the grouped rectangles while the next batch already runs)
  // is a HIGH-priority stream: the runtime keeps those on hardware queues of their own, so its few microseconds of work
  // neither queue behind the next batch's kernels nor wait for them to drain
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  if (!ctx->k.copy_prio) prio_greatest = prio_least;
  if (hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, prio_greatest) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_copied, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_prep, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_var, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming) != hipSuccess) {
    hc_context_destroy(ctx);
    return hc::set_err(HC_ERR_HIP, "cannot create the result stream / event");
  }
  for (int k = 0; k < 2; k++)
    if (ctx->k.band_rows[k] != 8 && ctx->k.band_rows[k] != 16 && ctx->k.band_rows[k] != 32) {
      delete ctx;
      return hc::set_err(HC_ERR_ARG, "band rows must be 8, 16 or 32");
    }
  int rc = hc::upload_model(ctx);
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
  hc::free_model_tables(ctx);
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
  if (!ctx) return hc::set_err(HC_ERR_ARG, "null context");
  if (!(scale_factor > 1.0)) return hc::set_err(HC_ERR_ARG, "scaleFactor must be > 1");   // CV_Assert upstream
  if (rows <= 0 || cols <= 0 || max_batch <= 0 || max_batch > 65535) return hc::set_err(HC_ERR_ARG, "bad geometry / batch");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->free_workspace();
  hc::Plan& P = ctx->plan;
  const bool cv24 = ctx->semantics == 1;
  if (cv24 && (prep_scales || leader)) return hc::set_err(HC_ERR_UNSUPPORTED, "shared prep passes exist for the OpenCV >= 4 semantics only");
  if (cv24 && (ctx->k.band_rows[0] != 16 || ctx->k.band_rows[1] != 16)) return hc::set_err(HC_ERR_UNSUPPORTED, "2.4 semantics: 16-row bands only");
  int rc = cv24 ? hc::build_plan_cv24(ctx->casc->model, rows, cols, scale_factor, min_w, min_h, max_w, max_h, ctx->band_step[0],
                                      ctx->band_step[1], P, ctx->cv)
                : hc::build_plan(ctx->casc->model, rows, cols, scale_factor, min_w, min_h, max_w, max_h, ctx->band_step[0],
                                 ctx->band_step[1], P, prep_scales);
  if (rc != HC_OK) return hc::set_err(rc, "cannot plan this configuration");
  for (const hc::DevLevel& L : P.levels)
    if (L.nx > 65535 || L.ny_all > 65535 || P.levels.size() > 255) return hc::set_err(HC_ERR_UNSUPPORTED, "frame too large");
  ctx->max_batch = max_batch;
  ctx->min_neighbors = min_neighbors;
  ctx->cand_per_frame = cand_capacity > 0 ? cand_capacity : 4096;
  const size_t B = (size_t)max_batch;
  if (!leader && !P.levels.empty() && ctx->k.var_split && (!cv24 || ctx->cv.fused_tables)) {
    // The fused prep kernel runs in its column-strip form wherever a strip plan exists for the geometry (any frame width;
    // inner rects of 18 or 22 rows, i.e. the 20 x 20 and 24 x 24 windows of the stock cascades); HC_VAR_SPLIT=0 keeps the
    // one-workgroup-per-level kernel.
    const hc::Model& m0 = ctx->casc->model;
    hc::plan_pyramid_strips(P.levels, P.xtab, cols, m0.win_w - 2, m0.win_h - 2, ctx->semantics, hc::prep_knobs(ctx), ctx->strips);
    ctx->use_strips = ctx->strips.ok;
    if (ctx->use_strips) {
      if ((rc = dev_upload(&ctx->d_strips, ctx->strips.strips))) return rc;
      if ((rc = dev_alloc(&ctx->d_carry, B * ctx->strips.carry_words))) return rc;
      if ((rc = dev_alloc(&ctx->d_strip_progress, B * ctx->strips.strips.size() + 1))) return rc;
    }
  }
  if (leader) {
    if (leader->plan.int_words != P.int_words || leader->max_batch != max_batch || leader->device != ctx->device)
      return hc::set_err(HC_ERR_STATE, "shared prep: integral layouts of leader and follower differ");
    ctx->d_sum = leader->d_sum;
    ctx->d_sq = leader->d_sq;
    ctx->leader = leader;
    leader->followers.push_back(ctx);
    ctx->seen_seq = leader->prep_seq;
    ctx->scan_in_flight = false;
    // Same window and the same grid as the leader's (alt, alt2 and alt_tree are all 20 x 20): the variance factors are the
    // same numbers, so the leader's k_variance serves this member too (HC_SHARE_VNF=0 gives every member its own).
    const bool share_ok = ctx->k.share_vnf;
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
  if ((rc = dev_alloc(&ctx->d_count, 1 + B))) return rc;   // the batch's candidate count, then one count per frame (K6 bins)
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
      for (int t0 = 0; t0 < ntiles; t0 += ctx->k.seg_tiles)
        seg.push_back(hc::WorkItem{it.level, it.a, (t0 * hc::TILE_POS) | (ctx->k.seg_tiles << 16)});
    }
    ctx->n_seg_items[k] = (int)seg.size();
    if ((rc = dev_upload(&ctx->d_seg_items[k], seg))) return rc;
  }
  if ((rc = dev_alloc(&ctx->d_ticket, 1))) return rc;
  HIPCHK(hipMemset(ctx->d_ticket, 0, sizeof(unsigned int)));
  if ((rc = hc::host_alloc_coherent((void**)&ctx->h_cand, ctx->cand_cap * sizeof(unsigned long long)))) return rc;
  if ((rc = hc::host_alloc_coherent((void**)&ctx->h_pub, 4 * sizeof(unsigned int)))) return rc;
  ctx->h_pub[0] = ctx->h_pub[1] = ctx->h_pub[2] = ctx->h_pub[3] = 0;
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
  if (!ctxs || n < 1) return hc::set_err(HC_ERR_ARG, "hc_context_configure_shared: no contexts");
  for (int i = 0; i < n; i++) {
    if (!ctxs[i]) return hc::set_err(HC_ERR_ARG, "null context");
    if (ctxs[i]->device != ctxs[0]->device) return hc::set_err(HC_ERR_ARG, "shared prep: contexts must live on one device");
    for (int j = 0; j < i; j++)
      if (ctxs[i] == ctxs[j]) return hc::set_err(HC_ERR_ARG, "shared prep: the same context twice");
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

void* hc_context_stream(const hc_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int hc_context_set_rerun_on_overflow(hc_context* ctx, int allow) {
  if (!ctx) return hc::set_err(HC_ERR_ARG, "null context");
  ctx->rerun_caller_frames = allow != 0;
  return HC_OK;
}

int hc_context_set_device_grouping(hc_context* ctx, int on) {
  if (!ctx) return hc::set_err(HC_ERR_ARG, "null context");
  if (ctx->pending) return hc::set_err(HC_ERR_STATE, "a batch is in flight");
  ctx->device_group = on != 0;
  return HC_OK;
}

int hc_context_group_host_frames(const hc_context* ctx) { return ctx ? ctx->last_group_host_frames : 0; }

int hc_context_set_semantics(hc_context* ctx, int semantics) {
  if (!ctx) return hc::set_err(HC_ERR_ARG, "null context");
  if (semantics != 0 && semantics != 1) return hc::set_err(HC_ERR_ARG, "semantics: 0 (OpenCV >= 4) or 1 (OpenCV 2.4)");
  if (ctx->pending) return hc::set_err(HC_ERR_STATE, "a batch is in flight");
  if (semantics == ctx->semantics) return HC_OK;
  const hc::Model& m = ctx->casc->model;
  if (semantics == 1) {
    if (m.has_tilted) return hc::set_err(HC_ERR_UNSUPPORTED, "tilted features");
    for (const hc::ModelNode& n : m.nodes)
      for (int k = 0; k < 3; k++)
        for (int j = 0; j < 4; j++)
          if (n.rect[k][j] < 0) return hc::set_err(HC_ERR_UNSUPPORTED, "rect coordinate out of range");
  }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->free_workspace();               // geometry, buffers and node tables all depend on the generation
  ctx->semantics = semantics;
  return hc::upload_model(ctx);
}

}  // extern "C"
