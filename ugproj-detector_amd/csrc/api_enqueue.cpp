// api_enqueue.cpp -- one batch of frames into the kernels: the body of cv::CascadeClassifier::detectMultiScale as the
// reference calls it (src/tracker/detector.cc:25-31), asynchronous and batched.
//
// A batch takes exactly one PREP route (pyramid + integrals [+ variance factors]):
//   prep_fused            OpenCV >= 4, k_pyramid_integral: one workgroup marching down each (level, frame)
//   prep_parallel         OpenCV >= 4, k_resize + k_integral_rows + k_integral_cols (small batches)
//   prep_shared_follower  member of a shared pass: waits for the leader's kernels, launches nothing
//   prep_cv24_fused       OpenCV 2.4 arithmetic in the same fused kernel
//   prep_cv24_parallel    OpenCV 2.4, four separate kernels
// then the variance kernels the route left open (variance_after_prep), the cascade launches (launch_cascades: one per
// scan-step class, the larger levels first) and the publication of the results (publish_results).  What was chosen is
// recorded in hc_context::last_launch (hc_context_last_launch).

#include <algorithm>
#include <cmath>

#include "hc_context.h"

namespace hc {

// A group that shares one prep pass is enqueued leader first, then every follower once, with the same batch.
int shared_enqueue_check(hc_context* c, const void* frames, int nframes) {
  if (c->leader) {
    hc_context* L = c->leader;
    if (L->prep_seq == c->seen_seq) return set_err(HC_ERR_STATE, "shared prep: enqueue the leader's batch first (once per follower scan)");
    if (L->prep_frames != nframes) return set_err(HC_ERR_ARG, "shared prep: follower batch size differs from the leader's");
    // A follower scans whatever the leader's LAST prep pass left in the integrals.  The sequence numbers name the batch: a
    // leader that was enqueued again before this follower scanned the previous batch would make the follower return the
    // newer batch's rectangles for the older frames -- also when both batches came from one recycled buffer, which a
    // pointer comparison cannot see.  The refusal resynchronises: the skipped batch is lost to this member, the next
    // enqueue scans the leader's current one.
    if (L->prep_seq != c->seen_seq + 1) {
      const unsigned long long missed = L->prep_seq - c->seen_seq - 1;
      c->seen_seq = L->prep_seq - 1;
      return set_err(HC_ERR_STATE, "shared prep: the leader has been enqueued " + std::to_string(missed + 1) + " times since this "
                                   "follower's last scan; its integrals now hold the newest batch (per batch: the leader first, "
                                   "then every follower once).  Enqueue the follower again to scan that batch");
    }
    if (L->prep_src != frames)       // an extra guard for callers that do pass distinct buffers
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

namespace {

struct Batch {
  const uint8_t* frames;
  size_t frame_stride, row_step;
  int nframes;
  bool timed;
  double equiv;            // batch size in 720p-frame equivalents: decides the small-batch policies
  size_t pyr_stride, int_stride, grid_stride;
  bool variance_done = false;   // the prep route has produced (or will receive) this context's variance factors
};

int hip_fail(const char* what, int e) { return set_err(HC_ERR_HIP, std::string(what) + " launch: " + hipGetErrorString((hipError_t)e)); }

void level_extent(const Plan& P, int& max_w, int& max_nx) {
  max_w = max_nx = 0;
  for (const DevLevel& L : P.levels) { max_w = std::max(max_w, L.w); max_nx = std::max(max_nx, L.nx); }
}

void note_prep(hc_context* c, int route, const PrepLaunchInfo* pi) {
  hc_launch_info& I = c->last_launch;
  I.prep_route = route;
  if (pi) {
    I.pi_frame_chunk = pi->frame_chunk; I.pi_remapped = pi->remapped; I.pi_tail_frames = pi->tail_frames;
    I.sq_stored = pi->sq_stored ? 1 : 0;
    if (pi->route) I.variance_route = pi->route;
  }
}

// Whether a 2.4 batch goes through the fused kernel: the geometry fits its ring route, every grid column has a ring, and the
// batch is large enough for one workgroup per (level, frame) (HC_CV24_FUSED=0 pins the four separate kernels).
bool cv24_takes_fused(const hc_context* c, int nframes) {
  return c->k.cv24_fused && c->cv.fused_tables && (c->cv_chain || c->cv_tree_tiled) && use_fused_prep(c, nframes, true) &&
         fused_kernel_has_rings(c);
}

// the fused prep kernel in the form this context was configured for: one workgroup per (level, frame), or per column strip
int launch_fused_prep(hc_context* c, const Batch& b, float* vnf, bool store_sq, int semantics, PrepLaunchInfo* pi) {
  const Plan& P = c->plan;
  const Model& m = c->casc->model;
  uint8_t* pyr = c->taps_enabled ? c->d_pyr : nullptr;
  c->strips_ran = c->use_strips;
  if (c->use_strips)
    return launch_pyramid_strips(c->stream, b.nframes, c->strips, c->d_strips, b.frames, b.frame_stride, b.row_step, P.rows, P.cols,
                                 c->d_levels, c->d_xtab, c->d_ytab, c->d_sum, c->d_sq, b.int_stride, pyr, b.pyr_stride, prep_knobs(c), vnf,
                                 b.grid_stride, store_sq, c->d_carry, c->d_strip_progress, pi, semantics);
  int max_w, max_nx;
  level_extent(P, max_w, max_nx);
  return launch_pyramid_integral(c->stream, b.nframes, (int)P.levels.size(), max_w, b.frames, b.frame_stride, b.row_step, P.rows, P.cols,
                                 c->d_levels, c->d_xtab, c->d_ytab, c->d_sum, c->d_sq, b.int_stride, pyr, b.pyr_stride, prep_knobs(c), vnf,
                                 b.grid_stride, m.win_w - 2, m.win_h - 2, max_nx, store_sq, pi, semantics);
}

// ---- prep routes -------------------------------------------------------------------------------------------------

// OpenCV 2.4 through the fused prep kernel: cvResize arithmetic, integrals and the f64 variance factors in one march,
// nothing materialised in between.
int prep_cv24_fused(hc_context* c, Batch& b) {
  PrepLaunchInfo pi;
  if (int e = launch_fused_prep(c, b, (float*)c->d_vnf64, c->taps_enabled, 1, &pi)) return hip_fail("fused prep kernel (2.4)", e);
  c->sq_valid = pi.sq_stored;
  if (b.timed) { HIPCHK(hipEventRecord(c->ev[1], c->stream)); HIPCHK(hipEventRecord(c->ev[2], c->stream)); }
  b.variance_done = true;
  note_prep(c, HC_PREP_CV24_FUSED, &pi);
  return HC_OK;
}

// OpenCV 2.4: cvResize pyramid (materialised), the two-pass integrals; its f64 variance kernel follows
int prep_cv24_parallel(hc_context* c, Batch& b) {
  const Plan& P = c->plan;
  hipStream_t s = c->stream;
  int e;
  if ((e = launch_cv24_resize(s, b.nframes, (int)P.levels.size(), c->cv.max_pix, b.frames, b.frame_stride, b.row_step, P.rows, P.cols,
                              c->d_cvlevels, c->d_cvtab_ofs, c->d_cvtab_a0, c->d_cvtab_a1, c->d_pyr, b.pyr_stride)))
    return hip_fail("k_cv24_resize", e);
  if (b.timed) HIPCHK(hipEventRecord(c->ev[1], s));
  if ((e = launch_integral_rows(s, b.nframes, (int)P.row_items.size(), c->d_pyr, b.pyr_stride, c->d_levels, c->d_row_items, c->d_sum,
                                c->d_sq, b.int_stride)))
    return hip_fail("k_integral_rows", e);
  if (b.timed) HIPCHK(hipEventRecord(c->ev[2], s));
  if ((e = launch_integral_cols(s, b.nframes, (int)P.col_items.size(), c->d_levels, c->d_col_items, c->d_sum, c->d_sq, b.int_stride)))
    return hip_fail("k_integral_cols", e);
  b.variance_done = true;            // the generic k_variance is the >= 4 gate; 2.4 has its own factor (variance_after_prep)
  note_prep(c, HC_PREP_CV24_PARALLEL, nullptr);
  c->last_launch.variance_route = HC_VAR_CV24_KERNEL;
  return HC_OK;
}

// shared prep: the leader's kernels have produced (or are producing) the integrals of this very batch
int prep_shared_follower(hc_context* c, Batch& b) {
  hipStream_t s = c->stream;
  HIPCHK(hipStreamWaitEvent(s, c->leader->ev_prep, 0));
  if (c->vnf_shared || c->vnf_from_leader) {      // ... and this member's variance factors
    HIPCHK(hipStreamWaitEvent(s, c->leader->ev_var, 0));
    b.variance_done = true;
    c->last_launch.variance_route = HC_VAR_LEADER;
  }
  if (b.timed) { HIPCHK(hipEventRecord(c->ev[1], s)); HIPCHK(hipEventRecord(c->ev[2], s)); }
  note_prep(c, HC_PREP_SHARED_FOLLOWER, nullptr);
  return HC_OK;
}

// The fused kernel.  A context on its own gets its variance factors from the same kernel (no k_variance launch, no second
// pass over the integrals); a shared prep pass serves cascades with different windows: the leader's factors from the
// kernel's rings, one follower window of another height from a second set of rings, the rest by k_variance.  The pyramid
// bytes only exist as a debug tap in this mode.
int prep_fused(hc_context* c, Batch& b) {
  const bool fused_variance = wants_fused_variance(c);
  // the squares' integral leaves the chip only for somebody who reads it: a debug tap, a k_variance launch of this context
  // (no ring route) or of a member of the shared pass whose window differs from the leader's
  bool store_sq = c->taps_enabled || !fused_variance;
  for (const hc_context* f : c->followers) store_sq = store_sq || !f->vnf_shared;
  PrepLaunchInfo pi;
  if (int e = launch_fused_prep(c, b, fused_variance ? c->d_vnf : nullptr, store_sq, 0, &pi)) return hip_fail("fused prep kernel", e);
  c->sq_valid = pi.sq_stored;
  if (b.timed) { HIPCHK(hipEventRecord(c->ev[1], c->stream)); HIPCHK(hipEventRecord(c->ev[2], c->stream)); }
  b.variance_done = fused_variance;
  note_prep(c, HC_PREP_FUSED, &pi);
  return HC_OK;
}

// The three fully parallel kernels: no march, no floor -- 0.07 instead of 0.39 ms for ONE 720p frame.
int prep_parallel(hc_context* c, Batch& b) {
  const Plan& P = c->plan;
  hipStream_t s = c->stream;
  int e;
  if ((e = launch_resize(s, b.nframes, (int)P.resize_items.size(), b.frames, b.frame_stride, b.row_step, P.rows, P.cols, c->d_levels,
                         c->d_xtab, c->d_ytab, c->d_resize_items, c->d_pyr, b.pyr_stride)))
    return hip_fail("k_resize", e);
  if (b.timed) HIPCHK(hipEventRecord(c->ev[1], s));
  if ((e = launch_integral_rows(s, b.nframes, (int)P.row_items.size(), c->d_pyr, b.pyr_stride, c->d_levels, c->d_row_items, c->d_sum,
                                c->d_sq, b.int_stride)))
    return hip_fail("k_integral_rows", e);
  if (b.timed) HIPCHK(hipEventRecord(c->ev[2], s));
  if ((e = launch_integral_cols(s, b.nframes, (int)P.col_items.size(), c->d_levels, c->d_col_items, c->d_sum, c->d_sq, b.int_stride)))
    return hip_fail("k_integral_cols", e);
  note_prep(c, HC_PREP_PARALLEL, nullptr);
  return HC_OK;
}

// the variance factors no prep route has produced: k_variance (>= 4) or the f64 kernel of the 2.4 parallel route
int variance_after_prep(hc_context* c, const Batch& b, bool cv24_parallel) {
  const Plan& P = c->plan;
  const Model& m = c->casc->model;
  hipStream_t s = c->stream;
  int e;
  if (!b.variance_done) {
    if ((e = launch_variance(s, b.nframes, (int)P.var_items.size(), c->d_levels, c->d_var_items, c->d_sum, c->d_sq, b.int_stride,
                             m.win_w - 2, m.win_h - 2, c->d_vnf, b.grid_stride)))
      return hip_fail("k_variance", e);
    c->last_launch.variance_route = HC_VAR_KERNEL;
  }
  if (!c->followers.empty()) HIPCHK(hipEventRecord(c->ev_var, s));
  if (cv24_parallel &&
      (e = launch_cv24_variance(s, b.nframes, (int)P.var_items.size(), c->d_levels, c->d_var_items, c->d_sum, c->d_sq, b.int_stride,
                                m.win_w - 2, m.win_h - 2, c->cv.inv_area, c->d_vnf64, b.grid_stride)))
    return hip_fail("k_cv24_variance", e);
  return HC_OK;
}

// ---- cascade launches ----------------------------------------------------------------------------------------------

// a real stage tree of a cascade the tiled tree kernel does not take: window-at-a-time evaluation that walks parent / next / child
int launch_cv24_window_kernel(hc_context* c, const Batch& b) {
  const Plan& P = c->plan;
  const Model& m = c->casc->model;
  Cv24Args T{};
  T.levels = c->d_cvlevels; T.sum = c->d_sum; T.sq = c->d_sq; T.int_stride = b.int_stride; T.grid_stride = b.grid_stride;
  T.stages = c->d_cvstages; T.trees = c->d_cvtrees; T.nodes = c->d_cvnodes; T.leaves = c->d_leaves; T.inv_area = c->cv.inv_area;
  T.win_w = m.win_w; T.win_h = m.win_h; T.cand = c->d_cand; T.cand_count = c->d_count; T.cand_cap = (unsigned int)c->cand_cap;
  T.taps = c->taps_enabled ? c->d_taps : nullptr;
  if (int e = launch_cv24_cascade(c->stream, b.nframes, (int)P.levels.size(), c->cv.max_pos, T)) return hip_fail("k_cv24_cascade", e);
  c->last_launch.variant[0] = c->last_launch.variant[1] = HC_VARIANT_CV24_WINDOW;
  return HC_OK;
}

int launch_cascades(hc_context* c, const Batch& b) {
  const Plan& P = c->plan;
  const Model& m = c->casc->model;
  hipStream_t s = c->stream;
  hc_launch_info& I = c->last_launch;
  // With one or two frames neither cascade launch fills 256 CUs and a band walk is a serial chain of tiles:
  // the two launches (independent: different levels) then run side by side on two streams.
  const bool side = b.equiv <= c->k.side_max_batch && !b.timed && !P.band_items[0].empty() && !P.band_items[1].empty();
  if (side) {
    if (!c->side_stream) {
      // a HIGH-priority stream: the runtime keeps those on hardware queues apart from the normal ones, so the side
      // launch can never land on the queue of `stream` itself (where "side by side" would be one after the other plus
      // two event hops: 0.33 instead of 0.28 ms for a 720p frame, depending on how many streams the process had made)
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      if (!c->k.side_prio) hi = lo;
      HIPCHK(hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, hi));
    }
    HIPCHK(hipEventRecord(c->ev_fork, s));
    HIPCHK(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
  }
  I.side_stream = side ? 1 : 0;
  int launches = 0;
  c->last_compiled = 0;
  for (int k = 1; k >= 0; k--) {      // ystep-2 levels (the largest) first
    CascadeArgs A{};
    A.sum = c->d_sum; A.int_stride = b.int_stride; A.vnf = c->d_vnf; A.vnf64 = c->d_vnf64; A.grid_stride = b.grid_stride;
    const bool seg = c->semantics == 0 && b.equiv <= c->k.seg_max_batch && c->n_seg_items[k] > 0;
    A.levels = c->d_levels;
    A.bands = seg ? c->d_seg_items[k] : c->d_band_items[k];
    A.nbands = seg ? c->n_seg_items[k] : (int)P.band_items[k].size();
    A.band_step = c->band_step[k];
    I.band_step[k] = c->band_step[k];
    I.lds_bytes[k] = (int32_t)(k == 0 && c->p16 && !seg && c->semantics == 0 && c->k.debug_max_stages == 0 ? c->lds_bytes_p16 : c->lds_bytes[k]);
    A.nodes = c->d_nodes[k]; A.nodes_sparse = c->d_nodes_sparse[k]; A.stages = c->d_stages; A.kinds = c->d_kinds; A.nstages = (int)m.stages.size();
    const bool p16 = k == 0 && c->p16 && !seg && c->semantics == 0 && c->k.debug_max_stages == 0;
    if (p16) { A.nodes = c->d_nodes_p16; A.nodes_sparse = c->d_nodes_sparse_p16; A.kinds = c->d_kinds_p16; }
    if (c->k.debug_max_stages > 0) A.nstages = std::min(A.nstages, c->k.debug_max_stages);   // timing experiments only (HC_DEBUG_MAX_STAGES)
    A.trees = c->d_trees; A.leaves = c->d_leaves; A.g = p16 ? c->geom_p16 : c->geom[k]; A.tile_words = p16 ? c->tile_words_p16 : c->tile_words[k];
    A.sparse_max = c->k.sparse_max;
    A.frame_chunk = c->k.frame_chunk;
    A.row_stages = c->k.row_stages;
    A.cand = c->d_cand; A.cand_count = c->d_count;
    A.cand_cap = (unsigned int)c->cand_cap;
    if (A.nstages < (int)m.stages.size()) A.cand_cap = 0;   // truncated cascade (timing experiments): emit nothing
    A.taps = c->taps_enabled ? c->d_taps : nullptr;
    A.int_weights = c->int_weights ? 1 : 0;
    A.spec = p16 ? c->spec_id_p16 : c->spec_id;
    A.tree_pass = c->d_tree_pass; A.tree_fail = c->d_tree_fail; A.tree_order = c->d_tree_order;
    hipStream_t ks = side && k == 1 ? c->side_stream : s;
    int compiled = 0, variant = HC_VARIANT_NONE, e;
    if (c->semantics == 1 && !c->cv_chain && c->cv_seg) {
      // the stage tree as alternative chains on the chain kernel: stage tables in evaluation order, segment bounds
      A.stages = c->d_stages_seg; A.kinds = c->d_kinds_seg;
      A.nseg = c->seg_n; A.spine_len = c->seg_spine;
      for (int g = 0; g < 4; g++) A.seg_end[g] = c->seg_end[g];
      e = launch_cascade_cv24(ks, b.nframes, false, c->k.band_rows[k], c->k.band_waves[k], c->lds_bytes[k], A, &compiled);
      variant = HC_VARIANT_CV24_CHAINS;
    } else if (c->semantics == 1 && !c->cv_chain) {
      e = launch_cv24_tree(ks, b.nframes, c->k.band_waves[k], c->lds_bytes[k], A);
      variant = HC_VARIANT_CV24_TREE;
    } else if (c->semantics == 1) {
      e = launch_cascade_cv24(ks, b.nframes, !m.is_stump(), c->k.band_rows[k], c->k.band_waves[k], c->lds_bytes[k], A, &compiled);
      variant = HC_VARIANT_CV24;
    } else {
      e = launch_cascade(ks, b.nframes, !m.is_stump(), seg, c->k.band_rows[k], c->k.band_waves[k],
                         (p16 ? c->lds_bytes_p16 : c->lds_bytes[k]) + (size_t)c->k.debug_lds_pad[k], A, &compiled, &variant);
    }
    if (e) return hip_fail("k_cascade", e);
    if (A.nbands > 0) {
      c->last_compiled = launches++ == 0 ? compiled : std::min(c->last_compiled, compiled);
      I.variant[k] = variant;
      I.compiled_stages[k] = compiled;
      I.segments = seg ? 1 : 0;
      I.casc_frame_chunk = A.frame_chunk;
      I.casc_remapped = A.frame_chunk > 0 && b.nframes > A.frame_chunk ? 1 : 0;
      I.casc_tail_frames = I.casc_remapped ? b.nframes % A.frame_chunk : 0;
    }
  }
  if (side) {
    HIPCHK(hipEventRecord(c->ev_join, c->side_stream));
    HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
  }
  return HC_OK;
}

// Results: the batch's own last kernel stores the candidates and {count, sequence number} in page-locked host memory;
// the host polls the sequence word.  No copy command, no wait on a stream or event of the runtime: those return when
// the hardware queue they were mapped to has drained, and with several contexts alive that queue also carries another
// context's pending work -- a software pipeline then runs one batch deep (see hc_context::ev_copied).
int publish_results(hc_context* c) {
  // (the strip kernel's error word sits behind the row counts of the batch it just ran with)
  const unsigned int* strip_err = c->last_launch.variance_route == HC_VAR_RING_STRIPS || c->strips_ran
                                      ? c->d_strip_progress + (size_t)c->last_launch.nframes * c->strips.strips.size() : nullptr;
  if (c->group_ran) {
    if (int e = launch_group(c->stream, c->last_launch.nframes, c->d_count, c->d_cand, (unsigned int)std::min<size_t>(c->cand_cap, 0xffffffffu),
                             c->d_count + 1, c->d_gbins, c->d_glevels, c->semantics, group_threshold(c), c->h_gboxes, c->h_gcount,
                             c->k.group_bin_cap, c->k.group_out_cap))
      return hip_fail("k_group", e);
    c->last_launch.device_grouping = 1;
  }
  if (int e = launch_publish(c->stream, c->d_count, c->d_cand, (unsigned int)std::min<size_t>(c->cand_cap, 0xffffffffu), c->h_cand,
                             c->h_pub, ++c->pub_seq, c->d_ticket, strip_err))
    return hip_fail("k_publish", e);
  HIPCHK(hipEventRecord(c->ev_done, c->stream));
  return HC_OK;
}

}  // namespace

int enqueue_device(hc_context* c, const uint8_t* d_frames, size_t frame_stride, size_t row_step, int nframes) {
  const Plan& P = c->plan;
  hipStream_t s = c->stream;
  c->inflight_src = d_frames; c->inflight_frame_stride = frame_stride; c->inflight_row_step = row_step;
  c->last_launch = hc_launch_info{};
  c->last_launch.nframes = nframes;
  c->strips_ran = false;
  for (hc_context* f : c->followers)   // the next prep pass overwrites the integrals: the followers' scans of the last one come first
    if (f->scan_in_flight) { HIPCHK(hipStreamWaitEvent(s, f->ev_done, 0)); f->scan_in_flight = false; }
  Batch b{};
  b.frames = d_frames; b.frame_stride = frame_stride; b.row_step = row_step; b.nframes = nframes;
  b.timed = c->timing_enabled;
  b.equiv = (double)nframes * (double)P.rows * (double)P.cols / (1280.0 * 720.0);
  b.pyr_stride = (size_t)P.pyr_bytes; b.int_stride = (size_t)P.int_words; b.grid_stride = (size_t)P.grid_elems;
  c->timed_last = b.timed;
  // K6: the batch groups its own rectangles unless the raw ones are the result (minNeighbors) or a debug build truncates the cascade
  // (bins are max_batch x 8 KB: contexts configured for more than 8192 frames per batch keep the host grouping)
  c->group_ran = c->device_group && nframes >= c->k.group_min_batch && group_threshold(c) > 0 && !P.levels.empty() &&
                 c->k.debug_max_stages == 0 && c->max_batch <= 8192;
  if (c->group_ran)
    if (int rc = ensure_group(c)) return rc;
  HIPCHK(hipMemsetAsync(c->d_count, 0, sizeof(unsigned int) * (c->group_ran ? 1 + (size_t)nframes : 1), s));
  if (!P.levels.empty()) {
    const bool cv24 = c->semantics == 1;
    const bool cv24_fused = cv24 && cv24_takes_fused(c, nframes);
    const bool fused = !cv24 && !c->leader && use_fused_prep(c, nframes, wants_fused_variance(c));
    int rc;
    // first-use allocations: the pyramid exists for the unfused kernels, the 2.4 route and the debug tap only
    if (!c->leader && (c->taps_enabled || (!fused && !cv24_fused)) && (rc = ensure_pyr(c))) return rc;
    if (c->taps_enabled) {
      if ((rc = ensure_taps(c))) return rc;
      HIPCHK(hipMemsetAsync(c->d_taps, 0xFE, b.grid_stride * nframes, s));
    }
    c->sq_valid = true;
    if (b.timed) HIPCHK(hipEventRecord(c->ev[0], s));
    if (cv24_fused) rc = prep_cv24_fused(c, b);
    else if (cv24) rc = prep_cv24_parallel(c, b);
    else if (c->leader) rc = prep_shared_follower(c, b);
    else if (fused) rc = prep_fused(c, b);
    else rc = prep_parallel(c, b);
    if (rc) return rc;
    if (!c->followers.empty()) HIPCHK(hipEventRecord(c->ev_prep, s));
    if (b.timed) HIPCHK(hipEventRecord(c->ev[3], s));
    if ((rc = variance_after_prep(c, b, cv24 && !cv24_fused))) return rc;
    if (b.timed) HIPCHK(hipEventRecord(c->ev[4], s));
    if (cv24 && !c->cv_chain && !c->cv_tree_tiled) rc = launch_cv24_window_kernel(c, b);
    else rc = launch_cascades(c, b);
    if (rc) return rc;
    if (b.timed) HIPCHK(hipEventRecord(c->ev[5], s));
  }
  if (int rc = publish_results(c)) return rc;
  if (c->leader) c->scan_in_flight = true;
  c->pending = nframes;
  return HC_OK;
}

}  // namespace hc

// =============================================================================== C ABI
extern "C" {

int hc_detect_batch_enqueue(hc_context* ctx, const uint8_t* d_frames, size_t frame_stride_bytes, size_t row_step_bytes,
                            int nframes) {
  if (!ctx || !ctx->configured) return hc::set_err(HC_ERR_STATE, "context not configured");
  if (!d_frames || nframes <= 0 || nframes > ctx->max_batch) return hc::set_err(HC_ERR_ARG, "bad frame pointer / batch size");
  if (row_step_bytes < (size_t)ctx->plan.cols) return hc::set_err(HC_ERR_ARG, "row step smaller than the row");
  if (ctx->pending) return hc::set_err(HC_ERR_STATE, "previous batch not waited");
  HIPCHK(hipSetDevice(ctx->device));
  if (int rc = hc::shared_enqueue_check(ctx, d_frames, nframes)) return rc;
  return hc::enqueue_device(ctx, d_frames, frame_stride_bytes, row_step_bytes, nframes);
}

// Test entry of the result side alone: nkeys raw candidate keys (frame << 40 | level << 32 | gy << 16 | gx, any order) take the
// place of a batch's candidates and go through grouping + publication exactly as the cascade kernels' own would.
int hc_debug_enqueue_keys(hc_context* ctx, const unsigned long long* keys, int nkeys, int nframes) {
  if (!ctx || !ctx->configured) return hc::set_err(HC_ERR_STATE, "context not configured");
  if (!keys || nkeys < 0 || nframes <= 0 || nframes > ctx->max_batch || (size_t)nkeys > ctx->cand_cap)
    return hc::set_err(HC_ERR_ARG, "bad key list / batch size");
  if (ctx->pending) return hc::set_err(HC_ERR_STATE, "previous batch not waited");
  for (int i = 0; i < nkeys; i++)
    if ((int)(keys[i] >> 40) >= nframes || ((keys[i] >> 32) & 0xFF) >= ctx->plan.levels.size()) return hc::set_err(HC_ERR_ARG, "key out of range");
  HIPCHK(hipSetDevice(ctx->device));
  hc_context* c = ctx;
  c->inflight_src = nullptr;
  c->last_launch = hc_launch_info{};
  c->last_launch.nframes = nframes;
  c->strips_ran = false;
  c->timed_last = false;
  c->group_ran = c->device_group && nframes >= c->k.group_min_batch && hc::group_threshold(c) > 0 && c->max_batch <= 8192;
  if (c->group_ran)
    if (int rc = hc::ensure_group(c)) return rc;
  HIPCHK(hipMemsetAsync(c->d_count, 0, sizeof(unsigned int) * (1 + (size_t)nframes), c->stream));
  const unsigned int n = (unsigned int)nkeys;
  HIPCHK(hipMemcpyAsync(c->d_count, &n, sizeof n, hipMemcpyHostToDevice, c->stream));
  if (nkeys) HIPCHK(hipMemcpyAsync(c->d_cand, keys, (size_t)nkeys * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));     // (n and keys are the caller's pageable memory)
  if (int rc = hc::publish_results(c)) return rc;
  c->pending = nframes;
  return HC_OK;
}

int hc_detect_batch_enqueue_host(hc_context* ctx, const uint8_t* h_frames, size_t frame_stride_bytes,
                                 size_t row_step_bytes, int nframes) {
  if (!ctx || !ctx->configured) return hc::set_err(HC_ERR_STATE, "context not configured");
  if (!h_frames || nframes <= 0 || nframes > ctx->max_batch) return hc::set_err(HC_ERR_ARG, "bad frame pointer / batch size");
  if (row_step_bytes < (size_t)ctx->plan.cols) return hc::set_err(HC_ERR_ARG, "row step smaller than the row");
  if (ctx->pending) return hc::set_err(HC_ERR_STATE, "previous batch not waited");
  HIPCHK(hipSetDevice(ctx->device));
  if (int rc = hc::shared_enqueue_check(ctx, h_frames, nframes)) return rc;
  if (ctx->leader) return hc::enqueue_device(ctx, h_frames, frame_stride_bytes, row_step_bytes, nframes);   // frames went up with the leader
  if (int rc = hc::ensure_frames(ctx)) return rc;
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
  return hc::enqueue_device(ctx, ctx->d_frames, rows * cols, cols, nframes);
}

}  // extern "C"
