// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the CASCADE half of the Haar-cascade path (the prep half:
// prep_kernels.hip):
//   k_cascade        A.6b/A.7/A.8/A.9 stage evaluation from LDS-staged integral tiles (cascade_body.inc, built per
//                    arithmetic: f32 / exact-integer rect weights), x-skip chain resolved in closed form with wave ballots
//   k_publish        the batch's raw candidates and {count, sequence number} to page-locked host memory
// plus the registry of compiled early stages (csrc/spec/*.hip and run-time plug-ins).
// Float work is IEEE single ops in the upstream order, compiled with -ffp-contract=off.  No MFMA.

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "device_types.h"
#include "kernels.h"
#include "device_common.h"

namespace hc {

// ------------------------------------------------------------------------------------------
// Cascade evaluation: cascade_body.inc, built twice (f32 rect weights / exact integer rect weights)
namespace casc_f32 {
#define HC_INTW 0
#include "cascade_body.inc"
#undef HC_INTW
}  // namespace casc_f32
namespace casc_int {
#define HC_INTW 1
#include "cascade_body.inc"
#undef HC_INTW

}  // namespace casc_int

// Compiled early stages (csrc/spec/*.hip, one translation unit per known cascade; spec_unit.inc): the table of units.
typedef const SpecUnit* (*SpecUnitFn)();
#if !defined(HC_PHASE) && !defined(HC_OFF_WORDS)
#define HC_SPEC(tag) const SpecUnit* spec_unit_##tag();
#include "spec/spec_list.inc"
#undef HC_SPEC
static const SpecUnitFn spec_units[] = {
#define HC_SPEC(tag) spec_unit_##tag,
#include "spec/spec_list.inc"
#undef HC_SPEC
};
#else
static const SpecUnitFn spec_units[] = {nullptr};   // instrumented / word-offset builds run the generic kernel only
#endif

// Units that arrive at run time (hc_spec_plugin_load: a cascade the library was not built with, compiled into its own
// small shared object by tools/build_spec_plugin.sh).  Ids continue behind the built-in units; units are never removed.
static std::mutex g_spec_mu;
static std::vector<const SpecUnit*> g_spec_extra;

static const SpecUnit* spec_unit_by_id(int id) {   // 1-based
  constexpr int nb = (int)(sizeof(spec_units) / sizeof(spec_units[0]));
  if (id >= 1 && id <= nb) return spec_units[id - 1] ? spec_units[id - 1]() : nullptr;
  std::lock_guard<std::mutex> l(g_spec_mu);
  const int k = id - nb - 1;
  return k >= 0 && k < (int)g_spec_extra.size() ? g_spec_extra[k] : nullptr;
}

const SpecUnit* spec_unit(int id) { return spec_unit_by_id(id); }

int spec_register(const SpecUnit* u) {
  if (!u || !u->launch || u->nstages <= 0) return -1;
  std::lock_guard<std::mutex> l(g_spec_mu);
  for (const SpecUnit* e : g_spec_extra)
    if (e->hash == u->hash && e->win_w == u->win_w && e->win_h == u->win_h && e->p16 == u->p16 && e->cv24 == u->cv24) return 0;   // already there
  g_spec_extra.push_back(u);
  return 0;
}

int spec_lookup(unsigned long long hash, int win_w, int win_h, int* nstages, bool p16, bool cv24) {
  constexpr int nb = (int)(sizeof(spec_units) / sizeof(spec_units[0]));
  int nextra;
  { std::lock_guard<std::mutex> l(g_spec_mu); nextra = (int)g_spec_extra.size(); }
  for (int id = 1; id <= nb + nextra; id++) {
    const SpecUnit* u = spec_unit_by_id(id);
    if (u && u->hash == hash && u->win_w == win_w && u->win_h == win_h && u->p16 == p16 && u->cv24 == cv24) {
      if (nstages) *nstages = u->nstages;
      return id;
    }
  }
  if (nstages) *nstages = 0;
  return 0;
}

// ------------------------------------------------------------------------------------------
// launchers
#define HC_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

// ---- results to the host without a copy command --------------------------------------------------
// The last launch of a batch: the raw candidates go to page-locked host memory by plain stores (a few thousand 8-byte
// words per batch); every thread fences its own stores at system scope, the workgroup takes a ticket, and the workgroup
// that draws the last ticket publishes {count, sequence number} with a release store -- one kernel, so the order
// "candidates, then flag" rests on fences and the ticket's atomic alone, not on how the runtime flushes between two
// kernels of a queue.  The host polls the sequence word (api_wait.cpp wait_published): a completion that does not pass
// through the runtime's stream / event waits, which return only when the hardware queue they share with another
// context's streams has drained (one whole batch late with four contexts alive).  h_cand / h_pub are coherent host
// allocations (hipHostMallocCoherent): device stores bypass the caches.
__global__ __launch_bounds__(256) void k_publish(const unsigned int* __restrict__ count, const unsigned long long* __restrict__ cand,
                                                 unsigned int cap, unsigned long long* __restrict__ h_cand, unsigned int* h_pub,
                                                 unsigned int seq, unsigned int* ticket, const unsigned int* err) {
  const unsigned int total = *count, n = min(total, cap);
  if (cand && h_cand)
    for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) h_cand[i] = cand[i];
  __threadfence_system();                       // this thread's candidate stores are performed for the host ...
  __syncthreads();                              // ... and so are those of every thread of the workgroup
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {                   // every workgroup has fenced its share
      *ticket = 0u;                             // ready for the next batch (stream order: the next k_publish starts after this one)
      h_pub[1] = total;
      h_pub[2] = err ? __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;   // a strip hand-off of the batch gave up waiting
      __threadfence_system();
      __hip_atomic_store(&h_pub[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

int launch_publish(hipStream_t st, const unsigned int* count, const unsigned long long* cand, unsigned int cap,
                   unsigned long long* h_cand, unsigned int* h_pub, unsigned int seq, unsigned int* ticket, const unsigned int* err) {
  const bool copy = cap && cand && h_cand;
  hipLaunchKernelGGL(k_publish, dim3(copy ? 32 : 1), dim3(256), 0, st, count, copy ? cand : nullptr, cap, copy ? h_cand : nullptr, h_pub,
                     seq, ticket, err);
  HC_LAUNCH_CHECK();
  return 0;
}

int launch_cascade(hipStream_t st, int nframes, bool tree, bool segments, int rows, int nwaves, size_t lds_bytes,
                   const CascadeArgs& args, int* compiled, int* variant) {
  if (compiled) *compiled = 0;
  if (variant) *variant = 0;
  if (args.nbands == 0 || nframes == 0) return 0;
  // two builds of the same kernel text (cascade_body.inc): rect weights as f32, or -- when the host has shown that every
  // weight is a small integer with w0 == -1 and every weighted rect sum stays below 2^24 -- as exact integer arithmetic
  // compiled early stages for this cascade (args.spec from spec_lookup); -1: no instance for this launch shape
  const SpecUnit* unit = args.int_weights && args.spec > 0 ? spec_unit_by_id(args.spec) : nullptr;
  if (unit && unit->tree == tree) {
    const int rc = unit->launch(st, nframes, segments, rows, nwaves, lds_bytes, args);
    if (rc >= 0) {
      if (compiled) *compiled = unit->nstages;
      if (variant) *variant = unit->p16 ? 4 /* HC_VARIANT_P16 */ : 3 /* HC_VARIANT_SPEC */;
      return rc;
    }
    if (unit->p16) return (int)hipErrorInvalidValue;   // the arguments are in that build's own formats: no other kernel may take them
  }
  if (variant) *variant = args.int_weights ? 2 /* HC_VARIANT_INT */ : 1 /* HC_VARIANT_F32 */;
  return args.int_weights ? casc_int::launch_cascade_ns(st, nframes, tree, segments, rows, nwaves, lds_bytes, args)
                          : casc_f32::launch_cascade_ns(st, nframes, tree, segments, rows, nwaves, lds_bytes, args);
}

}  // namespace hc
