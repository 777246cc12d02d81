// hc_internal.h -- what the translation units behind the C ABI share (never installed, never part of the ABI).
#pragma once

#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hip_cascade.h"
#include "cascade_model.h"

struct hc_cascade {
  hc::Model model;
  std::mutex mu;                 // guards sync_ctxs / last_result (sync API) and serialises the legacy-mode call
  // Contexts of the synchronous calls, most recently used first: one per (device, frame geometry, call parameters), so
  // a caller that alternates between a few image sizes (a tracker scanning at two resolutions) re-allocates nothing
  // after the first call of each.  At most sync_cache_size() of them (HC_SYNC_CONTEXTS, default 4); the least
  // recently used one is re-configured for a new combination.
  std::vector<hc_context*> sync_ctxs;
  int sync_cache_size = 4;       // HC_SYNC_CONTEXTS, read when the cascade is loaded
  // Rectangles of a synchronous call that ended in HC_ERR_CAPACITY (hc_detect_multiscale_reread), kept PER CALLING THREAD:
  // the mutex is released between the failing call and the re-read, and another thread's call on the same cascade
  // must not take the first thread's rectangles away.  A thread's next synchronous call drops its own entry.
  std::map<std::thread::id, std::vector<hc_rect>> pending_results;
};

namespace hc {
// records the thread's last error text (hc_last_error) and returns `code`
int set_error(int code, const std::string& what);
// counts device / pinned-host allocations made by the library (hc_debug_alloc_count)
void count_alloc();
// keeps `boxes` for hc_detect_multiscale_reread when they do not fit the caller's buffer; copies them out otherwise
int deliver_sync_result(hc_cascade* c, const std::vector<hc_rect>& boxes, hc_rect* out, int out_capacity, int* out_count);
}
