// hc_internal.h -- what the translation units behind the C ABI share (never installed, never part of the ABI).
#pragma once

#include <mutex>
#include <string>

#include "../../include/hip_cascade.h"
#include "cascade_model.h"

struct hc_cascade {
  hc::Model model;
  std::mutex mu;                 // guards default_ctx (sync API) and serialises the legacy-mode call
  hc_context* default_ctx = nullptr;
};

namespace hc {
// records the thread's last error text (hc_last_error) and returns `code`
int set_error(int code, const std::string& what);
}
