// api_plugins.cpp -- compiled early stages that arrive at run time: shared objects built by tools/build_spec_plugin.sh for a
// cascade the library was not built with (a retrained haarcascade_frontalface_alt2, the file the shipped configuration
// names: reference conf/apink.mrchu.720p.cfg:10), registered by hc_spec_plugin_load or HC_SPEC_PLUGINS.

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "hc_context.h"

namespace hc {

// compiled-stage units from HC_SPEC_PLUGINS (colon-separated shared objects), loaded once before the first lookup
void ensure_env_plugins() {
  static std::once_flag once;
  std::call_once(once, [] {
    const char* e = getenv("HC_SPEC_PLUGINS");
    if (!e) return;
    std::string s(e);
    size_t a = 0;
    while (a <= s.size()) {
      const size_t b = s.find(':', a);
      const std::string one = s.substr(a, b == std::string::npos ? std::string::npos : b - a);
      if (!one.empty() && plugin_load(one.c_str()) != HC_OK) fprintf(stderr, "hip_cascade: HC_SPEC_PLUGINS: %s\n", hc::last_err().c_str());
      if (b == std::string::npos) break;
      a = b + 1;
    }
  });
}

int plugin_load(const char* path) {
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return hc::set_err(HC_ERR_IO, std::string("cannot load the compiled-stage plug-in: ") + dlerror());
  typedef int (*abi_fn)(void);
  typedef const void* (*unit_fn)(void);
  const abi_fn abi = (abi_fn)dlsym(h, "hc_spec_plugin_abi");
  const unit_fn unit = (unit_fn)dlsym(h, "hc_spec_plugin_unit");
  if (!abi || !unit) { dlclose(h); return hc::set_err(HC_ERR_FORMAT, std::string(path) + " is not a hip_cascade compiled-stage plug-in"); }
  if (abi() != hc::SPEC_PLUGIN_ABI) {
    dlclose(h);
    return hc::set_err(HC_ERR_UNSUPPORTED, std::string(path) + " was built against another version of the kernel sources: rebuild it "
                                                             "(tools/build_spec_plugin.sh)");
  }
  if (hc::spec_register((const hc::SpecUnit*)unit()) != 0) { dlclose(h); return hc::set_err(HC_ERR_FORMAT, std::string(path) + ": empty unit"); }
  return HC_OK;                       // the object stays loaded for the life of the process: its kernels are launched from it
}

}  // namespace hc

extern "C" int hc_spec_plugin_load(const char* path) {
  if (!path) return hc::set_err(HC_ERR_ARG, "null path");
  return hc::plugin_load(path);
}
