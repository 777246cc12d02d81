// spec_plugin.hip -- compiled early stages of ONE cascade as a stand-alone shared object (tools/build_spec_plugin.sh):
// the stage code tools/gen_spec.py wrote for that cascade (tag "plugin", passed as -DHC_SPEC_FILE=...), wrapped into the
// same kernel text as the built-in units (spec_unit.inc), plus the two symbols hc_spec_plugin_load looks for.
#ifndef HC_SPEC_FILE
#error "build with tools/build_spec_plugin.sh (-DHC_SPEC_FILE=\"<generated .inc>\")"
#endif
#define HC_SPEC_TAG plugin
#include "spec_unit.inc"

extern "C" {
__attribute__((visibility("default"))) int hc_spec_plugin_abi(void) { return hc::SPEC_PLUGIN_ABI; }
__attribute__((visibility("default"))) const void* hc_spec_plugin_unit(void) { return hc::spec_unit_plugin(); }
}
