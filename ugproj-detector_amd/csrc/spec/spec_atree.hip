// compiled early stages of one stock cascade: see ../spec_unit.inc; the stage code is spec_atree.inc (make specs)
#define HC_SPEC_TAG atree
#define HC_SPEC_FILE "spec/spec_atree.inc"
#include "../spec_unit.inc"
