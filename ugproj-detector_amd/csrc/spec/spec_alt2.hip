// compiled early stages of one stock cascade: see ../spec_unit.inc; the stage code is spec_alt2.inc (make specs)
#define HC_SPEC_TAG alt2
#define HC_SPEC_FILE "spec/spec_alt2.inc"
#include "../spec_unit.inc"
