// compiled early stages of one stock cascade: see ../spec_unit.inc; the stage code is spec_alt.inc (make specs)
#define HC_SPEC_TAG alt
#define HC_SPEC_FILE "spec/spec_alt.inc"
#include "../spec_unit.inc"
