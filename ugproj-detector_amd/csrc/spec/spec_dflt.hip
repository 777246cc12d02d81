// compiled early stages of one stock cascade: see ../spec_unit.inc; the stage code is spec_dflt.inc (make specs)
#define HC_SPEC_TAG dflt
#define HC_SPEC_FILE "spec/spec_dflt.inc"
#include "../spec_unit.inc"
