// 16-bit-tile unit of haarcascade_frontalface_alt_tree: see ../spec_unit_p16.inc; the stage code is spec_atreep.inc (make specs)
#define HC_SPEC_TAG atreep
#define HC_SPEC_FILE "spec/spec_atreep.inc"
#include "../spec_unit_p16.inc"
