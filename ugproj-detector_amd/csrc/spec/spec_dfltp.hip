// 16-bit-tile unit of haarcascade_frontalface_default: see ../spec_unit_p16.inc; the stage code is spec_dfltp.inc (make specs)
#define HC_SPEC_TAG dfltp
#define HC_SPEC_FILE "spec/spec_dfltp.inc"
#include "../spec_unit_p16.inc"
