// 16-bit-tile unit of haarcascade_frontalface_alt: see ../spec_unit_p16.inc; the stage code is spec_altp.inc (make specs)
#define HC_SPEC_TAG altp
#define HC_SPEC_FILE "spec/spec_altp.inc"
#include "../spec_unit_p16.inc"
