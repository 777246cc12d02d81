// compiled early stages of haarcascade_frontalface_alt2 with OpenCV 2.4's node arithmetic: see ../spec_unit_cv24.inc; the stage code is spec_alt224.inc (make specs)
#define HC_SPEC_TAG alt224
#define HC_SPEC_FILE "spec/spec_alt224.inc"
#include "../spec_unit_cv24.inc"
