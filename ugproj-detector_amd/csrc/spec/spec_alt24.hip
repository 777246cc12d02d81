// compiled early stages of haarcascade_frontalface_alt with OpenCV 2.4's node arithmetic: see ../spec_unit_cv24.inc; the stage code is spec_alt24.inc (make specs)
#define HC_SPEC_TAG alt24
#define HC_SPEC_FILE "spec/spec_alt24.inc"
#include "../spec_unit_cv24.inc"
