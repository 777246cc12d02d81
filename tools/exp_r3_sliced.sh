#!/bin/bash
# round 3: sliced finish (several windows per wavefront, stage split over 64/m lanes) vs the one-window-per-wavefront
# sparse finish (-DHC_OLD_SPARSE build); then the entry threshold.  One box, alternating.
set -e
mkdir -p gpurun_out/r3c
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3c/gputest.log 2>&1 || { tail -40 gpurun_out/r3c/gputest.log; exit 1; }
tail -3 gpurun_out/r3c/gputest.log
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 20 --cpu-sample 0 --no-extra > gpurun_out/r3c/bench_$label.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c/bench_$label.json"))
print("%-22s fps=%.0f ms/step=%.3f k_cascade=%.3f prep=%.3f parity=%s" % ("$label", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_launch"]["k_cascade"], d["roofline"]["all_kernels_ms_per_launch"]["k_pyramid_integral"], d["parity"]["bit_exact"]))
PY
}
OLD=$PWD/ugproj-detector_amd/libhip_cascade_old.so
run new_1 A=1
run old_1 HC_LIB_PATH=$OLD
run new_2 A=1
run old_2 HC_LIB_PATH=$OLD
for sm in 16 24 48 64 96 128; do run new_sm$sm HC_SPARSE_MAX=$sm; done
