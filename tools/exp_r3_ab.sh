#!/bin/bash
# round 3: full GPU suite, then default bench A/B: this build vs the library of the previous commit (libhip_cascade_head.so)
set -e
mkdir -p gpurun_out/r3ab
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3ab/gputest.log 2>&1 || { tail -40 gpurun_out/r3ab/gputest.log; exit 1; }
tail -2 gpurun_out/r3ab/gputest.log
run() {
  local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --cpu-sample 0 --no-extra > gpurun_out/r3ab/bench_$label.json 2> gpurun_out/r3ab/bench_$label.err || { tail -5 gpurun_out/r3ab/bench_$label.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r3ab/bench_$label.json"))
print("%-12s fps=%.0f ms/step=%.3f k_cascade=%.3f prep=%.3f parity=%s" % ("$label", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_launch"]["k_cascade"], d["roofline"]["all_kernels_ms_per_launch"]["k_pyramid_integral"], d["parity"]["bit_exact"]))
PY
}
for rep in 1 2; do run new_$rep A=1; run head_$rep HC_LIB_PATH=$PWD/ugproj-detector_amd/libhip_cascade_head.so; done
