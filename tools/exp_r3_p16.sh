#!/bin/bash
# round 3: 16-bit tile + two windows per lane in the row-form stages (HC_P16) -- parity, then A/B against HC_P16=0
set -e
mkdir -p gpurun_out/r3l
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 600 python -m pytest tests/test_gpu_bench_route.py -m gpu -x -q -k "taps_720p or shared_prep_four or 1080p_batch" > gpurun_out/r3l/gputest1.log 2>&1 || { tail -40 gpurun_out/r3l/gputest1.log; exit 1; }
tail -2 gpurun_out/r3l/gputest1.log
HC_SEG_MAX_BATCH=0 HC_SIDE_MAX_BATCH=0 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3l/gputest2.log 2>&1 || { tail -40 gpurun_out/r3l/gputest2.log; exit 1; }
tail -2 gpurun_out/r3l/gputest2.log
fi
run() {
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 20 --cpu-sample 0 --no-extra > gpurun_out/r3l/bench_$label.json 2> gpurun_out/r3l/bench_$label.err || { tail -5 gpurun_out/r3l/bench_$label.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r3l/bench_$label.json"))
print("%-12s fps=%.0f ms/step=%.3f k_cascade=%.3f prep=%.3f parity=%s" % ("$label", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_launch"]["k_cascade"], d["roofline"]["all_kernels_ms_per_launch"]["k_pyramid_integral"], d["parity"]["bit_exact"]))
PY
}
for rep in 1 2; do run p16_$rep HC_P16=1; run std_$rep HC_P16=0; done
run p16_rows3 HC_LIB_PATH=$PWD/ugproj-detector_amd/libhip_cascade_ps3.so HC_ROW_STAGES=3
run p16_rows1 HC_ROW_STAGES=1
