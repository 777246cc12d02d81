#!/bin/bash
# round 3: nearly full layers of the bank-ordered list keep their holes (HC_STRICT_FILL = least fill of such a layer; 33 = never)
set -e
mkdir -p gpurun_out/r3f
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests/test_gpu_bench_route.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3f/gputest.log 2>&1 || { tail -40 gpurun_out/r3f/gputest.log; exit 1; }
tail -2 gpurun_out/r3f/gputest.log
fi
run() {
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 20 --cpu-sample 0 --no-extra > gpurun_out/r3f/bench_$label.json 2> gpurun_out/r3f/bench_$label.err || { tail -5 gpurun_out/r3f/bench_$label.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r3f/bench_$label.json"))
print("%-22s fps=%.0f ms/step=%.3f k_cascade=%.3f prep=%.3f parity=%s" % ("$label", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_launch"]["k_cascade"], d["roofline"]["all_kernels_ms_per_launch"]["k_pyramid_integral"], d["parity"]["bit_exact"]))
PY
}
D=$PWD/ugproj-detector_amd
for rep in 1 2; do
  run f24_$rep A=1
  run f33_$rep HC_LIB_PATH=$D/libhip_cascade_f33.so
done
for f in 28 20 16; do run f$f HC_LIB_PATH=$D/libhip_cascade_f$f.so; done
