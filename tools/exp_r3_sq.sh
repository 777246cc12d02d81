#!/bin/bash
# round 3: k_pyramid_integral keeps the squares' integral in LDS (ring words of the grid columns beyond 512 in LDS too) -- parity, A/B
set -e
mkdir -p gpurun_out/r3q
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_route.py -m gpu -x -q > gpurun_out/r3q/gputest.log 2>&1 || { tail -40 gpurun_out/r3q/gputest.log; exit 1; }
tail -2 gpurun_out/r3q/gputest.log
run() {
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 20 --cpu-sample 0 --no-extra > gpurun_out/r3q/bench_$label.json 2> gpurun_out/r3q/bench_$label.err || { tail -5 gpurun_out/r3q/bench_$label.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r3q/bench_$label.json"))
print("%-12s fps=%.0f ms/step=%.3f k_cascade=%.3f prep=%.3f parity=%s" % ("$label", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_launch"]["k_cascade"], d["roofline"]["all_kernels_ms_per_launch"]["k_pyramid_integral"], d["parity"]["bit_exact"]))
PY
}
for rep in 1 2; do run xring_$rep A=1; run noxring_$rep HC_VAR_XRING=0; run head_$rep HC_LIB_PATH=$PWD/ugproj-detector_amd/libhip_cascade_head.so; done
