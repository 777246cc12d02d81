#!/bin/bash
# round 3: sliced finish with dynamically pulled groups (HC_SLICE_GROUPS per wavefront) -- parity, then A/B in one box
set -e
mkdir -p gpurun_out/r3e
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests/test_gpu_bench_route.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3e/gputest.log 2>&1 || { tail -40 gpurun_out/r3e/gputest.log; exit 1; }
tail -2 gpurun_out/r3e/gputest.log
fi
run() {
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 20 --cpu-sample 0 --no-extra > gpurun_out/r3e/bench_$label.json 2> gpurun_out/r3e/bench_$label.err || { tail -5 gpurun_out/r3e/bench_$label.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r3e/bench_$label.json"))
print("%-22s fps=%.0f ms/step=%.3f k_cascade=%.3f prep=%.3f parity=%s" % ("$label", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_launch"]["k_cascade"], d["roofline"]["all_kernels_ms_per_launch"]["k_pyramid_integral"], d["parity"]["bit_exact"]))
PY
}
D=$PWD/ugproj-detector_amd
for rep in 1 2; do
  for g in g2 g1 g4 old; do run ${g}_$rep HC_LIB_PATH=$D/libhip_cascade_$g.so; done
done
for sm in 24 48 64; do run g2_sm$sm HC_LIB_PATH=$D/libhip_cascade_g2.so HC_SPARSE_MAX=$sm; done
for sm in 48 64; do run g4_sm$sm HC_LIB_PATH=$D/libhip_cascade_g4.so HC_SPARSE_MAX=$sm; done
