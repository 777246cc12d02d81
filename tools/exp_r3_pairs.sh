#!/bin/bash
# round 3: generic tree path in pair form (two-node trees: both nodes from scalar loads, leaf by two exact FMAs)
set -e
mkdir -p gpurun_out/r3g
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3g/gputest.log 2>&1 || { tail -40 gpurun_out/r3g/gputest.log; exit 1; }
tail -2 gpurun_out/r3g/gputest.log
fi
run() {
  local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --config 3 --cascade alt2 --steps 10 --cpu-sample 0 > gpurun_out/r3g/bench_$label.json 2> gpurun_out/r3g/bench_$label.err || { tail -5 gpurun_out/r3g/bench_$label.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r3g/bench_$label.json"))
print("%-28s fps=%.0f ms/step=%.3f kernels=%s parity=%s" % ("$label", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_launch"], d["parity"]["bit_exact"]))
PY
}
run generic_pairs HC_SPEC=0
run generic_walk HC_SPEC=0 HC_PAIR_FORM=0
run compiled8_pairs A=1
run compiled8_walk HC_PAIR_FORM=0
run generic_pairs_2 HC_SPEC=0
run compiled8_pairs_2 A=1
