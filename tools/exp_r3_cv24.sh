#!/bin/bash
# round 3: OpenCV-2.4 semantics through the LDS-tiled band kernel -- parity, then the rate next to the >= 4 semantics
set -e
mkdir -p gpurun_out/r3h
timeout -k 10 900 python -m pytest tests/test_gpu_legacy.py -m gpu -x -q > gpurun_out/r3h/gputest.log 2>&1 || { tail -60 gpurun_out/r3h/gputest.log; exit 1; }
tail -2 gpurun_out/r3h/gputest.log
run() {
  local label=$1; shift
  timeout -k 10 300 python bench.py --steps 10 --cpu-sample 0 --no-extra "$@" > gpurun_out/r3h/bench_$label.json 2> gpurun_out/r3h/bench_$label.err || { tail -5 gpurun_out/r3h/bench_$label.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r3h/bench_$label.json"))
print("%-22s fps=%.0f ms/step=%.3f kernels=%s frac=%.4f parity=%s" % ("$label", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_launch"], d["roofline"]["frac"], d["parity"]))
PY
}
run cv24_alt --semantics cv24
run cv4_alt
run cv24_alt2 --semantics cv24 --cascade alt2
run cv24_default --semantics cv24 --cascade default
run cv24_alt_tree --semantics cv24 --cascade alt_tree --batch 32
