#!/bin/bash
set -e
mkdir -p gpurun_out/r3j
timeout -k 10 900 python -m pytest tests/test_gpu_legacy.py -m gpu -x -q > gpurun_out/r3j/gputest.log 2>&1 || { tail -60 gpurun_out/r3j/gputest.log; exit 1; }
tail -2 gpurun_out/r3j/gputest.log
for c in alt_tree alt; do
timeout -k 10 300 python bench.py --steps 10 --cpu-sample 0 --no-extra --semantics cv24 --cascade $c > gpurun_out/r3j/bench_$c.json 2> gpurun_out/r3j/bench_$c.err || { tail -5 gpurun_out/r3j/bench_$c.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r3j/bench_$c.json"))
print("$c fps=%.0f ms/step=%.3f kernels=%s frac=%.4f parity=%s" % (d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_launch"], d["roofline"]["frac"], d["parity"]))
PY
done
