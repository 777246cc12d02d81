#!/bin/bash
# round 3: variance factors from the LDS row prefixes (register ring) vs reading the integrals back; A/B in one box
set -e
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3b/gputest.log 2>&1 || { tail -30 gpurun_out/r3b/gputest.log; exit 1; }
tail -3 gpurun_out/r3b/gputest.log
for rep in 1 2; do
  for ring in 1 0; do
    HC_VAR_RING=$ring timeout -k 10 200 python bench.py --steps 20 --cpu-sample 0 --no-extra > gpurun_out/r3b/bench_ring${ring}_$rep.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("gpurun_out/r3b/bench_ring${ring}_$rep.json"))
print("ring=$ring rep=$rep fps=%.0f ms/step=%.3f kernels=%s parity=%s" % (d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_launch"], d["parity"]["bit_exact"]))
PY
  done
done
