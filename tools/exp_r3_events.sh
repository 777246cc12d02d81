#!/bin/bash
# round 3: results awaited by event instead of by stream synchronisation (hardware-queue sharing between contexts) --
# timeline of config 3, benches of configs 3 / 5 / 2, PCIe-inclusive C++ rate; then the multi-context tests
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3e
rm -rf $O; mkdir -p $O
cd $R
for c in 3 5 2; do
  timeout -k 10 400 python bench.py --config $c --steps 10 --cpu-sample 0 --no-extra > $O/bench_c$c.json 2> $O/bench_c$c.err || { tail -5 $O/bench_c$c.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/bench_c$c.json"))
print("config $c fps=%.0f ms/step=%.3f kernels=%s parity=%s" % (d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d["roofline"]["all_kernels_ms_per_launch"].items() if v > 0.05}, d["parity"]["bit_exact"]))
PY
done
bash tools/exp_e2e.sh > $O/e2e.txt 2>&1; cat $O/e2e.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace3 -- python3 $R/bench.py --config 3 --steps 6 --warmup 2 --cpu-sample 0 --no-extra > $O/trace3.log 2>&1 || { tail -5 $O/trace3.log; exit 1; }
python3 $R/tools/timeline_gaps.py $O/trace3 > $O/gaps3.txt; rm -rf $O/trace3; head -8 $O/gaps3.txt
cd $R && timeout -k 10 900 python -m pytest tests/test_gpu_bench_route.py tests/test_gpu_legacy.py -m gpu -x -q > $O/gputest.log 2>&1; tail -3 $O/gputest.log
