#!/bin/bash
# round 3: PCIe-inclusive C++ driver with both pipeline slots on one stream + upload streams; the new multi-slot test; C++ driver tests
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3y
rm -rf $O; mkdir -p $O
cd $R

timeout -k 10 900 python -m pytest tests/test_gpu_bench_route.py -m gpu -x -q > $O/gputest.log 2>&1; tail -3 $O/gputest.log
for c in 5 3; do timeout -k 10 400 python bench.py --config $c --steps 10 --cpu-sample 0 --no-extra > $O/bench_c$c.json 2> $O/bench_c$c.err; python -c "import json; d=json.load(open(\"$O/bench_c$c.json\")); print($c, d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"all_kernels_ms_per_launch\"], d[\"parity\"][\"bit_exact\"])"; done
