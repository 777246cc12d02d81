#!/bin/bash
# K6 (rectangle grouping on the device): its tests, the host time of a wait with and without it, the headline both ways
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/group; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_bench_route.py -x -q -m gpu -k "grouping" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
for g in 1 0; do
  echo "== HC_DEVICE_GROUP=$g exp_hosttime 512"; HC_DEVICE_GROUP=$g timeout -k 10 200 python tools/exp_hosttime.py 512 2>&1 | tail -5
done | tee $O/hosttime.txt
for g in 1 0 1 0; do
  echo "== HC_DEVICE_GROUP=$g bench"; HC_DEVICE_GROUP=$g timeout -k 10 200 python bench.py --steps 20 --cpu-sample 0 --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['parity']['bit_exact'])"
done | tee $O/bench.txt
