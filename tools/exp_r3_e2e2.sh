#!/bin/bash
# round 3: PCIe-inclusive C++ driver with both pipeline slots on one stream + upload streams; the new multi-slot test; C++ driver tests
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3y
rm -rf $O; mkdir -p $O
cd $R
bash tools/exp_e2e.sh > $O/e2e.txt 2>&1; cat $O/e2e.txt
timeout -k 10 900 python -m pytest tests/test_gpu_bench_route.py -m gpu -x -q > $O/gputest.log 2>&1; tail -3 $O/gputest.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "demo or driver or shim or pool or cpp" > $O/gputest2.log 2>&1; tail -3 $O/gputest2.log
