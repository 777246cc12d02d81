#!/bin/bash
# a shared-pass member's own k_variance beside the leader's cascade launches (tree) vs on the member's stream (HC_VAR_SIDE=0)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/varside; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "shared or two_cascades or four_cascades" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --cpu-sample 0 --no-extra $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_launch'], d['parity']['bit_exact'])"; }
ARGS="--config 3 --steps 8"; for v in 1 0 1 0; do run HC_VAR_SIDE=$v; done 2>&1 | tee $O/c3.txt
ARGS="--config 5 --steps 4"; for v in 1 0 1 0; do run HC_VAR_SIDE=$v; done 2>&1 | tee $O/c5.txt
