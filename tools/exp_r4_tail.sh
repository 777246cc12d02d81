#!/bin/bash
# K6's kernels + publication on the side stream (tree) vs on the main stream (HC_TAIL_SIDE=0): fps, ms per step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/tail; mkdir -p $O; cd $R
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --cpu-sample 0 --no-extra $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['parity']['bit_exact'])"; }
ARGS="--steps 20"; for v in 1 0 1 0; do run HC_TAIL_SIDE=$v; done 2>&1 | tee $O/c2.txt
ARGS="--config 3 --steps 8"; for v in 1 0 1 0; do run HC_TAIL_SIDE=$v; done 2>&1 | tee $O/c3.txt
ARGS="--steps 12 --cascade default"; for v in 1 0; do run HC_TAIL_SIDE=$v; done 2>&1 | tee $O/dflt.txt
