#!/bin/bash
# row-form stage count per cascade (HC_ROW_STAGES 1 / 2 (default) / 3 / 4): fps, k_cascade ms
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/rowstages; mkdir -p $O; cd $R
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --steps 12 --cpu-sample 0 --no-extra $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['launch_ms'], d['parity']['bit_exact'])"; }
for c in default alt2 alt; do
  ARGS="--cascade $c"
  for r in 2 3 1 4 2; do run HC_ROW_STAGES=$r; done
done 2>&1 | tee $O/rowstages.txt
