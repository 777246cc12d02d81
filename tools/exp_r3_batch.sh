#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3b512; rm -rf $O; mkdir -p $O; cd $R
for cb in "3 96" "3 192" "5 32" "5 64" "2 768"; do
  set -- $cb
  timeout -k 10 400 python bench.py --config $1 --batch $2 --steps 8 --cpu-sample 0 --no-extra > $O/c$1_b$2.json 2> $O/c$1_b$2.err || { tail -3 $O/c$1_b$2.err; continue; }
  python -c "import json; d=json.load(open('$O/c$1_b$2.json')); print('config $1 batch $2', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['all_kernels_ms_per_launch'].items() if v > 0.02}, d['parity']['bit_exact'])"
done
