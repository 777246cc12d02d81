#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3st; rm -rf $O; mkdir -p $O; cd $R
for s in 1 2 1 2; do
  timeout -k 10 300 python bench.py --streams $s --steps 20 --cpu-sample 0 --no-extra > $O/s$s.json 2> $O/s$s.err || { tail -3 $O/s$s.err; continue; }
  python -c "import json; d=json.load(open('$O/s$s.json')); print('streams $s', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['all_kernels_ms_per_launch'].items() if v > 0.02}, d['parity']['bit_exact'])"
done
