#!/bin/bash
# round 3: single-cascade runs at 1080p and 2160p (prep route of wide levels: read-back, CMAX 4 / 8)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3w
rm -rf $O; mkdir -p $O
cd $R
for cfg in "3 alt 96" "5 alt 32" "3 default 96"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --config $1 --cascade $2 --batch $3 --steps 10 --cpu-sample 0 > $O/c$1_$2.json 2> $O/c$1_$2.err || { tail -5 $O/c$1_$2.err; continue; }
  python -c "import json; d=json.load(open('$O/c$1_$2.json')); print('config $1 $2', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['all_kernels_ms_per_launch'].items() if v > 0.02}, d['parity']['bit_exact'])"
done
