#!/bin/bash
# round 3: config #2's frames through each of the four stock cascades (the shipped cfg names alt2), >= 4 and 2.4 semantics
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3pcasc; rm -rf $O; mkdir -p $O; cd $R
for sem in cv4 cv24; do
for c in alt alt2 default alt_tree; do
  timeout -k 10 400 python bench.py --cascade $c --semantics $sem --batch 256 --steps 10 --cpu-sample 0 --no-extra > $O/${sem}_$c.json 2> $O/${sem}_$c.err || { tail -3 $O/${sem}_$c.err; continue; }
  python -c "import json; d=json.load(open('$O/${sem}_$c.json')); print('$sem $c', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['all_kernels_ms_per_launch'].items() if v > 0.02}, d['parity']['bit_exact'], d['parity']['boxes'])"
done
done
