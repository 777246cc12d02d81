#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3ch2; rm -rf $O; mkdir -p $O; cd $R
for rep in 1 2 3; do
for c in 128 0 64; do
  HC_FRAME_CHUNK=$c timeout -k 10 300 python bench.py --steps 30 --cpu-sample 0 --no-extra > $O/c${c}_$rep.json 2> $O/c${c}_$rep.err || { tail -3 $O/c${c}_$rep.err; continue; }
  python -c "import json; d=json.load(open('$O/c${c}_$rep.json')); print('chunk $c', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['all_kernels_ms_per_launch'].items() if v > 0.02}, d['parity']['bit_exact'])"
done
done
