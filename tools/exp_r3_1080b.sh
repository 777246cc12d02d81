#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3w2
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide_levels or tuning_knobs or 1080p or 2160" > $O/gputest.log 2>&1 || { tail -30 $O/gputest.log; exit 1; }
tail -2 $O/gputest.log
for f in policy 1; do
  if [ $f = 1 ]; then export HC_FUSE_VARIANCE=1; fi
  timeout -k 10 400 python bench.py --config 3 --cascade alt --batch 96 --steps 10 --cpu-sample 0 > $O/c3_alt_$f.json 2> $O/c3_alt_$f.err || { tail -5 $O/c3_alt_$f.err; continue; }
  python -c "import json; d=json.load(open('$O/c3_alt_$f.json')); print('1080p alt fuse_variance=$f', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['all_kernels_ms_per_launch'].items() if v > 0.02}, d['parity']['bit_exact'])"
done
