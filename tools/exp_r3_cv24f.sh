#!/bin/bash
# round 3: OpenCV-2.4 prep through the fused kernel (k_pyramid_integral<.., SEM = 1>) -- parity, then A/B against the four separate kernels
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3f
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_legacy.py -m gpu -x -q > $O/gputest.log 2>&1 || { tail -40 $O/gputest.log; exit 1; }
tail -2 $O/gputest.log
for rep in 1 2; do
  for f in 1 0; do
    HC_CV24_FUSED=$f timeout -k 10 300 python bench.py --semantics cv24 --steps 20 --cpu-sample 0 > $O/cv24_f${f}_$rep.json 2> $O/cv24_f${f}_$rep.err || { tail -5 $O/cv24_f${f}_$rep.err; exit 1; }
    python -c "import json; d=json.load(open('$O/cv24_f${f}_$rep.json')); print('fused=$f', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['all_kernels_ms_per_launch'].items()}, d['parity']['bit_exact'])"
  done
done
