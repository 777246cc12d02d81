#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3pf; rm -rf $O; mkdir -p $O; cd $R
for rep in 1 2; do
for lib in "" _pf4 _pf3; do
HC_LIB_PATH=$R/ugproj-detector_amd/libhip_cascade$lib.so timeout -k 10 300 python bench.py --steps 20 --cpu-sample 0 --no-extra > $O/b$lib.json 2> $O/b$lib.err || { tail -3 $O/b$lib.err; continue; }
python -c "import json; d=json.load(open('$O/b$lib.json')); print('prefetch lib$lib', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['all_kernels_ms_per_launch'].items() if v > 0.02}, d['parity']['bit_exact'])"
done
done
