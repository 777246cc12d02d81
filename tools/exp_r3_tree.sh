#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3tree; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_legacy.py -m gpu -x -q > $O/gputest.log 2>&1 || { tail -30 $O/gputest.log; exit 1; }
tail -2 $O/gputest.log
for lib in "" _t128 _t0; do
HC_LIB_PATH=$R/ugproj-detector_amd/libhip_cascade$lib.so timeout -k 10 400 python bench.py --cascade alt_tree --semantics cv24 --batch 256 --steps 10 --cpu-sample 0 --no-extra > $O/tree$lib.json 2> $O/tree$lib.err || { tail -3 $O/tree$lib.err; continue; }
python -c "import json; d=json.load(open('$O/tree$lib.json')); print('cv24 alt_tree tail$lib', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['all_kernels_ms_per_launch'].items() if v > 0.02}, d['parity']['bit_exact'], d['parity']['boxes'])"
done
