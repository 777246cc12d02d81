#!/bin/bash
# round 4: OpenCV-2.4 semantics per cascade at 720p; alt_tree's stage tree as alternative chains (default) vs the preorder sweep (HC_CV24_SEG=0)
out=gpurun_out/r4_cv24.txt
: > $out
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_cv24.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['ms_per_step'], j['roofline']['all_kernels_ms_per_launch'], j['config']['launch'][0]['variant'], j['parity']['bit_exact'])
" >> $out; }
run python bench.py --cascade alt_tree --semantics cv24 --steps 10 --warmup 2 --cpu-sample 0 --no-extra
run HC_CV24_SEG=0 python bench.py --cascade alt_tree --semantics cv24 --steps 5 --warmup 1 --cpu-sample 0 --no-extra
run python bench.py --cascade alt --semantics cv24 --steps 10 --warmup 2 --cpu-sample 0 --no-extra
run python bench.py --cascade alt2 --semantics cv24 --steps 10 --warmup 2 --cpu-sample 0 --no-extra
run python bench.py --cascade default --semantics cv24 --steps 10 --warmup 2 --cpu-sample 0 --no-extra
cat $out
