#!/bin/bash
# round 4: compiled early stages with the 2.4 node arithmetic (csrc/spec/spec_*24.hip) on / off (HC_SPEC=0), 720p, 512 frames per launch
out=gpurun_out/r4_cv24spec.txt
: > $out
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_cv24spec.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['ms_per_step'], j['roofline']['all_kernels_ms_per_launch']['k_cascade'], j['config']['launch'][0]['variant'], j['config']['launch'][0]['compiled_stages'], j['parity']['bit_exact'])
" >> $out; }
for c in alt alt2 default alt_tree; do
run python bench.py --cascade $c --semantics cv24 --steps 10 --warmup 2 --cpu-sample 0 --no-extra
run HC_SPEC=0 python bench.py --cascade $c --semantics cv24 --steps 10 --warmup 2 --cpu-sample 0 --no-extra
done
cat $out
