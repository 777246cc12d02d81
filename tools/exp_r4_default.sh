#!/bin/bash
# round 4: default at 720p -- compiled stages of its 32-bit unit (6 tree / 9 / 12), cooperative mode off in every build, finish threshold
out=gpurun_out/r4_default.txt
: > $out
D=$PWD/ugproj-detector_amd
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_default.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['roofline']['all_kernels_ms_per_launch'].get('k_cascade'), [l.get('compiled_stages') for l in j['config']['launch']], j['parity']['bit_exact'])
" >> $out; }
B="python bench.py --steps 12 --cpu-sample 0 --no-extra --cascade default"
run $B
run HC_LIB_PATH=$D/libhip_cascade_d9.so $B
run HC_LIB_PATH=$D/libhip_cascade_d12.so $B
run HC_LIB_PATH=$D/libhip_cascade_nc.so $B
run HC_LIB_PATH=$D/libhip_cascade_ncd12.so $B
run HC_SPARSE_MAX=48 $B
run HC_SPARSE_MAX=64 $B
run HC_SPARSE_MAX=24 $B
run $B
cat $out
