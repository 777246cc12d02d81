#!/bin/bash
# round 4: compiled stages of the 16-bit-tile unit of alt (scan-step-1 launch): 4 / 6 (tree) / 8, alternating in one box
out=gpurun_out/r4_p16stages.txt
: > $out
D=$PWD/ugproj-detector_amd
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_p16stages.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['roofline']['all_kernels_ms_per_launch']['k_cascade'], j['config']['launch'][0]['compiled_stages'], j['parity']['bit_exact'])
" >> $out; }
for rep in 1 2; do
run python bench.py --steps 20 --cpu-sample 0 --no-extra
run HC_LIB_PATH=$D/libhip_cascade_p10.so python bench.py --steps 20 --cpu-sample 0 --no-extra
run HC_LIB_PATH=$D/libhip_cascade_p8.so python bench.py --steps 20 --cpu-sample 0 --no-extra
run HC_LIB_PATH=$D/libhip_cascade_p12.so python bench.py --steps 20 --cpu-sample 0 --no-extra
done
cat $out
