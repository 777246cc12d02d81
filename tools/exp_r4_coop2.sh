#!/bin/bash
# round 4: without cooperative mode (-DHC_NO_COOP): compiled stages 12 / 16 / 22 (16-bit unit), 12 in both units, finish threshold
out=gpurun_out/r4_coop2.txt
: > $out
D=$PWD/ugproj-detector_amd
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_coop2.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['roofline']['all_kernels_ms_per_launch']['k_cascade'], j['config']['launch'][0]['compiled_stages'], j['parity']['bit_exact'])
" >> $out; }
B="python bench.py --steps 20 --cpu-sample 0 --no-extra"
run HC_LIB_PATH=$D/libhip_cascade_nc12.so $B
run HC_LIB_PATH=$D/libhip_cascade_nc16.so $B
run HC_LIB_PATH=$D/libhip_cascade_nc22.so $B
run HC_LIB_PATH=$D/libhip_cascade_nc12b.so $B
run HC_LIB_PATH=$D/libhip_cascade_nc12.so HC_SPARSE_MAX=8 $B
run HC_LIB_PATH=$D/libhip_cascade_nc12.so HC_SPARSE_MAX=16 $B
run HC_LIB_PATH=$D/libhip_cascade_nc12.so HC_SPARSE_MAX=48 $B
run HC_LIB_PATH=$D/libhip_cascade_nc12.so HC_SPARSE_MAX=64 $B
run HC_LIB_PATH=$D/libhip_cascade_nc12.so $B
cat $out
