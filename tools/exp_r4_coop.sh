#!/bin/bash
# round 4: cooperative mode (generic kind runs split over wavefronts) vs none (-DHC_NO_COOP: compiled dense loop), 6 and 12 compiled stages
out=gpurun_out/r4_coop.txt
: > $out
D=$PWD/ugproj-detector_amd
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_coop.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['roofline']['all_kernels_ms_per_launch']['k_cascade'], j['config']['launch'][0]['compiled_stages'], j['parity']['bit_exact'])
" >> $out; }
for rep in 1 2; do
run python bench.py --steps 20 --cpu-sample 0 --no-extra
run HC_LIB_PATH=$D/libhip_cascade_nc6.so python bench.py --steps 20 --cpu-sample 0 --no-extra
run HC_LIB_PATH=$D/libhip_cascade_nc12.so python bench.py --steps 20 --cpu-sample 0 --no-extra
done
cat $out
