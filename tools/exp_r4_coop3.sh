#!/bin/bash
# round 4: cooperative mode on compiled stage pieces (tree: 16-bit unit, 12 stages) vs no cooperative mode at all (nc12) vs
# compiled pieces in the 16-bit kernel + no cooperative mode in the 32-bit one (cp_nc2)
out=gpurun_out/r4_coop3.txt
: > $out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "p16 or frame_chunks or full_size" > gpurun_out/r4_coop3_test.log 2>&1; tail -1 gpurun_out/r4_coop3_test.log >> $out
D=$PWD/ugproj-detector_amd
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_coop3.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['roofline']['all_kernels_ms_per_launch']['k_cascade'], j['config']['launch'][0]['compiled_stages'], j['parity']['bit_exact'])
" >> $out; }
B="python bench.py --steps 20 --cpu-sample 0 --no-extra"
for rep in 1 2; do
run $B
run HC_LIB_PATH=$D/libhip_cascade_nc12.so $B
run HC_LIB_PATH=$D/libhip_cascade_cp_nc2.so $B
done
run HC_SPARSE_MAX=16 $B
run HC_SPARSE_MAX=48 $B
cat $out
