#!/bin/bash
# round 4: 16-bit-tile fill with two loads + a DPP wave shift per row (tree) vs three loads (-DHC_P16_FILL_LOADS3), alternating in one box
out=gpurun_out/r4_fill.txt
: > $out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "p16_build or frame_chunks" > gpurun_out/r4_fill_test.log 2>&1; tail -1 gpurun_out/r4_fill_test.log >> $out
D=$PWD/ugproj-detector_amd
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_fill.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['ms_per_step'], j['roofline']['all_kernels_ms_per_launch']['k_cascade'], j['parity']['bit_exact'])
" >> $out; }
for rep in 1 2 3; do
run python bench.py --steps 20 --cpu-sample 0 --no-extra
run HC_LIB_PATH=$D/libhip_cascade_l3.so python bench.py --steps 20 --cpu-sample 0 --no-extra
done
cat $out
