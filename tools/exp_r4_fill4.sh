#!/bin/bash
# round 4: 16-bit-tile fill with four entries per lane (tree) vs two (-DHC_P16_FILL_PAIRS), alternating in one box
out=gpurun_out/r4_fill4.txt
: > $out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "p16 or frame_chunks or full_size or strips" > gpurun_out/r4_fill4_test.log 2>&1; tail -1 gpurun_out/r4_fill4_test.log >> $out
D=$PWD/ugproj-detector_amd
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_fill4.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['ms_per_step'], j['roofline']['all_kernels_ms_per_launch']['k_cascade'], j['parity']['bit_exact'])
" >> $out; }
for rep in 1 2 3; do
run python bench.py --steps 20 --cpu-sample 0 --no-extra
run HC_LIB_PATH=$D/libhip_cascade_pairs.so python bench.py --steps 20 --cpu-sample 0 --no-extra
done
cat $out
