#!/bin/bash
# round 4: 2048-window tiles (32 grid rows, 8 wavefronts) for the scan-step-1 launch of the 16-bit-tile build vs the 1024-window production shape
out=gpurun_out/r4_tile2048.txt
: > $out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tuning_knobs" > gpurun_out/r4_tile2048_test.log 2>&1; tail -2 gpurun_out/r4_tile2048_test.log >> $out
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_tile2048.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['ms_per_step'], j['roofline']['all_kernels_ms_per_launch']['k_cascade'], j['config']['launch'][0]['variant'], j['parity']['bit_exact'])
" >> $out; }
for rep in 1 2; do
run python bench.py --steps 20 --cpu-sample 0 --no-extra
run HC_BAND_ROWS1=32 HC_BAND_WAVES1=8 python bench.py --steps 20 --cpu-sample 0 --no-extra
done
run HC_BAND_ROWS1=32 HC_BAND_WAVES1=8 HC_SPARSE_MAX=64 python bench.py --steps 20 --cpu-sample 0 --no-extra
run HC_BAND_ROWS1=32 HC_BAND_WAVES1=8 HC_FRAME_CHUNK=64 python bench.py --steps 20 --cpu-sample 0 --no-extra
cat $out
