#!/bin/bash
# round 4: 32-bit tile fill with four entries per lane (tree) vs the 64-column pieces (-DHC_FILL_PIECES)
out=gpurun_out/r4_fill32.txt
: > $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_legacy.py -m gpu -x -q > gpurun_out/r4_fill32_test.log 2>&1; tail -1 gpurun_out/r4_fill32_test.log >> $out
D=$PWD/ugproj-detector_amd
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_fill32.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['roofline']['all_kernels_ms_per_launch'].get('k_cascade'), j['parity']['bit_exact'])
" >> $out; }
B="python bench.py --steps 12 --cpu-sample 0 --no-extra"
for args in "" "--cascade alt2" "--cascade default" "--semantics cv24" "--config 3 --steps 5"; do
for rep in 1 2; do
run $B $args
run HC_LIB_PATH=$D/libhip_cascade_pieces.so $B $args
done
done
cat $out
