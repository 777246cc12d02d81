#!/bin/bash
# round 4 (after the 16-bit build lost its cooperative mode): launch-shape knobs again
out=gpurun_out/r4_knobs.txt
: > $out
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_knobs.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['roofline']['all_kernels_ms_per_launch'].get('k_cascade'), [(l.get('variant'), l.get('compiled_stages')) for l in j['config']['launch']], j['parity']['bit_exact'])
" >> $out; }
B="python bench.py --steps 20 --cpu-sample 0 --no-extra"
run $B
run HC_BAND_ROWS1=32 HC_BAND_WAVES1=8 $B
run HC_BAND_ROWS2=32 HC_BAND_WAVES2=8 $B
run HC_BAND_ROWS2=16 HC_BAND_WAVES2=4 $B
run HC_ROW_STAGES=1 $B
run HC_ROW_STAGES=3 $B
run HC_SPARSE_MAX=40 $B
run HC_SPARSE_MAX=48 $B
run HC_FRAME_CHUNK=64 $B
run HC_FRAME_CHUNK=128 $B
run $B
cat $out
