#!/bin/bash
# round 4 (after the 16-bit build lost its cooperative mode): wavefront priorities and finish group count again
out=gpurun_out/r4_prio.txt
: > $out
D=$PWD/ugproj-detector_amd
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_prio.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['roofline']['all_kernels_ms_per_launch'].get('k_cascade'), j['parity']['bit_exact'])
" >> $out; }
B="python bench.py --steps 20 --cpu-sample 0 --no-extra"
for rep in 1 2; do
run $B
for n in dp0 dp2 sp1 sg2; do run HC_LIB_PATH=$D/libhip_cascade_$n.so $B; done
done
cat $out
