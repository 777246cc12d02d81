#!/bin/bash
# round 4: no cooperative mode on the other cascades / configs / semantics: tree vs nc (6 compiled stages) vs nc12all (12 in the 16-bit units)
out=gpurun_out/r4_coop4.txt
: > $out
D=$PWD/ugproj-detector_amd
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_coop4.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['roofline']['all_kernels_ms_per_launch'].get('k_cascade'), [l.get('compiled_stages') for l in j['config']['launch']], j['parity']['bit_exact'])
" >> $out; }
B="python bench.py --steps 10 --cpu-sample 0 --no-extra"
for args in "--cascade default" "--cascade alt_tree" "--cascade alt2" "--semantics cv24" "--semantics cv24 --cascade default" "--semantics cv24 --cascade alt_tree" "--config 3 --steps 5" "--config 5 --steps 4" "--kind noise" "--kind texture"; do
run $B $args
run HC_LIB_PATH=$D/libhip_cascade_nc.so $B $args
run HC_LIB_PATH=$D/libhip_cascade_nc12all.so $B $args
done
cat $out
