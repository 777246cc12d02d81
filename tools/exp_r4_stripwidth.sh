#!/bin/bash
# round 4: width of the column strips (pixel columns a strip computes at most) -- config #2 through the strip kernel (HC_VAR_SPLIT=2)
# against the one-workgroup-per-level kernel, and configs #3 / #5 at several widths
out=gpurun_out/r4_stripwidth.txt
: > $out
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_stripwidth.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['ms_per_step'], j['roofline']['all_kernels_ms_per_launch'], j['config']['launch'][0]['variance_route'], j['parity']['bit_exact'])
" >> $out; }
run python bench.py --steps 20 --cpu-sample 0 --no-extra
for w in 448 576 640 832 1024; do
run HC_VAR_SPLIT=2 HC_STRIP_COMP_MAX=$w python bench.py --steps 20 --cpu-sample 0 --no-extra
done
for w in 448 576 640 768; do
run HC_STRIP_COMP_MAX=$w python bench.py --config 3 --steps 5 --cpu-sample 0 --no-extra
run HC_STRIP_COMP_MAX=$w python bench.py --config 5 --steps 5 --cpu-sample 0 --no-extra
done
cat $out
