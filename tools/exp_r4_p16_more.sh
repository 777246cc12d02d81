#!/bin/bash
# round 4: the 16-bit-tile build for default and alt_tree (scan-step-1 launch): on (default) / off (HC_P16=0), 720p, and configs #3 / #5
out=gpurun_out/r4_p16_more.txt
: > $out
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_p16_more.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['ms_per_step'], j['roofline']['all_kernels_ms_per_launch']['k_cascade'], [l['variant'] for l in j['config']['launch']], j['parity']['bit_exact'])
" >> $out; }
for c in default alt_tree; do
for rep in 1 2; do
run python bench.py --cascade $c --steps 10 --warmup 2 --cpu-sample 0 --no-extra
run HC_P16=0 python bench.py --cascade $c --steps 10 --warmup 2 --cpu-sample 0 --no-extra
done
done
run python bench.py --config 3 --steps 5 --cpu-sample 0 --no-extra
run HC_P16=0 python bench.py --config 3 --steps 5 --cpu-sample 0 --no-extra
run python bench.py --config 5 --steps 5 --cpu-sample 0 --no-extra
run HC_P16=0 python bench.py --config 5 --steps 5 --cpu-sample 0 --no-extra
cat $out
