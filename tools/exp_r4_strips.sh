#!/bin/bash
# round 4: the column-strip prep kernel -- configs #3 / #5 (extra.configs of the default run), default at 720p, and the strip
# form against the one-workgroup-per-level kernel on config #2 (HC_VAR_SPLIT=2)
out=gpurun_out/r4_strips.txt
: > $out
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_strips.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['ms_per_step'], j['roofline']['all_kernels_ms_per_launch'], j['config']['launch'][0]['variance_route'], j['parity']['bit_exact'])
for k,v in j.get('extra',{}).get('configs',{}).items(): print(' config', k, v.get('value'), v.get('ms_per_step'), v.get('roofline',{}).get('all_kernels_ms_per_launch'), v.get('parity',{}).get('bit_exact'), v.get('error'))
" >> $out; }
run python bench.py --steps 20 --cpu-sample 0
run HC_VAR_SPLIT=2 python bench.py --steps 20 --cpu-sample 0 --no-extra
run HC_VAR_SPLIT=0 python bench.py --steps 20 --cpu-sample 0 --no-extra
run python bench.py --cascade default --steps 10 --cpu-sample 0 --no-extra
run python bench.py --cascade default --semantics cv24 --steps 10 --cpu-sample 0 --no-extra
run python bench.py --config 3 --cascade alt2 --steps 5 --cpu-sample 0 --no-extra
cat $out
