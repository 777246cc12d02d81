#!/bin/bash
# round 4: 15-row bands where they fit one more workgroup per CU (tree) vs HC_BAND_TRIM=0
out=gpurun_out/r4_trim.txt
: > $out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "trimmed_bands or frame_chunks or compiled_early" > gpurun_out/r4_trim_test.log 2>&1; tail -1 gpurun_out/r4_trim_test.log >> $out
run() { echo "== $*" >> $out; env "$@" 2>>gpurun_out/r4_trim.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['roofline']['all_kernels_ms_per_launch'].get('k_cascade'), j['parity']['bit_exact'])
" >> $out; }
B="python bench.py --steps 12 --cpu-sample 0 --no-extra"
for args in "--cascade default" "--cascade alt2" "--config 3 --steps 5" "--config 5 --steps 4"; do
for rep in 1 2; do
run $B $args
run HC_BAND_TRIM=0 $B $args
done
done
cat $out
