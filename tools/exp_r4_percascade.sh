#!/bin/bash
# round 4: per-cascade rates on config #2's frames, both generations of the call (one JSON line each -> gpurun_out/r4_percascade.txt)
out=gpurun_out/r4_percascade.txt
: > $out
for c in alt alt2 default alt_tree; do
  for sem in cv4 cv24; do
    echo "== $c $sem" >> $out
    python bench.py --cascade $c --semantics $sem --steps 10 --warmup 2 --cpu-sample 0 --no-extra 2>>gpurun_out/r4_percascade.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['ms_per_step'], j['roofline']['all_kernels_ms_per_launch'], j['config']['launch'], j['parity']['bit_exact'])" >> $out
  done
done
cat $out
