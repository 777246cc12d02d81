#!/bin/bash
# round 4: cumulative k_cascade time by cascade depth (HC_DEBUG_MAX_STAGES truncates the cascade: timing only, results are wrong by
# construction) for default, alt2 and alt at 720p, 512 frames per launch
out=gpurun_out/r4_depth.txt
: > $out
for c in default alt2 alt; do
  echo "== $c" >> $out
  for s in 1 2 3 4 6 8 12 16 99; do
    echo -n "stages<=$s: " >> $out
    HC_DEBUG_MAX_STAGES=$s python bench.py --cascade $c --steps 5 --warmup 1 --cpu-sample 0 --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['all_kernels_ms_per_launch']['k_cascade'])" >> $out
  done
done
cat $out
