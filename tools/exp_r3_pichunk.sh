#!/bin/bash
# round 3: k_pyramid_integral over chunks of frames (HC_PI_FRAME_CHUNK): time, parity, fetch
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3pc; rm -rf $O; mkdir -p $O; cd $R
for c in 32 16 8 0 32 16; do
  HC_PI_FRAME_CHUNK=$c timeout -k 10 300 python bench.py --steps 20 --cpu-sample 0 --no-extra > $O/c$c.json 2> $O/c$c.err || { tail -3 $O/c$c.err; continue; }
  python -c "import json; d=json.load(open('$O/c$c.json')); print('pi chunk $c', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['all_kernels_ms_per_launch'].items() if v > 0.02}, d['parity']['bit_exact'])"
done
cd /tmp && export TMPDIR=/tmp
for c in 16 8; do
  export HC_PI_FRAME_CHUNK=$c
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p$c -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-timing-events --no-extra > $O/p$c.log 2>&1 || { echo "chunk $c pmc failed"; tail -3 $O/p$c.log; continue; }
  python3 $R/tools/traffic_from_pmc.py $O/p$c $O/t$c.json 512 > /dev/null
  python3 -c "import json; d=json.load(open('$O/t$c.json')); print('pi chunk $c fetch', {k: round(v['fetch_bytes_corrected'] / 1e9, 2) for k, v in d['kernels'].items() if v['fetch_bytes_corrected'] > 1e8})"
  rm -rf $O/p$c
done
