#!/bin/bash
# round 3: k_cascade walks a large batch in chunks of frames (HC_FRAME_CHUNK): time, parity, HBM fetch per launch (512 frames)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3ch; rm -rf $O; mkdir -p $O; cd $R
for c in 128 0 64 256; do
  HC_FRAME_CHUNK=$c timeout -k 10 300 python bench.py --steps 20 --cpu-sample 0 --no-extra > $O/c$c.json 2> $O/c$c.err || { tail -3 $O/c$c.err; continue; }
  python -c "import json; d=json.load(open('$O/c$c.json')); print('chunk $c', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['all_kernels_ms_per_launch'].items() if v > 0.02}, d['parity']['bit_exact'])"
done
cd /tmp && export TMPDIR=/tmp
for c in 128 0 64; do
  export HC_FRAME_CHUNK=$c
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p$c -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-timing-events --no-extra > $O/p$c.log 2>&1 || { echo "chunk $c pmc failed"; tail -3 $O/p$c.log; continue; }
  echo -n "chunk $c fetch: "; python3 $R/tools/traffic_from_pmc.py $O/p$c $O/t$c.json 512 | python3 -c "import json,sys; d=json.load(sys.stdin); print({k: round(v['fetch_bytes_corrected'] / 1e9, 2) for k, v in d['kernels'].items() if v['fetch_bytes_corrected'] > 1e8})"
  rm -rf $O/p$c
done
