#!/bin/bash
# k_cascade with G adjacent bands of a frame dispatched back to back on one XCD (HC_BAND_GROUP): time, parity, HBM fetch
R=$PWD; out=gpurun_out/bg; mkdir -p $out
for g in 0 2 4 8; do
  HC_BAND_GROUP=$g python3 bench.py --steps 20 --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('G=$g',d['value'],d['ms_per_step'],d['roofline']['all_kernels_ms_per_launch']['k_cascade'],d['parity']['bit_exact'])" >> $out/summary.txt
done
cd /tmp && export TMPDIR=/tmp
for g in 0 2 4; do
  export HC_BAND_GROUP=$g
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/p$g -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-timing-events > $R/$out/p$g.log 2>&1 || { echo "G=$g pmc failed" >> $R/$out/summary.txt; continue; }
  echo -n "G=$g fetch: " >> $R/$out/summary.txt
  python3 $R/tools/traffic_from_pmc.py $R/$out/p$g $R/$out/t$g.json 256 | tr -d '\n ' >> $R/$out/summary.txt; echo >> $R/$out/summary.txt
  rm -rf $R/$out/p$g
done
cat $R/$out/summary.txt
