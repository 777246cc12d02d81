#!/bin/bash
# HBM fetch of k_cascade per launch at 128 / 256 frames per launch, compiled stages on, and at 128 with the generic kernel.
# FETCH_SIZE alone per pass (with WRITE_SIZE in the same pass the counter request exceeds the hardware and rocprofv3 aborts).
R=$PWD; out=gpurun_out/f128; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # tag batch
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/p$1 -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-timing-events --batch $2 > $R/$out/p$1.log 2>&1 || { echo "$1 failed" >> $R/$out/summary.txt; return; }
  echo -n "$1: " >> $R/$out/summary.txt
  python3 $R/tools/traffic_from_pmc.py $R/$out/p$1 $R/$out/t$1.json $2 | tr -d '\n ' >> $R/$out/summary.txt; echo >> $R/$out/summary.txt
  rm -rf $R/$out/p$1
}
run spec128 128
run spec256 256
export HC_SPEC=0
run generic128 128
cat $R/$out/summary.txt
