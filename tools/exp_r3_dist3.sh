#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3d3
rm -rf $O; mkdir -p $O
cd $R
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29535
HC_BENCH_TRACE=1 timeout -k 10 300 python bench.py --steps 30 --cpu-sample 0 --no-extra --force-dist --backend nccl > $O/dist.json 2> $O/dist.err
grep "\[trace\]" $O/dist.err | tail -34
tail -1 $O/dist.json | cut -c1-300
