#!/bin/bash
# round 3: is a run throttled by the CPU quota of the box?  cgroup cpu.stat before / after (plain, forced-dist RCCL, 2-rank gloo rehearsal)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3d4
rm -rf $O; mkdir -p $O
cd $R
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29536
echo "nproc=$(nproc) cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
stat() { cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -E "nr_throttled|throttled_usec|usage_usec" | tr '\n' ' '; echo " <- $1"; }
for mode in plain dist ranks2; do
  extra=""
  [ $mode = dist ] && extra="--force-dist --backend nccl"
  [ $mode = ranks2 ] && extra="--gpus 2 --backend gloo --share-gpu --batch 128"
  stat "before $mode"
  timeout -k 10 300 python bench.py --steps 30 --cpu-sample 0 --no-extra $extra > $O/$mode.json 2> $O/$mode.err || tail -5 $O/$mode.err
  stat "after $mode"
  head -c 300 $O/$mode.json; echo
done
timeout -k 10 200 python tools/exp_latency.py > $O/latency.txt 2>&1; tail -12 $O/latency.txt
