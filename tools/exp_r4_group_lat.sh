#!/bin/bash
# K6 at small batches: synchronous-call latency and small host batches with the device grouping on / off
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/group
for g in 1 0 1 0; do
  echo "== HC_DEVICE_GROUP=$g"
  HC_DEVICE_GROUP=$g python tools/exp_latency.py 2>&1 | grep -v "^$" | tail -6
  HC_DEVICE_GROUP=$g python tools/exp_small_batch.py 2>&1 | tail -1
done | tee gpurun_out/group/latency_ab.txt
