#!/bin/bash
# round 4: latency of the synchronous calls and small host batches with the strip kernel in the tree; small batches with the fused prep
# pinned on / off (is the cost model of use_fused_prep still right now that the fused kernel marches down strips?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/lat
bash tools/exp_round_latency.sh > /dev/null 2>&1
cp gpurun_out/lat/latency.txt gpurun_out/lat/r4_latency.txt
for fp in 0 1; do HC_FUSED_PREP=$fp python tools/exp_small_batch.py >> gpurun_out/lat/r4_latency.txt 2>&1; done
HC_VAR_SPLIT=0 python tools/exp_small_batch.py >> gpurun_out/lat/r4_latency.txt 2>&1
cat gpurun_out/lat/r4_latency.txt
