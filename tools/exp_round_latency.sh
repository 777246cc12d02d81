#!/bin/bash
# latency of the synchronous calls, small host batches, the PCIe-inclusive batched rate and the batched detectFaces rate at the state of the tree
# (profiles/rNN_latency.txt).  Run on the GPU box: bash tools/exp_round_latency.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/lat
python tools/exp_latency.py > gpurun_out/lat/latency.txt 2>&1
python tools/exp_small_batch.py >> gpurun_out/lat/latency.txt 2>&1
bash tools/exp_e2e.sh >> gpurun_out/lat/latency.txt 2>&1
python tests/tools/faces_throughput.py 2>/dev/null | grep detectFaces >> gpurun_out/lat/latency.txt
cat gpurun_out/lat/latency.txt
