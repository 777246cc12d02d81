#!/bin/bash
# PCIe-inclusive throughput of the batched C ABI from plain C++ (hip_cascade/throughput_demo.cpp) on 256
# 1280x720 'planted' frames for the three host frame pools: batch 128, one pass (two batches per round: fill and
# drain dominate) and batch 256, twenty passes per round (steady state: 5120 frames).  Run on the GPU box:  bash tools/exp_e2e.sh
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, importlib
sys.path.insert(0, "tests")
import conftest
conftest.load_package()
synth = importlib.import_module("ugproj_detector_amd.synth")
fr = synth.make_frames("planted", 16, 720, 1280)
import numpy as np
np.concatenate([fr] * 16).tofile("/tmp/frames720.u8")
PY
for pool in pageable registered pinned; do
  timeout -k 10 200 ugproj-detector_amd/throughput_demo ugproj-detector_amd/data/cascades/haarcascade_frontalface_alt.hcc /tmp/frames720.u8 256 720 1280 128 6 $pool | tail -1 || exit 1
  timeout -k 10 200 ugproj-detector_amd/throughput_demo ugproj-detector_amd/data/cascades/haarcascade_frontalface_alt.hcc /tmp/frames720.u8 256 720 1280 256 4 $pool 20 | tail -1 || exit 1
done
