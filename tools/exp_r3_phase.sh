#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for rep in 1 2; do
HC_LIB_PATH=$R/ugproj-detector_amd/libhip_cascade_PH.so timeout -k 10 200 python bench.py --steps 3 --cpu-sample 0 --no-extra 2>&1 >/dev/null | grep "^\[phase" | tail -2 | cut -c1-260
done
HC_LIB_PATH=$R/ugproj-detector_amd/libhip_cascade_PH.so timeout -k 10 200 python bench.py --steps 3 --batch 256 --cpu-sample 0 --no-extra 2>&1 >/dev/null | grep "^\[phase" | tail -2 | cut -c1-260
