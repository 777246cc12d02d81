#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 200 python tools/exp_mid_batch.py 2>/dev/null
for v in "HC_SEG_MAX_BATCH=32" "HC_SEG_MAX_BATCH=64" "HC_SEG_MAX_BATCH=0"; do env $v timeout -k 10 200 python tools/exp_mid_batch.py 2>/dev/null; done
