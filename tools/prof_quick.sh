#!/bin/bash
# usage: prof_quick.sh [ENV=VAL ...]  -- LDS/VALU counters of k_cascade for one configuration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcq
rm -rf $O; mkdir -p $O
ARGS="$R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-timing-events"
env "$@" rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 $ARGS > $O/p1.log 2>&1
python3 $R/tools/pmc_summary.py $O | grep -A9 "k_cascade<false, 16"
