#!/bin/bash
# usage: prof_quick.sh <kernel-substring> [ENV=VAL ...]  -- SQ counters of one kernel for one configuration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; shift
O=$R/gpurun_out/pmcq
rm -rf $O; mkdir -p $O
ARGS="$R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-timing-events"
env "$@" rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 $ARGS > $O/p1.log 2>&1
env "$@" rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_CYCLES SQ_WAVES --output-format csv -d $O/p2 -- python3 $ARGS > $O/p2.log 2>&1
python3 $R/tools/pmc_summary.py $O | grep -A17 "$K"
