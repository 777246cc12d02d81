#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc2
rm -rf $O; mkdir -p $O
ARGS="$R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-timing-events"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $O/p1 -- python3 $ARGS > $O/p1.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/p2 -- python3 $ARGS > $O/p2.log 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS --output-format csv -d $O/p3 -- python3 $ARGS > $O/p3.log 2>&1
python3 $R/tools/pmc_summary.py $O | grep -A24 "k_cascade"
