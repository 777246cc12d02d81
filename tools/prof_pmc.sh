#!/bin/bash
# PMC passes for k_cascade (own runs, no tracing domains mixed in).  Output under gpurun_out/pmc/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
ARGS="$R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-timing-events"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $O/p1 -- python3 $ARGS > $O/p1.log 2>&1 &&
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d $O/p2 -- python3 $ARGS > $O/p2.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_CYCLES --output-format csv -d $O/p3 -- python3 $ARGS > $O/p3.log 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/p4 -- python3 $ARGS > $O/p4.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p5 -- python3 $ARGS > $O/p5.log 2>&1
echo done; ls $O/*/*/ | head -30
