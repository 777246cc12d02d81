#!/bin/bash
# LDS activity / bank conflicts of k_cascade as a function of the evaluated stage depth
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_stage
rm -rf $O; mkdir -p $O
ARGS="$R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-timing-events"
for s in 1 2 3 4 6 8 22; do
  HC_DEBUG_MAX_STAGES=$s rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/s$s -- python3 $ARGS > $O/s$s.log 2>&1
  echo "== max_stages $s"; python3 $R/tools/pmc_summary.py $O/s$s | grep -A8 "k_cascade<false, 16"
done
