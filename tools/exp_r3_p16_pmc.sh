#!/bin/bash
R=$PWD; out=gpurun_out/r3m; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-timing-events --no-extra"
for v in ${VARIANTS:-1 0}; do
export HC_P16=$v
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $R/$out/p${v}_sq1 -- python3 $ARGS > $R/$out/p${v}_sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $R/$out/p${v}_sq2 -- python3 $ARGS > $R/$out/p${v}_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/$out/p${v}_f -- python3 $ARGS > $R/$out/p${v}_f.log 2>&1
done
cd $R
for v in ${VARIANTS:-1 0}; do mkdir -p $out/sum$v; mv $out/p${v}_sq1 $out/p${v}_sq2 $out/p${v}_f $out/sum$v/; python3 tools/pmc_summary.py $out/sum$v > $out/pmc_p16_$v.txt; rm -rf $out/sum$v; done
grep -A18 "k_cascade<false, 16, 4" $out/pmc_p16_1.txt | head -20
