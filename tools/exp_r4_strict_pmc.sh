#!/bin/bash
# round 4: why did strict layers (tools/patches/r3_strict_layers.patch, now -DHC_STRICT_FILL=24) cost +4 % although the bank
# model gives them 14 % fewer LDS cycles in the dense stages?  Both builds with the 32-bit tile on every launch (HC_P16=0),
# timed, then under --pmc (counters only, no tracing).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4_strict
rm -rf $O; mkdir -p $O
D=$R/ugproj-detector_amd
cd $R
for rep in 1 2; do
for v in base strict; do
  lib=$D/libhip_cascade.so; [ $v = strict ] && lib=$D/libhip_cascade_strict.so
  HC_P16=0 HC_LIB_PATH=$lib python bench.py --steps 20 --cpu-sample 0 --no-extra 2>>$O/err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('$v', j['value'], j['roofline']['all_kernels_ms_per_launch']['k_cascade'], j['config']['launch'][0]['variant'], j['parity']['bit_exact'])" >> $O/timing.txt
done
done
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-timing-events --no-extra"
for v in base strict; do
  lib=$D/libhip_cascade.so; [ $v = strict ] && lib=$D/libhip_cascade_strict.so
  export HC_P16=0 HC_LIB_PATH=$lib
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${v}_a -- python3 $ARGS > $O/pmc_${v}_a.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/pmc_${v}_b -- python3 $ARGS > $O/pmc_${v}_b.log 2>&1 || exit 1
  python3 $R/tools/pmc_summary.py $O/pmc_${v}_a > $O/summary_$v.txt
  python3 $R/tools/pmc_summary.py $O/pmc_${v}_b >> $O/summary_$v.txt
  rm -rf $O/pmc_${v}_a $O/pmc_${v}_b
done
cat $O/timing.txt; grep -A12 "k_cascade" $O/summary_base.txt | head -60
