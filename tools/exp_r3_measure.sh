#!/bin/bash
# round 3: where the time goes after the variance ring and the sliced finish (one box)
R=$PWD; out=gpurun_out/r3d; mkdir -p $out
echo "== phase shares (HC_PHASE build, generic kernel)" > $out/summary.txt
HC_LIB_PATH=$R/ugproj-detector_amd/libhip_cascade_PH.so timeout -k 10 200 python bench.py --steps 3 --cpu-sample 0 --no-extra 2> $out/phase.err > $out/phase.json
grep "^\[phase" $out/phase.err | tail -2 >> $out/summary.txt
echo "== prep without the fused variance phase (HC_FUSE_VARIANCE=0)" >> $out/summary.txt
HC_FUSE_VARIANCE=0 timeout -k 10 200 python bench.py --steps 20 --cpu-sample 0 --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['all_kernels_ms_per_launch'])" >> $out/summary.txt
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-timing-events --no-extra"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $R/$out/pmc_sq1 -- python3 $ARGS > $R/$out/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/$out/pmc_sq2 -- python3 $ARGS > $R/$out/pmc_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $R/$out/pmc_f -- python3 $ARGS > $R/$out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$out/pmc_w -- python3 $ARGS > $R/$out/pmc_w.log 2>&1
cd $R; python3 tools/pmc_summary.py $out > $out/pmc_summary.txt; rm -rf $out/pmc_sq1 $out/pmc_sq2 $out/pmc_f $out/pmc_w
cat $out/summary.txt
