R=$PWD; out=gpurun_out/r2i; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-timing-events"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $R/$out/pmc_sq1 -- python3 $ARGS > $R/$out/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/$out/pmc_sq2 -- python3 $ARGS > $R/$out/pmc_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/$out/pmc_f -- python3 $ARGS > $R/$out/pmc_f.log 2>&1
cd $R; python3 tools/pmc_summary.py $out > $out/pmc_summary.txt; rm -rf $out/pmc_sq1 $out/pmc_sq2 $out/pmc_f; grep -A20 "casc_int::k_cascade<false, 16, 4" $out/pmc_summary.txt | head -22
