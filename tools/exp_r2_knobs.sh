#!/bin/bash
# round 2: row-form stages / sparse threshold sweep on the default workload, then the SQ counters of the default build
out=gpurun_out/r2d
mkdir -p $out
for rs in 1 2 3; do for sm in 16 32 64; do
  HC_ROW_STAGES=$rs HC_SPARSE_MAX=$sm timeout -k 10 180 python bench.py --steps 20 --cpu-sample 0 > $out/b.json 2> $out/b.err
  python - $rs $sm $out/b.json >> $out/summary.txt <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    print("row_stages %s sparse_max %s: %8.1f fps  k_cascade %.3f ms  parity %s" % (sys.argv[1], sys.argv[2], d["value"], d["roofline"]["all_kernels_ms_per_launch"]["k_cascade"], d["parity"]["bit_exact"]))
except Exception as e:
    print("row_stages %s sparse_max %s: no result %s" % (sys.argv[1], sys.argv[2], e))
PY
  tail -1 $out/summary.txt
done; done
R=$PWD
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-timing-events"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $R/$out/pmc_sq1 -- python3 $ARGS > $R/$out/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/$out/pmc_sq2 -- python3 $ARGS > $R/$out/pmc_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $R/$out/pmc_f -- python3 $ARGS > $R/$out/pmc_f.log 2>&1
cd $R; python3 tools/pmc_summary.py $out > $out/pmc_summary.txt; head -50 $out/pmc_summary.txt
