#!/bin/bash
# round 2, after the bank-ordered lists: tiling / row-form / sparse knobs on the default workload
out=gpurun_out/r2h
mkdir -p $out
run() {
  env "$@" timeout -k 10 180 python bench.py --steps 20 --cpu-sample 0 > $out/b.json 2> $out/b.err
  python - "$*" $out/b.json >> $out/summary.txt <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print("%-60s %8.1f fps  k_cascade %.3f ms  parity %s" % (sys.argv[1], d["value"], d["roofline"]["all_kernels_ms_per_launch"]["k_cascade"], d["parity"]["bit_exact"]))
except Exception as e:
    print("%-60s no result %s" % (sys.argv[1], e))
PY
  tail -1 $out/summary.txt
}
run HC_ROW_STAGES=2
run HC_ROW_STAGES=3
run HC_SPARSE_MAX=24
run HC_SPARSE_MAX=48
run HC_BAND_ROWS1=32 HC_BAND_WAVES1=8
run HC_BAND_ROWS1=32 HC_BAND_WAVES1=4
run HC_BAND_WAVES1=8
run HC_BAND_ROWS2=32
run HC_BAND_WAVES2=4
run HC_BAND_ROWS1=8
