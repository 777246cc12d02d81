#!/bin/bash
# round 2: A/B of k_cascade build variants (one gpurun call).  Each line: bench.py on the default workload with an
# alternative build of the library (HC_LIB_PATH); bench.py itself refuses to print a number if boxes differ from the oracle.
out=gpurun_out/r2b
mkdir -p $out
for v in ${VARIANTS:-"" _SM _OB _OBSM _PF _PF6 _W7 _SM7}; do
  lib=ugproj-detector_amd/libhip_cascade$v.so
  [ -f $lib ] || continue
  echo "start variant $v" >> $out/progress.txt
  HC_LIB_PATH=$PWD/$lib timeout -k 10 180 python bench.py --steps 20 --cpu-sample 0 > $out/bench$v.json 2> $out/bench$v.err || echo "variant $v failed" >> $out/summary.txt
  python - "$v" $out/bench$v.json >> $out/summary.txt <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print("%-6s %8.1f fps  %s  parity %s" % (sys.argv[1] or "base", d["value"], d["roofline"]["all_kernels_ms_per_launch"], d["parity"]["bit_exact"]))
except Exception as e:
    print("%-6s no result: %s" % (sys.argv[1] or "base", e))
PY
done
cat $out/summary.txt
