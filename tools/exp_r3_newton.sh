#!/bin/bash
# round 3: one Newton step instead of two behind v_rsq_f64 in the variance factor (the acceptance test keeps the result exact either way)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3n1
rm -rf $O; mkdir -p $O
cd $R
for rep in 1 2; do
  for lib in n1 std; do
    L=$R/ugproj-detector_amd/libhip_cascade.so; [ $lib = n1 ] && L=$R/ugproj-detector_amd/libhip_cascade_n1.so
    HC_LIB_PATH=$L timeout -k 10 300 python bench.py --steps 20 --cpu-sample 0 --no-extra > $O/${lib}_$rep.json 2> $O/${lib}_$rep.err || { tail -5 $O/${lib}_$rep.err; exit 1; }
    python -c "import json; d=json.load(open('$O/${lib}_$rep.json')); print('$lib', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['all_kernels_ms_per_launch'].items() if v > 0.02}, d['parity']['bit_exact'])"
  done
done
