#!/bin/bash
# round 3 probe: every corner of the compiled stages 0-1 read twice (HC_GEN_DUP_READS=2 build) -- how much of the kernel
# time follows the LDS cycles of the row-form stages?
set -e
mkdir -p gpurun_out/r3k
D=$PWD/ugproj-detector_amd
for rep in 1 2; do
for v in base dup2; do
  if [ $v = base ]; then L=$D/libhip_cascade.so; else L=$D/libhip_cascade_dup2.so; fi
  HC_LIB_PATH=$L timeout -k 10 200 python bench.py --steps 20 --cpu-sample 0 --no-extra > gpurun_out/r3k/bench_${v}_$rep.json 2> gpurun_out/r3k/err_${v}_$rep.txt || { tail -3 gpurun_out/r3k/err_${v}_$rep.txt; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/r3k/bench_${v}_$rep.json'))
print('$v $rep fps=%.0f k_cascade=%.3f parity=%s' % (d['value'], d['roofline']['all_kernels_ms_per_launch']['k_cascade'], d['parity']['bit_exact']))"
done
done
