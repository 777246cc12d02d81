#!/bin/bash
# frame chunks of k_cascade / the prep kernel at 1080p (config #3, 384 frames) and 2160p (config #5, 128 frames): fps, ms per step, kernel ms
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/chunks35; mkdir -p $O; cd $R
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --cpu-sample 0 --no-extra $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_launch']; print(d['value'], d['ms_per_step'], k['k_pyramid_integral'], k['k_cascade'], d['parity']['bit_exact'])"; }
ARGS="--config 3 --steps 5"
for c in 128 64 32 0; do run HC_FRAME_CHUNK=$c; done 2>&1 | tee $O/c3.txt
for c in 16 8 64; do run HC_PI_FRAME_CHUNK=$c; done 2>&1 | tee -a $O/c3.txt
ARGS="--config 5 --steps 3"
for c in 128 32 16 0; do run HC_FRAME_CHUNK=$c; done 2>&1 | tee $O/c5.txt
for c in 8 16; do run HC_PI_FRAME_CHUNK=$c; done 2>&1 | tee -a $O/c5.txt
