cd $GRAFT_REPO_ROOT
run() { echo -n "$*: "; env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['all_kernels_ms_per_launch']['k_cascade'], d['parity']['bit_exact'])" || exit 1; }
run HC_BAND_WAVES1=4
run HC_BAND_WAVES1=8
run HC_SPARSE_MAX=16
run HC_SPARSE_MAX=48
run HC_SPARSE_MAX=64
run HC_ROW_STAGES=2
