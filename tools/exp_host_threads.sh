cd $GRAFT_REPO_ROOT
run() { echo -n "$*: "; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1; }
run HC_HOST_THREADS=8
run HC_HOST_THREADS=4
run HC_HOST_THREADS=2
run HC_HOST_THREADS=1
