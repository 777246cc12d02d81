#!/bin/bash
# round 3: the N > 1 code path on one GPU (process group + RCCL all-gather of box records with one rank) against the plain run;
# timeline of the forced-dist run (does the gather's wait serialise the pipeline?)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3d
rm -rf $O; mkdir -p $O
cd $R
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
for rep in 1 2; do
  timeout -k 10 300 python bench.py --cpu-sample 0 --no-extra > $O/plain_$rep.json 2> $O/plain_$rep.err
  timeout -k 10 300 python bench.py --cpu-sample 0 --no-extra --force-dist --backend nccl > $O/dist_$rep.json 2> $O/dist_$rep.err || tail -5 $O/dist_$rep.err
  python - <<PY
import json
for l in ("plain", "dist"):
    try:
        d = json.load(open("$O/%s_$rep.json" % l))
        print(l, d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_launch"] if "roofline" in d else None)
    except Exception as e:
        print(l, "failed", e)
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-extra --force-dist --backend nccl > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python3 $R/tools/timeline_gaps.py $O/trace > $O/gaps.txt; rm -rf $O/trace; head -14 $O/gaps.txt
