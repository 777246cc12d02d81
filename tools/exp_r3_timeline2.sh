#!/bin/bash
# round 3: the idle gaps of config 3 at the context boundaries -- kernel trace + HIP runtime API trace side by side
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3t2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/bench.py --config 3 --steps 4 --warmup 1 --cpu-sample 0 --no-extra > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
ls -R $O/trace | head -20
python3 $R/tools/timeline_api.py $O/trace > $O/api.txt
rm -rf $O/trace
grep -A200 "merged timeline" $O/api.txt
