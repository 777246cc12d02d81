#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3ft
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tests/tools/faces_throughput.py > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
grep detectFaces $O/trace.log
python3 $R/tools/timeline_gaps.py $O/trace > $O/gaps.txt; rm -rf $O/trace; head -30 $O/gaps.txt
