#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3z
rm -rf $O; mkdir -p $O
cd $R
bash tools/exp_e2e.sh > $O/e2e.txt 2>&1; cat $O/e2e.txt
cd /tmp && export TMPDIR=/tmp
D=$R/ugproj-detector_amd
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- $D/throughput_demo $D/data/cascades/haarcascade_frontalface_alt.hcc synth:planted 256 720 1280 256 2 pinned 20 > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
tail -2 $O/trace.log
python3 $R/tools/timeline_gaps.py $O/trace > $O/gaps.txt; head -12 $O/gaps.txt
rm -rf $O/trace
