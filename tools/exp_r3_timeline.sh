#!/bin/bash
# round 3: where does the step time of the shared-pass configs go?  Kernel timeline (rocprofv3 --kernel-trace) of
# `bench.py --config 3` and the default config: union of the busy intervals, idle gaps and which kernels bracket them.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3t
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in 3 2; do
  rocprofv3 --kernel-trace --output-format csv -d $O/trace$cfg -- python3 $R/bench.py --config $cfg --steps 6 --warmup 2 --cpu-sample 0 --no-extra > $O/trace$cfg.log 2>&1 || { tail -5 $O/trace$cfg.log; exit 1; }
  python3 $R/tools/timeline_gaps.py $O/trace$cfg > $O/gaps$cfg.txt
  rm -rf $O/trace$cfg
  tail -30 $O/gaps$cfg.txt
done
cd $R && bash tools/exp_e2e.sh > $O/e2e.txt 2>&1; cat $O/e2e.txt
