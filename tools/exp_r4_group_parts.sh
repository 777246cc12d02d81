#!/bin/bash
# where k_group_frames spends its time: variants that skip the sort (gs1), the unions (gs2), the whole pair loop (gs4) -- wrong results, timing only.
# The variants: apply tools/patches/r4_group_skip.patch, compile csrc/group_kernels.hip with -DHC_GROUP_SKIP=1 / 2 / 4 and link it with the other objects as libhip_cascade_gs<N>.so
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/gparts; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in "" _gs1 _gs2 _gs4; do
  export HC_LIB_PATH=$R/ugproj-detector_amd/libhip_cascade$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$v -- python3 $R/tools/exp_hosttime.py 512 > $O/log$v.txt 2>&1
  echo "== libhip_cascade$v"; python3 -c "
import csv,glob,sys
for f in glob.glob('$O/t$v/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_group' in r['Name']: print(r['Name'][:22], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
"
  rm -rf $O/t$v
done | tee $O/parts.txt
