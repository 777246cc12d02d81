#!/bin/bash
# round 4: what a cascade workgroup more or less per CU is worth (HC_DEBUG_LDS_PAD1/2 pad the dynamic LDS of the scan-step-1 / -2 launch):
# per-kernel average durations from rocprofv3 --kernel-trace --stats
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4_occ
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {
  local label=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$label -- python3 $R/bench.py --steps 10 --cpu-sample 0 --no-extra > $O/$label.log 2>&1
  echo "== $label ($*)" >> $O/summary.txt
  python3 - $O/t_$label >> $O/summary.txt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_cascade" in r["Name"] or "k_pyramid" in r["Name"]:
            print("   %-70s calls %s avg %.3f ms" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
  rm -rf $O/t_$label
}
run base A=1
run ys2_2wg HC_DEBUG_LDS_PAD2=20000      # 43 KB -> 63 KB: 2 workgroups of 8 wavefronts per CU instead of 3
run ys1_7wg HC_DEBUG_LDS_PAD1=2000       # 18.6 KB -> 20.6 KB: 7 workgroups of 4 wavefronts per CU instead of 8
run ys1_6wg HC_DEBUG_LDS_PAD1=5000       # 23.6 KB: 6
cat $O/summary.txt
