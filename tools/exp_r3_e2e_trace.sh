#!/bin/bash
# round 3: kernel + memory-copy timeline of the PCIe-inclusive C++ driver (pinned frames, batch 256, 6 passes per round)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3x
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=$R/ugproj-detector_amd
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- $D/throughput_demo $D/data/cascades/haarcascade_frontalface_alt.hcc synth:planted 256 720 1280 256 3 pinned 6 > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
tail -2 $O/trace.log
python3 $R/tools/timeline_gaps.py $O/trace > $O/gaps.txt; head -12 $O/gaps.txt
python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob("$O/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction",""), r))
rows.sort()
big=[r for r in rows if r[1]-r[0] > 500000]
print("memory copies > 0.5 ms:", len(big))
for r in big[-6:]:
    print("  %s %.3f ms" % (r[2], (r[1]-r[0])/1e6), {k:v for k,v in r[3].items() if k in ("Bytes","Size","Agent_Id")})
PY
rm -rf $O/trace
