#!/bin/bash
# round 3: forced-dist (RCCL, one rank) run: kernel trace + HIP API trace, merged timeline
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3d2
rm -rf $O; mkdir -p $O
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29534
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 14 --warmup 2 --cpu-sample 0 --no-extra --force-dist --backend nccl > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python3 $R/tools/timeline_api.py $O/trace > $O/api.txt
python3 - <<PY
import csv, glob, re
api=[]
for f in glob.glob("$O/trace/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id","")))
api.sort()
tend=api[-1][1]
print("=== host calls longer than 3 ms in the last 700 ms")
for a in api:
    if a[0] > tend-700e6 and a[1]-a[0] > 3000000:
        print("%9.3f  %-34s %8.3f ms thread %s" % ((a[0]-(tend-700e6))/1e6, a[2], (a[1]-a[0])/1e6, a[3][-6:]))
ker=[]
for f in glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ker.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])[:50], r.get("Queue_Id",""), r.get("Stream_Id","")))
ker.sort()
print("=== kernel gaps > 5 ms in the last 700 ms (time, gap, previous kernel [queue], next kernel [queue])")
for a, b in zip(ker, ker[1:]):
    if b[0] > tend-700e6 and b[0]-a[1] > 5000000:
        print("%9.3f  gap %7.3f ms  %s [q%s]  ->  %s [q%s]" % ((a[1]-(tend-700e6))/1e6, (b[0]-a[1])/1e6, a[2], a[3], b[2], b[3]))
g=[(b[0]-a[1], a, b) for a, b in zip(ker, ker[1:]) if b[0] > tend-700e6 and b[0]-a[1] > 20000000]
if g:
    gap, a, b = g[0]
    print("=== around the first long gap: gap start = 0; kernels (K) and ALL api calls (H) from -1 ms to gap end + 1 ms")
    ev=[(k[0], "K q%s %-50s %.3f ms" % (k[3], k[2], (k[1]-k[0])/1e6)) for k in ker if k[1] >= a[1]-1e6 and k[0] <= b[0]+1e6]
    ev+=[(x[0], "H %-30s %.3f ms thread %s" % (x[2], (x[1]-x[0])/1e6, x[3][-5:])) for x in api if x[1] >= a[1]-1e6 and x[0] <= b[0]+1e6]
    for t, txt in sorted(ev): print("%10.3f  %s" % ((t-a[1])/1e6, txt))
qs={}
for k in ker: qs.setdefault((k[3], k[2]), 0); qs[(k[3], k[2])]+=1
print("kernels by queue:", sorted(qs.items())[:40])
PY
rm -rf $O/trace

