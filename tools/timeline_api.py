#!/usr/bin/env python3
"""Around the largest idle gaps of a kernel trace: which HIP API calls the host made, and when.  usage: timeline_api.py <trace dir>"""
import csv, glob, sys, re
def load(pat):
    out = []
    for f in glob.glob(sys.argv[1] + "/**/*" + pat, recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
ker = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])[:60]) for r in load("kernel_trace.csv")]
api = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "")) for r in load("hip_api_trace.csv")]
cp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", "")) for r in load("memory_copy_trace.csv")]
ker.sort(); api.sort()
ev = sorted(ker + cp)
t0 = ev[0][0] + (ev[-1][1] - ev[0][0]) * 0.5
ev = [e for e in ev if e[0] >= t0]
gaps = []; cur_e = ev[0][1]; last = ev[0]
for e in ev[1:]:
    if e[0] > cur_e: gaps.append((e[0] - cur_e, cur_e, e[0], last[2], e[2]))
    if e[1] > cur_e: cur_e = e[1]; last = e
gaps.sort(reverse=True)
for g in gaps[:4]:
    print("=== gap %.3f ms after %s before %s" % (g[0] / 1e6, g[3], g[4]))
    lo, hi = g[1] - 3000000, g[2] + 300000
    for a in api:
        if a[1] >= lo and a[0] <= hi:
            print("   api %-28s thread %s  start %+9.3f ms  end %+9.3f ms (relative to the gap's start)" % (a[2], a[3][-5:], (a[0] - g[1]) / 1e6, (a[1] - g[1]) / 1e6))
# merged timeline of the last 100 ms: kernels (K) and the host's long synchronisations / batch starts (H)
tend = ev[-1][1]
lines = []
for k in ker + cp:
    if k[0] >= tend - 100e6: lines.append((k[0], "K  %-62s %8.3f ms" % (k[2], (k[1] - k[0]) / 1e6)))
for a in api:
    if a[0] >= tend - 100e6 and (a[2] == "hipMemsetAsync" or (a[2] in ("hipStreamSynchronize", "hipEventSynchronize") and a[1] - a[0] > 50000)):
        lines.append((a[0], "H      %-58s %8.3f ms" % (a[2], (a[1] - a[0]) / 1e6)))
print("=== merged timeline (last 100 ms)")
for t, s in sorted(lines):
    print("%9.3f  %s" % ((t - (tend - 100e6)) / 1e6, s))
