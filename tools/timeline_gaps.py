#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace run: busy union, idle gaps, overlap.  usage: timeline_gaps.py <trace dir>"""
import csv, glob, sys, re
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])[:70], r.get("Queue_Id", "")))
rows.sort()
if not rows:
    print("no kernels"); sys.exit(1)
# steady part: the last 60 % of the trace
t0 = rows[0][0] + (rows[-1][1] - rows[0][0]) * 0.4
rows = [r for r in rows if r[0] >= t0]
span = rows[-1][1] - rows[0][0]
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]; gaps = []; last = rows[0]
ksum = {}
for r in rows:
    ksum[r[2]] = ksum.get(r[2], 0) + (r[1] - r[0])
for r in rows[1:]:
    if r[0] > cur_e:
        gaps.append((r[0] - cur_e, last[2], r[2]))
        busy += cur_e - cur_s; cur_s, cur_e = r[0], r[1]
    if r[1] > cur_e:
        cur_e = r[1]; last = r
busy += cur_e - cur_s
print("span %.2f ms, busy union %.2f ms (%.1f %%), sum of kernel durations %.2f ms (overlap factor %.2f)" % (span / 1e6, busy / 1e6, 100.0 * busy / span, sum(ksum.values()) / 1e6, sum(ksum.values()) / busy))
print("idle: %.2f ms in %d gaps; gaps > 20 us: %d totalling %.2f ms" % ((span - busy) / 1e6, len(gaps), sum(1 for g in gaps if g[0] > 20000), sum(g[0] for g in gaps if g[0] > 20000) / 1e6))
agg = {}
for g in gaps:
    k = (g[1], g[2]); a = agg.setdefault(k, [0, 0]); a[0] += g[0]; a[1] += 1
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
    print("  %8.3f ms in %4d gaps  after %-60s before %s" % (a[0] / 1e6, a[1], k[0], k[1]))
print("kernel time shares:")
for k, v in sorted(ksum.items(), key=lambda kv: -kv[1])[:10]:
    print("  %8.2f ms  %s" % (v / 1e6, k))
