#!/usr/bin/env python3
"""HBM traffic per launch of each kernel from the PMC passes of tools/prof_round.sh, corrected as
MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a streaming read (TCC_EA0_RDREQ tallied at 64 B for
128-B requests) -> doubled; WRITE_SIZE is exact.  Writes profiles/<tag>_traffic.json, which bench.py
reports as roofline.traffic (per launch of frames_per_launch frames, scaled to its own batch size; the two
k_cascade launches of a batch are summed).  usage: traffic_from_pmc.py <pmc dir> <out.json> [frames per launch]"""
import collections
import re
import csv
import glob
import json
import sys

root, out = sys.argv[1], sys.argv[2]
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 64
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            k = re.sub(r"casc_\w+::", "", r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hc::", "")).split("<")[0]
            acc[k][r["Counter_Name"] + "|" + r["Kernel_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in acc.items():
    if not k.startswith("k_"):
        continue
    fetch = sum(sum(v) / len(v) for n, v in d.items() if n.startswith("FETCH_SIZE"))
    write = sum(sum(v) / len(v) for n, v in d.items() if n.startswith("WRITE_SIZE"))
    res[k] = {"fetch_bytes_corrected": int(fetch * 1024 * 2), "write_bytes": int(write * 1024),
              "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024)}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 4, %d frames per launch" % frames,
           "frames_per_launch": frames,
           "correction": "FETCH_SIZE KiB x2 (gfx950), WRITE_SIZE KiB x1", "kernels": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
