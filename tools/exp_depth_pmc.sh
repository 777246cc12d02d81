#!/bin/bash
# k_cascade cost by cascade depth: time and VALU / LDS counters with the cascade truncated after K stages (HC_DEBUG_MAX_STAGES,
# candidates suppressed).  One gpurun call; writes gpurun_out/depth/summary.txt
R=$PWD; out=gpurun_out/depth; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in "1:" "2:" "3:" "4:" "6:" "8:" "12:" "22:" "22:HC_SPARSE_MAX=0"; do
  K=${cfg%%:*}; X=${cfg#*:}
  tag=k${K}${X:+_nosparse}
  echo "start $tag" >> $R/$out/progress.txt
  env HC_DEBUG_MAX_STAGES=$K $X python3 $R/bench.py --steps 5 --warmup 1 --cpu-sample 0 > $R/$out/t_$tag.json 2> $R/$out/t_$tag.err
  export HC_DEBUG_MAX_STAGES=$K; [ -n "$X" ] && export $X
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $R/$out/pmc_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-timing-events > $R/$out/pmc_$tag.log 2>&1
  unset HC_DEBUG_MAX_STAGES; [ -n "$X" ] && unset ${X%%=*}
  python3 - $R/$out $tag >> $R/$out/summary.txt <<'PY'
import csv, glob, json, sys, collections
root, tag = sys.argv[1], sys.argv[2]
try:
    d = json.load(open("%s/t_%s.json" % (root, tag)))
    ms = d["roofline"]["all_kernels_ms_per_launch"]["k_cascade"]
except Exception as e:
    ms = float("nan")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("%s/pmc_%s/**/*counter_collection.csv" % (root, tag), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_cascade" in r["Kernel_Name"]:
            nw = "nw8" if ", 8," in r["Kernel_Name"].split("(")[0] else "nw4"
            acc[nw][r["Counter_Name"]].append(float(r["Counter_Value"]))
line = "%-14s k_cascade %.3f ms |" % (tag, ms)
for nw in ("nw4", "nw8"):
    m = {c: sum(v) / len(v) for c, v in acc[nw].items()}
    line += " %s VALU %.0fM SALU %.0fM LDS %.0fM ldsact %.0fM conf %.0fM wavecyc %.0fM |" % (
        nw, m.get("SQ_INSTS_VALU", 0) / 1e6, m.get("SQ_INSTS_SALU", 0) / 1e6, m.get("SQ_INSTS_LDS", 0) / 1e6,
        m.get("SQ_LDS_IDX_ACTIVE", 0) / 1e6, m.get("SQ_LDS_BANK_CONFLICT", 0) / 1e6, m.get("SQ_WAVE_CYCLES", 0) / 1e6)
print(line)
PY
  rm -rf $R/$out/pmc_$tag
done
cat $R/$out/summary.txt
