R=$PWD; out=gpurun_out/trace2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$out/s2 -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-sample 0 --streams 2 --no-timing-events > $R/$out/s2.log 2>&1
cd $R; ls -R $out | head -20
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/trace2/s2/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    out = open("gpurun_out/trace2/timeline.txt", "w")
    for r in rows[-80:]:
        out.write("%10.3f %10.3f q%s %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
PY
rm -rf $out/s2
