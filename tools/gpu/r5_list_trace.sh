#!/bin/bash
# per-launch durations of the visible-list launch on the final build: config 2, 300 frames behind 20 of warm-up (kernel trace only)
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r5listtrace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_c2 -o t -- python3 $R/bench.py --config 2 --steps 300 --warmup 20 --no-cpu-baseline --no-extra-legs --timer-frames 1 > $R/$O/trace_c2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, json, statistics
f = glob.glob("gpurun_out/r5listtrace/trace_c2/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "visible_list_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0 for r in rows]
json.dump({"what": "visible_list_kernel<true,true,true>, one entry per launch in launch order (frame 0 = fresh scene)", "us": us}, open("gpurun_out/r5listtrace/list_launches.json", "w"))
st = us[100:320]
q = statistics.quantiles(st, n=10)
print("launches", len(us), "frame 0", us[0], "steady state (frames 100-319): mean %.2f median %.2f p90 %.2f max %.2f min %.2f" % (statistics.mean(st), statistics.median(st), q[8], max(st), min(st)))
print("phases above 14 us:", sorted(set(k % 100 for k in range(100, 320) if us[k] > 14.0)))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
