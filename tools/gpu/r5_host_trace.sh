#!/bin/bash
# GPU timeline of frames that arrive from page-locked host memory (main_engine_demo --bench-map-host): where does the time between frames go?
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r5hosttrace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/t -o t -- $R/tests/cpp/main_engine_demo --bench-map-host 400 > $R/$O/run.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, statistics
f = glob.glob("gpurun_out/r5hosttrace/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]          # steady state
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void itm::", "").replace("itm::", "")[:40]
# per frame: from one request_kernel start to the next
starts = [i for i, r in enumerate(rows) if "request_kernel" in r["Kernel_Name"]]
per = []
for a, b in zip(starts[:-1], starts[1:]):
    fr = rows[a:b]; t0 = int(fr[0]["Start_Timestamp"])
    per.append([(name(r), (int(r["Start_Timestamp"]) - t0) / 1000.0, (int(r["End_Timestamp"]) - t0) / 1000.0, r.get("Queue_Id", "")) for r in fr] + [("next frame", (int(rows[b]["Start_Timestamp"]) - t0) / 1000.0, 0, "")])
print("frames", len(per), "period us: median %.1f" % statistics.median(p[-1][1] for p in per))
mid = per[len(per) // 2]
for k in mid: print("  %-42s start %7.1f end %7.1f  queue %s" % k)
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
