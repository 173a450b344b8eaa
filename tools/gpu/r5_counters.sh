#!/bin/bash
# round 5: PMC traffic + SQ counters of the kernels that are >= 15 % of a BASELINE config's frame but were never
# profiled -- config 3's DENSE ray cast, config 5's ray cast and request kernel -- and the per-launch durations of config 2's
# visible-list launch against the frame index (which frames produce the long ones).  PMC counters in their own passes, kernel trace only.
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r5counters; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM"
for job in "3 raycast_kernel c3ray" "5 raycast_kernel c5ray" "5 request_kernel c5req" "2 visible_list_kernel c2list" "2 integrate_project_kernel c2int"; do
  set -- $job; c=$1; rx=$2; tag=$3
  for pass in FETCH_SIZE WRITE_SIZE a c; do
    case $pass in a) P="$P1";; c) P="$P3";; *) P="$pass";; esac
    rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "$rx" --output-format csv -d $R/$O/pmc_${tag}_$pass -o p -- python3 $R/bench.py --config $c --steps 20 --warmup 5 --min-measured-s 0 --no-cpu-baseline --no-extra-legs --timer-frames 1 > $R/$O/pmc_${tag}_$pass.log 2>&1
  done
done
# per-launch durations of the whole frame, config 2, 300 frames behind 20 of warm-up (kernel trace only)
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_c2 -o t -- python3 $R/bench.py --config 2 --steps 300 --warmup 20 --no-cpu-baseline --no-extra-legs --timer-frames 1 > $R/$O/trace_c2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, json, statistics
O = "gpurun_out/r5counters"; out = {}
for d in sorted(glob.glob(O + "/pmc_*_*")):
    if not d.split("/")[-1].startswith("pmc_") or d.endswith(".log"): continue
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files: print(d, "no counter file"); continue
    acc = collections.defaultdict(float); n = collections.defaultdict(int); kn = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1; kn[row["Kernel_Name"][:90]] += 1
    tag = d.split("pmc_")[1].rsplit("_", 1)[0]
    if tag.endswith("_FETCH") or tag.endswith("_WRITE"): tag = tag.rsplit("_", 1)[0]
    e = out.setdefault(tag, {})
    e.update({k: round(acc[k] / max(1, n[k]), 1) for k in acc}); e["launches"] = max(n.values()) if n else 0; e["kernels"] = dict(kn.most_common(3))
json.dump(out, open(O + "/counters.json", "w"), indent=1); print(json.dumps(out, indent=1))
# the visible-list launch against the frame index
files = glob.glob(O + "/trace_c2/**/*kernel_trace.csv", recursive=True)
if files:
    rows = [r for r in csv.DictReader(open(files[0]))]
    per = collections.defaultdict(list)
    for r in rows: per[r["Kernel_Name"].split("(")[0][:60]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    summary = {}
    for k, v in per.items():
        v.sort(); d = [(b - a) / 1e3 for a, b in v]
        summary[k] = {"launches": len(d), "avg_us": round(sum(d) / len(d), 2), "min_us": round(min(d), 2), "max_us": round(max(d), 2), "stdev_us": round(statistics.pstdev(d), 2)}
    name = [k for k in per if "visible_list_kernel" in k]
    tab = {}
    if name:
        v = sorted(per[name[0]]); d = [(b - a) / 1e3 for a, b in v]
        # launches of the timed legs: the LAST 300 + 20 + ... are the timed region; index launches from the end so that frame numbers line up
        tab["per_launch_us"] = [round(x, 2) for x in d]
    json.dump({"kernels": summary, **tab}, open(O + "/trace_c2_summary.json", "w"), indent=1)
    for k, s in sorted(summary.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"])[:8]: print("%-62s" % k, s)
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete; find $O -name "*counter_collection.csv" -delete
