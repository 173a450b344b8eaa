#!/bin/bash
# Collects what profiles/ holds for this round: rocprofv3 kernel stats of bench.py for configs 2, 3, 5 and the HBM traffic
# counters (FETCH_SIZE / WRITE_SIZE, separate passes) of each config's roofline kernel.
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r2prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 2 3 5; do
  steps=200
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_c$c -o s -- python3 $R/bench.py --config $c --steps $steps --warmup 10 --no-cpu-baseline > $R/$O/stats_c$c.log 2>&1
done
declare -A RX=( [2]="raycast_kernel" [3]="integrate_dense" [5]="integrate_hash_kernel" )
for c in 2 3 5; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "${RX[$c]}" --output-format csv -d $R/$O/pmc_c${c}_$ctr -o p -- python3 $R/bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/pmc_c${c}_$ctr.log 2>&1
  done
done
cd $R
python3 - <<'PY'
import csv, glob, json, collections, subprocess
O = "gpurun_out/r2prof"
out = {}
for c in (2, 3, 5):
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(f"{O}/pmc_c{c}_{ctr}/**/*counter_collection.csv", recursive=True)
        if not files: continue
        rows = [r for r in csv.DictReader(open(files[0])) if r["Counter_Name"] == ctr]
        per_kernel = collections.defaultdict(list)
        for r in rows: per_kernel[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
        name, v = max(per_kernel.items(), key=lambda kv: len(kv[1]))
        vals[ctr] = {"kernel": name, "launches": len(v), "avg_KB": sum(v) / len(v)}
    out[f"config{c}"] = vals
json.dump(out, open(f"{O}/traffic_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete; find $O -name "*counter_collection.csv" -size +2000k -delete
ls -R $O | head -40
for c in 2 3 5; do echo "== config $c"; cut -c1-100 $O/stats_c$c/*kernel_stats.csv | head -8; done
