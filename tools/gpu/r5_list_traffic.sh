#!/bin/bash
# PMC traffic of the visible-list launch after it began to share the excess region's re-tests (configs 2 and 5); counters in their own passes
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r5listtraffic; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 2 5; do for P in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $P --kernel-include-regex visible_list_kernel --output-format csv -d $R/$O/pmc_c${c}_$P -o p -- python3 $R/bench.py --config $c --steps 20 --warmup 5 --min-measured-s 0 --no-cpu-baseline --no-extra-legs --timer-frames 1 > $R/$O/pmc_c${c}_$P.log 2>&1
done; done
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r5listtraffic/pmc_*")):
    if d.endswith(".log"): continue
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no file"); continue
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for row in csv.DictReader(open(f[0])): acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    print(d.split("/")[-1], {k: (round(acc[k] / n[k], 1), n[k]) for k in acc})
PY
find $O -name "*.db" -delete
