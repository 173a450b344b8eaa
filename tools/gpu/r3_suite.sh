#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3suite; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|FAILED" $O/pytest_gpu.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench_default.json 2>$O/bench_default.err; python -c "import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], d['roofline']['frac'], d['with_h2d_raw_depth']['value'], d['table_walk_fallback'])"
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_c2 -o s -- python3 $R/bench.py --config 2 --steps 200 --warmup 10 --no-cpu-baseline --no-extra-legs > $R/$O/stats_c2.log 2>&1
cd $R; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete
python3 - <<'PY'
import csv
for i,r in enumerate(csv.DictReader(open('gpurun_out/r3suite/stats_c2/s_kernel_stats.csv'))):
    if i<6: print("%-62s calls %5s avg_us %9.2f" % (r['Name'][:62], r['Calls'], float(r['AverageNs'])/1e3))
PY
