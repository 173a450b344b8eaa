#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3n; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_main_engine.py tests/test_cpp_adapter.py -m gpu -q 2>&1 | grep -E "passed|failed|^E" | cut -c1-300 | head -20
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extra-legs"
$B --force-exchange > $O/bench_ex8.json 2>$O/e.err
$B --force-exchange --exchange-batch 1 > $O/bench_ex1.json 2>$O/e.err
for f in $O/bench_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['config']['exchange_cost_measured'])" 2>&1 | tail -1)"; done
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_ex1 -o s -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extra-legs --force-exchange --exchange-batch 1 > $R/$O/stats_ex1.log 2>&1
cd $R; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete
cut -c1-120 $O/stats_ex1/s_kernel_stats.csv | head -12
