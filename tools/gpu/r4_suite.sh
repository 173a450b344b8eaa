#!/bin/bash
# round 4: the GPU suite, the smoke check, the default bench line and its rocprofv3 kernel statistics (one gpurun call)
cd "$(dirname "$0")/../.."
O=gpurun_out/r4suite; rm -rf $O; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q ${PYTEST_ARGS} > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|FAILED|Error" $O/pytest_gpu.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench_default.json 2>$O/bench_default.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/r4suite/bench_default.json'))
r = d['roofline']
print('value', d['value'], 'ms', d['ms_per_step'], 'reps', d['repetitions'], '| entry', d.get('process_frame_entry_point', {}).get('value'), 'ahead', d.get('with_lookahead', {}).get('value'))
print('raycast us', r['avg_kernel_us'], 'bracket', r['avg_bracket_us'], 'pair', r['event_pair_us'], 'frac', r['frac'], 'n', r['launches_timed'], r.get('other_kernels'))
PY
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/bench_short.json 2>$O/bench_short.err; python -c "
import json; d=json.load(open('$O/bench_short.json')); print('short run:', d['value'], d['ms_per_step'], d['repetitions'], d['roofline']['avg_kernel_us'], d['roofline']['launches_timed'])"
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_c2 -o s -- python3 $R/bench.py --config 2 --steps 200 --warmup 10 --no-cpu-baseline --no-extra-legs --timer-frames 1 > $R/$O/stats_c2.log 2>&1
cd $R; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete
python3 - <<'PY'
import csv
for i,r in enumerate(csv.DictReader(open('gpurun_out/r4suite/stats_c2/s_kernel_stats.csv'))):
    if i<8: print("%-62s calls %5s avg_us %9.2f" % (r['Name'][:62], r['Calls'], float(r['AverageNs'])/1e3))
PY
