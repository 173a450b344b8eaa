#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3k; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_accel_origin.py -m gpu -x -q > $O/pytest_accel.log 2>&1; tail -25 $O/pytest_accel.log
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; tail -5 $O/pytest_all.log
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extra-legs"
$B > $O/bench.json 2>$O/bench.err
$B --origin-offset 20,-12,8 > $O/bench_offset.json 2>$O/bench_offset.err
for f in $O/bench*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'])" 2>&1 | tail -1)"; done
