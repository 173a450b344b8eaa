#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3l; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; tail -8 $O/pytest_all.log
for i in 1 2 3; do timeout 600 python -m pytest tests/test_tracker.py -m gpu -q -k "four_trackers" 2>&1 | tail -1; done
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extra-legs"
$B --streams-per-gpu 3 > $O/bench_k3.json 2>$O/bench_k3.err
ITM_ONE_PASS_LIST=0 $B --streams-per-gpu 3 > $O/bench_k3_twopass.json 2>$O/bench_k3_twopass.err
GPU_MAX_HW_QUEUES=8 $B --streams-per-gpu 6 > $O/bench_k6_q8.json 2>$O/bench_k6_q8.err
GPU_MAX_HW_QUEUES=8 ITM_ONE_PASS_LIST=1 timeout 120 $B --streams-per-gpu 6 --steps 100 > $O/bench_k6_q8_onepass.json 2>$O/bench_k6_q8_onepass.err
GPU_MAX_HW_QUEUES=8 $B --streams-per-gpu 4 > $O/bench_k4_q8.json 2>$O/bench_k4_q8.err
for f in $O/bench*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"; done
