#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3p; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_frame_ahead.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|^E" | cut -c1-300 | head -20
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extra-legs"
for rep in 1 2; do
$B > $O/bench_ahead_$rep.json 2>$O/e.err
$B --no-lookahead > $O/bench_plain_$rep.json 2>$O/e.err
done
python bench.py --config 5 --no-cpu-baseline --no-extra-legs > $O/bench_c5_ahead.json 2>$O/e.err
python bench.py --config 5 --no-cpu-baseline --no-extra-legs --no-lookahead > $O/bench_c5_plain.json 2>$O/e.err
$B --streams-per-gpu 3 > $O/bench_k3_ahead.json 2>$O/e.err
for f in $O/bench_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'])" 2>&1 | tail -1)"; done
