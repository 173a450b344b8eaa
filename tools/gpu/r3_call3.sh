#!/bin/bash
# round 3, GPU call 3: where the per-frame exchange cost goes; dense classification tests + config 3 A/B
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r3c; rm -rf $O; mkdir -p $O
timeout 600 python tools/exchange_cost.py 400 > $O/exchange_cost.jsonl 2>$O/exchange_cost.err; cat $O/exchange_cost.jsonl
timeout 1200 python -m pytest tests/test_dense_cull.py tests/test_dense_512_properties.py -m gpu -x -q > $O/pytest_dense.log 2>&1; tail -15 $O/pytest_dense.log
B="python bench.py --config 3 --no-cpu-baseline --no-extra-legs"
for k in 0 1 2; do
  if [ $k = 0 ]; then $B > $O/bench_c3_m$k.json 2>$O/bench_c3_m$k.err; else ITM_X=1 $B --debug-keys 16=$k > $O/bench_c3_m$k.json 2>$O/bench_c3_m$k.err; fi
done
for f in $O/bench_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], d['roofline']['frac'])" 2>&1 | tail -1)"; done
