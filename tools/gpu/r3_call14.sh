#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3m; rm -rf $O; mkdir -p $O
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extra-legs"
for rep in 1 2; do
$B > $O/bench_noex_$rep.json 2>$O/bench_noex_$rep.err
$B --force-exchange --exchange-batch 1 > $O/bench_ex1_nccl_$rep.json 2>$O/e.err
$B --force-exchange --exchange-batch 1 --control-backend gloo > $O/bench_ex1_gloo_$rep.json 2>$O/e.err
$B --force-exchange --exchange-batch 8 --control-backend gloo > $O/bench_ex8_gloo_$rep.json 2>$O/e.err
$B --force-exchange --exchange-batch 2 --control-backend gloo > $O/bench_ex2_gloo_$rep.json 2>$O/e.err
done
for f in $O/bench_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['config']['collective_backend'])" 2>&1 | tail -1)"; done
timeout 300 python tools/exchange_cost.py 400 | grep -E "no exchange|full hand-off" 
