#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3ex; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_native_exchange.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extra-legs"
for rep in 1 2; do
$B --force-exchange --exchange-batch 1 > $O/bench_ex1_thread_$rep.json 2>$O/e1.err
ITM_EXCHANGE_INLINE=1 $B --force-exchange --exchange-batch 1 > $O/bench_ex1_inline_$rep.json 2>$O/e2.err
$B --force-exchange > $O/bench_ex8_thread_$rep.json 2>$O/e3.err
ITM_EXCHANGE_INLINE=1 $B --force-exchange > $O/bench_ex8_inline_$rep.json 2>$O/e4.err
done
grep -c unavailable $O/*.err
for f in $O/bench_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['config']['exchange_cost_measured']['cost_percent'])" 2>&1 | tail -1 | cut -c1-200)"; done
