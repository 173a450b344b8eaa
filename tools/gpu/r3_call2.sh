#!/bin/bash
# round 3, GPU call 2: device-scope event release (exchange + timers), bench extra legs, origin offset cliff
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r3b; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_native_exchange.py -m gpu -x -q > $O/pytest_exchange.log 2>&1; tail -2 $O/pytest_exchange.log
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extra-legs"
for rep in 1 2; do
$B > $O/bench_noex_$rep.json 2>$O/bench_noex_$rep.err
$B --force-exchange --exchange-batch 1 > $O/bench_ex1_$rep.json 2>$O/bench_ex1_$rep.err
$B --force-exchange --exchange-batch 2 > $O/bench_ex2_$rep.json 2>$O/bench_ex2_$rep.err
$B --force-exchange --exchange-batch 8 > $O/bench_ex8_$rep.json 2>$O/bench_ex8_$rep.err
ITM_EXCHANGE_SYSTEM_SCOPE_EVENTS=1 $B --force-exchange --exchange-batch 1 > $O/bench_ex1sys_$rep.json 2>$O/bench_ex1sys_$rep.err
done
$B --timer-every 1 > $O/bench_timer1.json 2>$O/bench_timer1.err
$B --origin-offset 20,-12,8 > $O/bench_offset.json 2>$O/bench_offset.err
$B --raw-depth > $O/bench_raw.json 2>$O/bench_raw.err
python bench.py > $O/bench_default.json 2>$O/bench_default.err
for c in 3 5; do python bench.py --config $c --no-cpu-baseline > $O/bench_c$c.json 2>$O/bench_c$c.err; done
for f in $O/bench_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], d['config']['exchange'][:60])" 2>&1 | tail -1)"; done
cat $O/bench_default.json
