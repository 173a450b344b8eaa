#!/bin/bash
# round 3, GPU call 1: real RCCL communicator at world 1 -- tests, per-frame vs batched exchange on the real collective, rocprofv3
# kernel stats of the exchange run, and the stream-copy peak of this box.
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r3a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_native_exchange.py -m gpu -x -q > $O/pytest_exchange.log 2>&1; tail -3 $O/pytest_exchange.log
./tools/microbench/stream_copy 4096 > $O/stream_copy.json 2>&1; cat $O/stream_copy.json
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline"
for rep in 1 2; do
$B > $O/bench_noex_$rep.json 2>$O/bench_noex_$rep.err
$B --force-exchange --exchange-batch 1 > $O/bench_ex1_$rep.json 2>$O/bench_ex1_$rep.err
$B --force-exchange --exchange-batch 8 > $O/bench_ex8_$rep.json 2>$O/bench_ex8_$rep.err
ITM_EXCHANGE_DEVICE_COPY=1 $B --force-exchange --exchange-batch 1 > $O/bench_ex1copy_$rep.json 2>$O/bench_ex1copy_$rep.err
$B --force-exchange --exchange-batch 1 --exchange-impl torch > $O/bench_ex1torch_$rep.json 2>$O/bench_ex1torch_$rep.err
done
for f in $O/bench_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['config']['exchange'][:90])" 2>&1 | tail -1)"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_ex1 -o s -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --force-exchange --exchange-batch 1 > $R/$O/stats_ex1.log 2>&1
cd $R
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete
cut -c1-140 $O/stats_ex1/*kernel_stats.csv | head -12
