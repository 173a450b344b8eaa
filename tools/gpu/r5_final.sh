#!/bin/bash
# round 5 final measurements: the whole GPU suite, smoke, the bench lines of configs 2 / 3 / 5 (default runs: parity_check, cpu_baseline, extra legs), the driver-like
# short run, k streams, exchange, the reference binding's rate, closed loops, then rocprofv3 kernel stats of each config (-> profiles/r5_*)
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r5final; rm -rf $O; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|FAILED" $O/pytest_gpu.log | tail -5
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; echo "bench default rc $?"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_like.json 2>$O/e.err
timeout 900 python bench.py --config 3 > $O/bench_c3.json 2>$O/bench_c3.err; echo "bench c3 rc $?"
timeout 900 python bench.py --config 5 > $O/bench_c5.json 2>$O/bench_c5.err; echo "bench c5 rc $?"
B="timeout 600 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extra-legs"
$B --frame-call process_frame > $O/bench_c2_process_frame.json 2>$O/e.err
$B --frame-call ahead > $O/bench_c2_ahead.json 2>$O/e.err
$B --origin-offset 20,-12,8 > $O/bench_c2_offset.json 2>$O/e.err
for k in 2 3 4; do $B --streams-per-gpu $k > $O/bench_k$k.json 2>$O/e.err; done
GPU_MAX_HW_QUEUES=8 $B --streams-per-gpu 4 > $O/bench_k4_q8.json 2>$O/e.err
$B --force-exchange > $O/bench_ex8.json 2>$O/e.err
$B --force-exchange --exchange-batch 1 > $O/bench_ex1.json 2>$O/e.err
for i in 1 2; do timeout 600 oracle/_ref/ref_hip_demo --bench 1000 2>/dev/null | grep '"bench"' > $O/binding_$i.json; done
timeout 600 ./tests/cpp/main_engine_demo --bench 300 > $O/closed_loop_cpp.json 2>&1
for m in --bench-host --bench-map --bench-map-host; do timeout 600 ./tests/cpp/main_engine_demo $m 1000 > $O/cpp$m.json 2>&1; done
timeout 600 python tools/closed_loop_bench.py 100 > $O/closed_loop_py.jsonl 2>&1
for f in $O/bench_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); r=d.get('roofline') or {}; print(d['value'], d['ms_per_step'], d['repetitions']['count'], r.get('avg_kernel_us'), r.get('frac'), (d.get('parity_check') or {}).get('equal'), {k:(v.get('avg_kernel_us'), v.get('frac')) for k,v in (r.get('other_kernels') or {}).items()})" 2>&1 | tail -1)"; done
cat $O/binding_*.json | cut -c1-60,330-520; cat $O/closed_loop_cpp.json | tail -2; tail -qn 1 $O/cpp--*.json | cut -c1-300; tail -2 $O/closed_loop_py.jsonl | cut -c1-250
cd /tmp && export TMPDIR=/tmp
for c in 2 3 5; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_c$c -o s -- python3 $R/bench.py --config $c --steps 200 --warmup 10 --no-cpu-baseline --no-extra-legs --timer-frames 1 > $R/$O/stats_c$c.log 2>&1
done
cd $R
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete
for c in 2 3 5; do echo "== config $c"; cut -c1-110 $O/stats_c$c/*kernel_stats.csv | head -8; done
