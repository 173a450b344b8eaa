#!/bin/bash
# round 6 final measurements: GPU suite (+ the hash tests under the paged mirror), smoke, the default bench line (other_configs, exploring, parity
# checks, CPU samples), the driver's 20-step form, secondary legs, rocprofv3 kernel stats of each configuration, HBM traffic (FETCH_SIZE /
# WRITE_SIZE, separate passes, every kernel of the frame)  (-> profiles/r6_*, traffic_r06.json; SQ / TA / TCP counters: r6_counters.sh)
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r6final; rm -rf $O; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
  timeout 2700 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|FAILED" $O/pytest_gpu.log | tail -5
  ITM_MIRROR=paged timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_random_stress.py tests/test_accel_origin.py tests/test_swapping.py tests/test_engine_api.py tests/test_golden.py tests/test_frame_ahead.py tests/test_deferred_fusion.py -m gpu -q > $O/pytest_gpu_paged.log 2>&1; grep -E "passed|failed|FAILED" $O/pytest_gpu_paged.log | tail -3
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
timeout 1200 python bench.py > $O/bench_default.json 2>$O/bench_default.err; echo "bench default rc $?"
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_driver_like.json 2>$O/bench_driver_like.err; echo "bench driver-like rc $?"
B="timeout 600 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extra-legs"
$B --config 3 > $O/bench_c3.json 2>$O/e.err
$B --config 5 > $O/bench_c5.json 2>$O/e.err
if [ -z "$SKIP_LEGS" ]; then
  $B --frame-call process_frame > $O/bench_c2_process_frame.json 2>$O/e.err
  $B --frame-call ahead > $O/bench_c2_ahead.json 2>$O/e.err
  $B --origin-offset 20,-12,8 > $O/bench_c2_offset.json 2>$O/e.err
  ITM_MIRROR_BITS=8 $B > $O/bench_c2_mirror256.json 2>$O/e.err
  ITM_MIRROR=paged $B > $O/bench_c2_paged.json 2>$O/e.err
  for k in 2 3 4; do $B --streams-per-gpu $k > $O/bench_k$k.json 2>$O/e.err; done
  GPU_MAX_HW_QUEUES=8 $B --streams-per-gpu 4 > $O/bench_k4_q8.json 2>$O/e.err
  $B --force-exchange > $O/bench_ex8.json 2>$O/e.err
  $B --force-exchange --exchange-batch 1 > $O/bench_ex1.json 2>$O/e.err
  for i in 1 2; do timeout 600 oracle/_ref/ref_hip_demo --bench 1000 2>/dev/null | grep '"bench"' > $O/binding_$i.json; done
fi
for f in $O/bench_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); r=d.get('roofline') or {}; print(d['value'], d['ms_per_step'], d['repetitions']['count'], r.get('avg_kernel_us'), r.get('frac'), (d.get('parity_check') or {}).get('equal'), {k:(v.get('avg_kernel_us'), v.get('frac')) for k,v in (r.get('other_kernels') or {}).items()})" 2>&1 | tail -1)"; done
cat $O/binding_*.json 2>/dev/null | cut -c1-60,330-520
cd /tmp && export TMPDIR=/tmp
for c in 2 3 5; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_c$c -o s -- python3 $R/bench.py --config $c --steps 200 --warmup 10 --no-cpu-baseline --no-extra-legs --timer-frames 1 > $R/$O/stats_c$c.log 2>&1
done
# HBM traffic: one pass per counter and configuration, every kernel of the frame (PMC in its own run, kernel trace only)
for c in 2 3 5; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/$O/pmc_c${c}_$ctr -o p -- python3 $R/bench.py --config $c --steps 20 --warmup 5 --min-measured-s 0 --no-cpu-baseline --no-extra-legs --timer-frames 1 > $R/$O/pmc_c${c}_$ctr.log 2>&1
  done
done
cd $R
python3 - <<'PY'
import csv, glob, json, collections
O = "gpurun_out/r6final"
def short(name):
    n = name.replace("void ", "").replace("itm::", "")
    return n.split("(")[0][:80]
traffic = {}
for c in (2, 3, 5):
    per = collections.defaultdict(dict)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(f"{O}/pmc_c{c}_{ctr}/**/*counter_collection.csv", recursive=True)
        if not files: continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] == ctr: acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if len(v) >= 20: per[k][ctr + "_KB"] = round(sum(v) / len(v), 1); per[k]["launches"] = len(v)
    traffic[f"config{c}"] = per
json.dump(traffic, open(f"{O}/traffic_raw.json", "w"), indent=1)
print(json.dumps(traffic, indent=1)[:6000])
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete; find $O -name "*counter_collection.csv" -delete
for c in 2 3 5; do echo "== config $c"; cut -c1-110 $O/stats_c$c/*kernel_stats.csv | head -8; done
