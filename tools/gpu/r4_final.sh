#!/bin/bash
# round 4 final measurements: the whole GPU suite (dense mirror = default, then the hash tests once more with the PAGED mirror), smoke, the bench
# lines of configs 2 / 3 / 5, k streams, exchange, closed loops, then rocprofv3 kernel stats + HBM traffic counters (separate --pmc passes) of each
# config's roofline kernel (-> profiles/r4_*)
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r4final; rm -rf $O; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|FAILED" $O/pytest_gpu.log | tail -5
ITM_MIRROR=paged timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_random_stress.py tests/test_accel_origin.py tests/test_swapping.py tests/test_engine_api.py tests/test_golden.py tests/test_frame_ahead.py tests/test_deferred_fusion.py tests/test_checkpoint.py tests/test_meshing.py -m gpu -q > $O/pytest_gpu_paged.log 2>&1; grep -E "passed|failed|FAILED" $O/pytest_gpu_paged.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench_default.json 2>$O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_like.json 2>$O/e.err
python bench.py --config 3 > $O/bench_c3.json 2>$O/bench_c3.err
python bench.py --config 5 > $O/bench_c5.json 2>$O/bench_c5.err
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extra-legs"
$B --frame-call process_frame > $O/bench_c2_process_frame.json 2>$O/e.err
$B --frame-call ahead > $O/bench_c2_ahead.json 2>$O/e.err
$B --origin-offset 20,-12,8 > $O/bench_c2_offset.json 2>$O/e.err
$B --raw-depth > $O/bench_c2_raw.json 2>$O/e.err
ITM_MIRROR=paged $B > $O/bench_c2_paged.json 2>$O/e.err
for k in 2 3 4; do $B --streams-per-gpu $k > $O/bench_k$k.json 2>$O/e.err; done
GPU_MAX_HW_QUEUES=8 $B --streams-per-gpu 6 > $O/bench_k6_q8.json 2>$O/e.err
ITM_MIRROR=paged GPU_MAX_HW_QUEUES=8 $B --streams-per-gpu 8 > $O/bench_k8_q8_paged.json 2>$O/e.err
$B --force-exchange > $O/bench_ex8.json 2>$O/e.err
$B --force-exchange --exchange-batch 1 > $O/bench_ex1.json 2>$O/e.err
./tests/cpp/main_engine_demo --bench 300 > $O/closed_loop_cpp.json 2>&1
python tools/closed_loop_bench.py 100 > $O/closed_loop_py.jsonl 2>&1
for f in $O/bench_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); r=d.get('roofline') or {}; print(d['value'], d['ms_per_step'], d['repetitions']['count'], r.get('avg_kernel_us'), r.get('frac'), {k:v.get('avg_kernel_us') for k,v in (r.get('other_kernels') or {}).items()})" 2>&1 | tail -1)"; done
cat $O/closed_loop_cpp.json | tail -2; tail -2 $O/closed_loop_py.jsonl | cut -c1-250
cd /tmp && export TMPDIR=/tmp
for c in 2 3 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_c$c -o s -- python3 $R/bench.py --config $c --steps 200 --warmup 10 --no-cpu-baseline --no-extra-legs --timer-frames 1 > $R/$O/stats_c$c.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_ex -o s -- python3 $R/bench.py --force-exchange --exchange-batch 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extra-legs --timer-frames 1 > $R/$O/stats_ex.log 2>&1
declare -A RX=( [2]="raycast_kernel" [3]="integrate_dense" [5]="integrate_hash_kernel" )
for c in 2 3 5; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "${RX[$c]}" --output-format csv -d $R/$O/pmc_c${c}_$ctr -o p -- python3 $R/bench.py --config $c --steps 20 --warmup 5 --min-measured-s 0 --no-cpu-baseline --no-extra-legs --timer-frames 1 > $R/$O/pmc_c${c}_$ctr.log 2>&1
  done
done
cd $R
python3 - <<'PY'
import csv, glob, json, collections
O = "gpurun_out/r4final"
out = {}
for c in (2, 3, 5):
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(f"{O}/pmc_c{c}_{ctr}/**/*counter_collection.csv", recursive=True)
        if not files: continue
        rows = [r for r in csv.DictReader(open(files[0])) if r["Counter_Name"] == ctr]
        per_kernel = collections.defaultdict(list)
        for r in rows: per_kernel[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
        name, v = max(per_kernel.items(), key=lambda kv: len(kv[1]))
        vals[ctr] = {"kernel": name, "launches": len(v), "avg_KB": sum(v) / len(v)}
    out[f"config{c}"] = vals
json.dump(out, open(f"{O}/traffic_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete; find $O -name "*counter_collection.csv" -size +2000k -delete
for c in 2 3 5 ex; do echo "== config $c"; cut -c1-110 $O/stats_$( [ $c = ex ] && echo ex || echo c$c )/*kernel_stats.csv | head -8; done
