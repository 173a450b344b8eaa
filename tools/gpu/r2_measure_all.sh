#!/bin/bash
# Everything DESIGN.md / BASELINE.md / profiles/ quote for this round, in one GPU call.
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r2final; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --config 3 > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --config 5 > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --streams-per-gpu 2 --no-cpu-baseline > $O/bench_k2.json 2> /dev/null
python bench.py --streams-per-gpu 3 --no-cpu-baseline > $O/bench_k3.json 2> /dev/null
python bench.py --streams-per-gpu 4 --no-cpu-baseline > $O/bench_k4.json 2> /dev/null
GPU_MAX_HW_QUEUES=8 python bench.py --streams-per-gpu 4 --no-cpu-baseline > $O/bench_k4_q8.json 2> /dev/null
python bench.py --timer-every 1 --no-cpu-baseline > $O/bench_c2_timer_every_launch.json 2> /dev/null
python bench.py --force-exchange --exchange-batch 1 --no-cpu-baseline 2> /dev/null | tail -1 > $O/bench_fx1.json
python bench.py --force-exchange --exchange-batch 8 --no-cpu-baseline 2> /dev/null | tail -1 > $O/bench_fx8.json
ITM_BENCH_SHARED_GPU=1 python bench.py --gpus 2 --no-cpu-baseline --steps 50 --warmup 5 2> /dev/null | tail -1 > $O/bench_shared2.json
python tools/tracker_bench.py > $O/tracker.txt 2>&1
python tools/closed_loop_bench.py 100 > $O/closed_loop.txt 2>&1
python tools/closed_loop_bench.py 100 >> $O/closed_loop.txt 2>&1
ITM_LIB=gpurun_variants/lib_trktrace.so python tools/closed_loop_bench.py 8 2>&1 | grep -B12 '"bilateral": false' | head -12 > $O/tracker_trace.txt
python tools/host_cost.py > $O/host_cost.txt 2>&1
python tools/config_bench.py 2 200 | tail -1 > $O/cfg2.json
python tools/config_bench.py 3 60 | tail -1 > $O/cfg3.json
python tools/config_bench.py 5 60 | tail -1 > $O/cfg5.json
ITM_DEBUG_KEYS=8 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_single_phase_raycast.json
ITM_DEBUG_KEYS=12 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_no_sdf_mirror.json
ITM_NO_DIRECTORY=1 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_table_walk.json
python tools/raycast_timeline.py gpurun_variants/lib_rs.so > $O/raycast_timeline.txt 2>&1
python tools/fused_stamps.py gpurun_variants/lib_stamps.so > $O/fused_stamps.txt 2>&1
python tools/list_timeline.py gpurun_variants/lib_ls.so > $O/list_timeline.txt 2>&1
ITM_DEBUG_KEYS=13 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_separate_sweep.json
bash tools/gpu/r2_profiles.sh > $O/profiles.log 2>&1
for f in $O/bench_*.json; do echo "$f: $(cut -c1-230 $f)"; done; cat $O/tracker.txt $O/closed_loop.txt $O/raycast_timeline.txt $O/fused_stamps.txt $O/list_timeline.txt; cut -c1-330 $O/cfg*.json; tail -40 $O/profiles.log
