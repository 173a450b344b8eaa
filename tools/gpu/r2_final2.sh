#!/bin/bash
# final numbers after the last kernel changes: bench lines of the three configs, rocprofv3 stats + traffic, tracker figures
O=gpurun_out/r2final2; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --config 3 > $O/bench_c3.json 2> /dev/null
python bench.py --config 5 > $O/bench_c5.json 2> /dev/null
python bench.py --steps 2000 --warmup 100 --no-cpu-baseline > $O/bench_c2_2000.json 2> /dev/null
for k in 2 3; do python bench.py --streams-per-gpu $k --no-cpu-baseline > $O/bench_k$k.json 2> /dev/null; done
GPU_MAX_HW_QUEUES=8 python bench.py --streams-per-gpu 4 --no-cpu-baseline > $O/bench_k4_q8.json 2> /dev/null
python tools/tracker_bench.py > $O/tracker.txt 2>&1
python tools/closed_loop_bench.py 100 > $O/closed_loop.txt 2>&1
python tools/closed_loop_multi.py 4 100 > $O/closed_loop_multi.txt 2>&1
bash tools/gpu/r2_profiles.sh > $O/profiles.log 2>&1
for f in $O/bench_*.json; do echo "$(basename $f): $(cut -c1-160 $f)"; done; tail -3 $O/tracker.txt; cat $O/closed_loop.txt $O/closed_loop_multi.txt | cut -c1-250
