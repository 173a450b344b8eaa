#!/bin/bash
# round 2, GPU call 18: k streams per GPU fed by k host threads
cd "$(dirname "$0")/../.."
O=gpurun_out/r2r; mkdir -p $O
for k in 2 3 4 6 8; do python bench.py --streams-per-gpu $k --no-cpu-baseline > $O/bench_k$k.json 2> $O/bench_k$k.err; done
python bench.py --streams-per-gpu 4 --no-host-threads --no-cpu-baseline > $O/bench_k4_1thread.json 2> $O/bench_k4_1thread.err
python bench.py --config 5 --streams-per-gpu 2 --no-cpu-baseline > $O/bench_c5_k2.json 2> $O/bench_c5_k2.err
python bench.py --config 3 --streams-per-gpu 2 --no-cpu-baseline > $O/bench_c3_k2.json 2> $O/bench_c3_k2.err
for f in $O/bench_*.json; do echo "$f $(cut -c1-200 $f)"; done; tail -3 $O/*.err | head -40
