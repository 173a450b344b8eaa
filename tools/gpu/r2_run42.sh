#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_tracker.py -m gpu -x -q 2>&1 | tail -2
ITM_LIB=gpurun_variants/lib_trktrace.so timeout 120 python tools/closed_loop_bench.py 8 > gpurun_out/run42.txt 2>&1
grep -B12 '"bilateral": false' gpurun_out/run42.txt | head -12 | cut -c20-330
for i in 1 2 3; do timeout 120 python tools/closed_loop_bench.py 100 | cut -c1-60,150-260; done
timeout 120 python tools/tracker_bench.py | tail -12
