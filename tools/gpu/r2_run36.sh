#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_tracker.py -m gpu -x -q 2>&1 | tail -5
ITM_LIB=gpurun_variants/lib_trktrace.so timeout 120 python tools/closed_loop_bench.py 8 > gpurun_out/run36.txt 2>&1
grep -B12 '"bilateral": false' gpurun_out/run36.txt | head -14 | cut -c1-250
timeout 120 python tools/closed_loop_bench.py 100
timeout 120 python tools/tracker_bench.py | tail -4
