#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_tracker.py -m gpu -x -q 2>&1 | tail -2
for v in "" sleep0 sleep32; do
  echo "variant: $v"
  if [ -n "$v" ]; then export ITM_LIB=gpurun_variants/lib_$v.so; fi
  timeout 120 python tools/closed_loop_bench.py 100 | cut -c1-60,150-260
  timeout 120 python tools/tracker_bench.py | tail -1
done
