#!/bin/bash
for i in 1 2; do
for v in "" file0 file1 file3; do
  if [ -n "$v" ]; then export ITM_LIB=gpurun_variants/lib_$v.so; else unset ITM_LIB; fi
  ITM_DEBUG_KEYS=14 python tools/config_bench.py 2 200 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('variant', '$v' or 'full', d['kernels_us'], 'fps', d['fps_with_timers'])"
done; done
