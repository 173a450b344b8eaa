#!/bin/bash
for i in 1 2; do
for v in "" prio1 es1 es2 es4; do
  if [ -n "$v" ]; then export ITM_LIB=gpurun_variants/lib_$v.so; else unset ITM_LIB; fi
  for c in 2 5; do
    n=200; [ $c = 5 ] && n=40
    python tools/config_bench.py $c $n | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('variant', '$v' or 'base', 'config', $c, 'raycast', d['kernels_us']['raycast'], 'fps', d['fps_with_timers'])"
  done
done; done
