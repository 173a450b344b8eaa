#!/bin/bash
for keys in "" 14; do
  for c in 2 5 3; do
    n=200; [ $c != 2 ] && n=40
    ITM_DEBUG_KEYS=$keys python tools/config_bench.py $c $n | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('keys', '$keys' or '-', 'config', $c, d['kernels_us'], 'fps', d['fps_with_timers'])"
  done
done
