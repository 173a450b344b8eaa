#!/bin/bash
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_golden_pool40000.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
for keys in "" 14; do
  for c in 2 5 3; do
    n=200; [ $c != 2 ] && n=40
    ITM_DEBUG_KEYS=$keys python tools/config_bench.py $c $n | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('keys', '$keys' or '-', 'config', $c, 'raycast', d['kernels_us']['raycast'], 'fps', d['fps_with_timers'])"
  done
done; done
