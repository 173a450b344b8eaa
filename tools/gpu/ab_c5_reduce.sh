#!/bin/bash
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_golden_pool40000.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do
for keys in "" 6; do
  python bench.py --config 5 --no-cpu-baseline --debug-keys "$keys" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('keys', '$keys' or '-', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'])"
done; done
