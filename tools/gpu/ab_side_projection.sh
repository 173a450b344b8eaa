#!/bin/bash
# config 5 with the projection beside the integration (default) and after it (debug key 14), whole frames without per-kernel timers
for i in 1 2 3; do
for keys in "" 14; do
  python bench.py --config 5 --no-cpu-baseline --debug-keys "$keys" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('keys', '$keys' or '-', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'])"
done; done
