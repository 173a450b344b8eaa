#!/bin/bash
# effect of the event pair on the throughput: timer on every frame, every 8th, every 64th
mkdir -p gpurun_out
for e in 1 8 64 1 8 64; do
  python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --timer-every $e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('every', $e, d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline'].get('launch_us'))" 
done > gpurun_out/run34.txt 2>&1
for c in 3 5; do for e in 1 8; do
  python bench.py --config $c --steps 500 --warmup 50 --no-cpu-baseline --timer-every $e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config', $c, 'every', $e, d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done; done >> gpurun_out/run34.txt 2>&1
cat gpurun_out/run34.txt
