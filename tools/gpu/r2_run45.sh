#!/bin/bash
for q in 2 4 8; do
for k in 3 4 6; do
    GPU_MAX_HW_QUEUES=$q python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --streams-per-gpu $k 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('queues', $q, 'k', $k, d['value'], d['ms_per_step'])"
done; done
