#!/bin/bash
for k in 1 2 3 4 6 8; do
  for ht in "" "--no-host-threads"; do
    python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --streams-per-gpu $k $ht 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('k', $k, '$ht', d['value'], d['ms_per_step'])"
  done
done
