#!/bin/bash
# host-bound paths: k scenes on k streams with the ICP maps in the ray-cast launch and (debug key 26) as their own launch; the reference binding's rate
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 400 --warmup 40"
V='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"])'
for rep in 1 2; do for k in 1 2 3; do
  echo "k=$k maps in the ray cast: $($B --streams-per-gpu $k 2>/dev/null | python3 -c "$V")   own launch: $($B --streams-per-gpu $k --debug-keys 26 2>/dev/null | python3 -c "$V")"
done; done
echo "k=4 q8 maps in the ray cast: $(GPU_MAX_HW_QUEUES=8 $B --streams-per-gpu 4 2>/dev/null | python3 -c "$V")   own launch: $(GPU_MAX_HW_QUEUES=8 $B --streams-per-gpu 4 --debug-keys 26 2>/dev/null | python3 -c "$V")"
for i in 1 2; do echo "binding: $(timeout 300 oracle/_ref/ref_hip_demo --bench 1000 2>/dev/null | grep '"bench"' | cut -c330-520)"; done
