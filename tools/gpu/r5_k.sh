#!/bin/bash
# k scenes on k streams: the in-tree library against gpurun_variants/lib_head.so, twice
B="python bench.py --no-cpu-baseline --no-extra-legs --steps 400 --warmup 40"
for rep in 1 2; do for k in 2 3; do
  echo "k=$k base: $($B --streams-per-gpu $k 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"])')"
  echo "k=$k head: $(ITM_LIB_OVERRIDE=$PWD/gpurun_variants/lib_head.so $B --streams-per-gpu $k 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"])')"
done; done
echo "k=4 q8 base: $(GPU_MAX_HW_QUEUES=8 $B --streams-per-gpu 4 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"])')"
echo "k=4 q8 head: $(GPU_MAX_HW_QUEUES=8 ITM_LIB_OVERRIDE=$PWD/gpurun_variants/lib_head.so $B --streams-per-gpu 4 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"])')"
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_config4_streams.py tests/test_swapping.py tests/test_fatal_status.py -q -m gpu -x 2>&1 | tail -3
