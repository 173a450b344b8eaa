#!/bin/bash
# round 2, GPU call 13: dense integration with the per-column frustum interval; parity first
cd "$(dirname "$0")/../.."
O=gpurun_out/r2m; mkdir -p $O
timeout 900 python -m pytest tests/test_dense_cull.py tests/test_hip_parity.py tests/test_dense_512_properties.py tests/test_golden.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python tools/config_bench.py 3 100 | tail -1 > $O/cfg3_main.json
ITM_DEBUG_KEYS=9 timeout 300 python tools/config_bench.py 3 100 | tail -1 > $O/cfg3_groupcull.json
for f in $O/cfg3*.json; do echo "$f $(cut -c1-400 $f)"; done
