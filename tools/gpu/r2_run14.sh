#!/bin/bash
# round 2, GPU call 14: integration with two voxels per lane (packed fp32); parity first
cd "$(dirname "$0")/../.."
O=gpurun_out/r2n; mkdir -p $O
timeout 1200 python -m pytest tests/test_dense_cull.py tests/test_hip_parity.py tests/test_dense_512_properties.py tests/test_golden.py tests/test_golden_pool40000.py tests/test_golden_widening.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2.json
timeout 300 python tools/config_bench.py 3 100 | tail -1 > $O/cfg3.json
timeout 300 python tools/config_bench.py 5 60 | tail -1 > $O/cfg5.json
for f in $O/cfg*.json; do echo "$f $(cut -c1-400 $f)"; done
