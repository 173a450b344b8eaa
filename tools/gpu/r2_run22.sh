#!/bin/bash
# round 2, GPU call 22: far-field look-ahead of the dense ray cast; parity first
cd "$(dirname "$0")/../.."
O=gpurun_out/r2u; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_dense_512_properties.py tests/test_golden.py tests/test_golden_widening.py tests/test_dense_cull.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 300 python tools/config_bench.py 3 100 | tail -1 > $O/cfg3_dl8.json
for k in 0 4 12 16; do ITM_LIB=gpurun_variants/lib_dl$k.so timeout 300 python tools/config_bench.py 3 100 | tail -1 > $O/cfg3_dl$k.json; done
for f in $O/cfg3*.json; do echo "$f $(cut -c1-200 $f)"; done
