#!/bin/bash
# round 2, GPU call 12: ray cast with a barrier-free second pass (rays enter the queue when they park; idle waves take them); parity first
cd "$(dirname "$0")/../.."
O=gpurun_out/r2l; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_golden_pool40000.py tests/test_golden.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
timeout 120 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_main.json
for v in pa0 pa1 pa3 pa4 pa2s16 pa2s8 pa2cm1; do ITM_LIB=gpurun_variants/lib_$v.so timeout 120 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_$v.json; done
timeout 120 python tools/config_bench.py 5 60 | tail -1 > $O/cfg5_main.json
for v in pa0 pa2s16 pa2s8; do ITM_LIB=gpurun_variants/lib_$v.so timeout 120 python tools/config_bench.py 5 60 | tail -1 > $O/cfg5_$v.json; done
timeout 120 python tools/raycast_timeline.py gpurun_variants/lib_rs.so > $O/timeline.txt 2>&1
for f in $O/cfg2_*.json $O/cfg5*.json; do echo "$f $(cut -c1-330 $f)"; done; cat $O/timeline.txt
