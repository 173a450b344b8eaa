#!/bin/bash
# round 2, GPU call 10: projection priority / 64 partial images; tracker with bounded workgroups; parity
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r2j; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
ITM_TEST_LIB=gpurun_variants/lib_parts64.so timeout 900 python -m pytest tests/test_hip_parity.py tests/test_edge_cases.py -m gpu -x -q 2>&1 | tail -2
python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_default.json
for v in noprio parts64 parts64_noprio; do ITM_LIB=gpurun_variants/lib_$v.so python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_$v.json; done
python tools/config_bench.py 5 60 | tail -1 > $O/cfg5.json
ITM_LIB=gpurun_variants/lib_parts64.so python tools/config_bench.py 5 60 | tail -1 > $O/cfg5_parts64.json
ITM_LIB=gpurun_variants/lib_noprio.so python tools/config_bench.py 5 60 | tail -1 > $O/cfg5_noprio.json
python tools/tracker_bench.py > $O/tracker.txt 2>&1
python tools/closed_loop_bench.py 60 > $O/closed_loop.txt 2>&1
python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
for f in $O/cfg*.json; do echo "$f $(cut -c1-330 $f)"; done; cat $O/tracker.txt $O/closed_loop.txt; cut -c1-200 $O/bench_n1.json
