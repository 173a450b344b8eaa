#!/bin/bash
# round 2, GPU call 26: allocation sweep inside the visible-list launch; parity first
cd "$(dirname "$0")/../.."
O=gpurun_out/r2y; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2.json
ITM_DEBUG_KEYS=13 timeout 300 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_separate_sweep.json
timeout 300 python tools/config_bench.py 5 60 | tail -1 > $O/cfg5.json
for f in $O/cfg*.json; do echo "$f $(cut -c1-330 $f)"; done
python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-200 $O/bench_c2.json
python bench.py --no-cpu-baseline > $O/bench_c2b.json 2> $O/bench_c2.err; cut -c1-200 $O/bench_c2b.json
