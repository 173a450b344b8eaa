#!/bin/bash
# round 2, GPU call 17: full GPU suite + the three bench configurations after the integration rewrite
cd "$(dirname "$0")/../.."
O=gpurun_out/r2q; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --config 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --config 5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --streams-per-gpu 2 --no-cpu-baseline > $O/bench_c2_k2.json 2> $O/bench_c2_k2.err
for f in $O/bench_*.json; do echo "$f $(cut -c1-420 $f)"; done
