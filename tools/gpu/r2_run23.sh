#!/bin/bash
# round 2, GPU call 23: sdf mirror (voxels addressed by position for the ray cast); parity first
cd "$(dirname "$0")/../.."
O=gpurun_out/r2v; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 300 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_mirror.json
ITM_DEBUG_KEYS=12 timeout 300 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_nomirror.json
for f in $O/cfg2*.json; do echo "$f $(cut -c1-330 $f)"; done
python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-200 $O/bench_c2.json
