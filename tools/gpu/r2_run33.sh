#!/bin/bash
# round 2, GPU call 33: allocation sweep with its loads in three independent rounds; parity first
cd "$(dirname "$0")/../.."
O=gpurun_out/r2ae; mkdir -p $O; rm -f $O/*.json
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python tools/list_timeline.py gpurun_variants/lib_ls.so
timeout 300 python tools/config_bench.py 2 200 | tail -1 | cut -c1-220
timeout 300 python tools/config_bench.py 5 60 | tail -1 | cut -c1-260
python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-160
python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-160
