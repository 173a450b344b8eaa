#!/bin/bash
# round 4: the GPU suite, the smoke check and the default bench line (one gpurun call)
cd "$(dirname "$0")/../.."
O=gpurun_out/r4suite; rm -rf $O; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x ${PYTEST_ARGS} > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|FAILED|Error" $O/pytest_gpu.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 1500 $O/bench_default.json
