#!/bin/bash
# round 5: the whole GPU suite + smoke + the default bench line (one gpurun call)
cd "$(dirname "$0")/../.."
O=gpurun_out/r5suite; rm -rf $O; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q --durations=8 ${PYTEST_ARGS} > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|FAILED|Error|skipped" $O/pytest_gpu.log | tail -12; tail -12 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
