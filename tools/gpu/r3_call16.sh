#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3o; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_swapping.py -m gpu -x -q 2>&1 | grep -E "passed|failed|^E" | cut -c1-400 | head -20
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; grep -E "passed|failed|FAILED" $O/pytest_all.log | tail -6
