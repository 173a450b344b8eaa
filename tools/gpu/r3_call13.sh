#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3l; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; grep -E "passed|failed|FAILED" $O/pytest_all.log | tail -8
