#!/bin/bash
for i in 1 2 3; do timeout 120 python tools/closed_loop_bench.py 100 | cut -c1-60,150-260; done
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
