#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do python bench.py --steps 2000 --warmup 100 --no-cpu-baseline 2>/dev/null | cut -c1-120; done
python bench.py 2>/dev/null | cut -c1-700
