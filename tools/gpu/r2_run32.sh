#!/bin/bash
cd "$(dirname "$0")/../.."
tools/gpu/r2_run20.sh
for v in th0 th32 th256; do echo "== $v"; ITM_LIB=gpurun_variants/lib_$v.so timeout 120 python tools/tracker_bench.py 2>&1 | tail -1; done
