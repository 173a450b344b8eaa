#!/bin/bash
cd "$(dirname "$0")/../.."
for v in ghw8 ghw16 ghg512; do echo "== $v"; ITM_LIB=gpurun_variants/lib_$v.so timeout 120 python tools/tracker_bench.py 2>&1 | tail -1; done
