#!/bin/bash
mkdir -p gpurun_out
ITM_LIB=gpurun_variants/lib_trktrace.so timeout 120 python tools/closed_loop_bench.py 8 > gpurun_out/run37.txt 2>&1
grep -B12 '"bilateral": false' gpurun_out/run37.txt | head -14 | cut -c1-250
