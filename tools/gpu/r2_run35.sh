#!/bin/bash
mkdir -p gpurun_out
ITM_LIB=gpurun_variants/lib_trktrace.so python tools/closed_loop_bench.py 8 > gpurun_out/run35.txt 2>&1
grep -B12 '"bilateral": false' gpurun_out/run35.txt | head -30
