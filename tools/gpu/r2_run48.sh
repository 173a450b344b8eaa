#!/bin/bash
for i in 1 2; do
for v in "" prio1 prio2 prio3; do
  if [ -n "$v" ]; then export ITM_LIB=gpurun_variants/lib_$v.so; else unset ITM_LIB; fi
  echo "variant '$v': $(python tools/config_bench.py 2 200 | tail -1 | cut -c40-200)"
done; done
