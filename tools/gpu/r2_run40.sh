#!/bin/bash
mkdir -p gpurun_out
nproc; python - <<'PY'
import os, glob
print("affinity", sorted(os.sched_getaffinity(0)))
for d in glob.glob("/sys/class/drm/card*/device"):
    try:
        print(d, open(d + "/vendor").read().strip(), "numa", open(d + "/numa_node").read().strip(), "cpus", open(d + "/local_cpulist").read().strip())
    except Exception as e:
        print(d, e)
PY
lscpu | grep -E "Model name|Socket|NUMA|Thread|Core" | head -12
for i in 1 2 3; do
for v in "" sleep0; do
  echo "variant: $v"
  if [ -n "$v" ]; then export ITM_LIB=gpurun_variants/lib_$v.so; else unset ITM_LIB; fi
  timeout 120 python tools/closed_loop_bench.py 100 | cut -c1-60,150-260 | head -1
done; done
