#!/bin/bash
# the reference-side binding: correctness demo (verbose closed loop) and its bench
mkdir -p gpurun_out/r5demo
ITM_DEMO_VERBOSE=1 timeout 600 oracle/_ref/ref_hip_demo > gpurun_out/r5demo/demo.log 2> gpurun_out/r5demo/demo.err; echo "demo rc=$?"
cat gpurun_out/r5demo/demo.log | cut -c1-400; tail -20 gpurun_out/r5demo/demo.err

