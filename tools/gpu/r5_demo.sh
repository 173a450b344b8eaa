#!/bin/bash
# the reference-side binding: correctness demo (verbose closed loop) and its bench
mkdir -p gpurun_out/r5demo
ITM_DEMO_VERBOSE=1 timeout 600 oracle/_ref/ref_hip_demo > gpurun_out/r5demo/demo.log 2> gpurun_out/r5demo/demo.err; echo "demo rc=$?"
cat gpurun_out/r5demo/demo.log | cut -c1-400; tail -20 gpurun_out/r5demo/demo.err

for i in 1 2; do timeout 600 oracle/_ref/ref_hip_demo --bench 1000 2>&1 | grep bench | cut -c330-900; done
timeout 1200 python -m pytest tests/test_reference_integration.py tests/test_main_engine.py tests/test_cpp_adapter.py -q -m gpu -x 2>&1 | tail -4
