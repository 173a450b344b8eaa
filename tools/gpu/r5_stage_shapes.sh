#!/bin/bash
timeout 900 python -m pytest tests/test_depth_stager.py tests/test_main_engine.py tests/test_cpp_adapter.py tests/test_reference_integration.py tests/test_deferred_fusion.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED" | tail -5
echo "== without the flush in release:"; ITM_TEST_LIB=$PWD/gpurun_variants/lib_noflush.so timeout 300 python -m pytest tests/test_depth_stager.py -q -m gpu -k release_launches 2>&1 | grep -E "passed|failed|AssertionError" | cut -c1-200 | tail -3
for i in 1 2; do timeout 600 oracle/_ref/ref_hip_demo --bench 1000 2>/dev/null | grep '"bench"' | cut -c330-520; done
