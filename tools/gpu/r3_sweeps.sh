#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3sweeps; rm -rf $O; mkdir -p $O
timeout 900 python tools/dense_classify_sweep.py 0 400 2>&1 | tail -4 | tee $O/dense_sweep.log
timeout 1500 python tests/stress_sweep.py 6000 300 2>&1 | tail -4 | tee $O/stress_sweep.log
