#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3sweeps_big; rm -rf $O; mkdir -p $O
timeout 1500 python tools/dense_classify_sweep.py 400 800 2>&1 | tail -2 | tee $O/dense_sweep.log
timeout 2400 python tests/stress_sweep.py 7000 1500 2>&1 | tail -2 | tee $O/stress_sweep.log
timeout 1200 python tools/tracker_sweep.py 2>&1 | tail -3 | tee $O/tracker_sweep.log
