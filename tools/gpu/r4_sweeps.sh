#!/bin/bash
# round 4: parity sweeps on the final build -- random stress cases (fresh seeds; both mirror forms), dense classification seeds, tracker sweep
cd "$(dirname "$0")/../.."
O=gpurun_out/r4sweeps; rm -rf $O; mkdir -p $O
timeout 2400 python tests/stress_sweep.py ${SEED0:-20000} ${COUNT:-1200} 2>&1 | tail -2 | tee $O/stress_sweep.log
ITM_MIRROR=paged timeout 1800 python tests/stress_sweep.py ${SEED1:-40000} ${COUNT_PAGED:-600} 2>&1 | tail -2 | tee $O/stress_sweep_paged.log
timeout 1500 python tools/dense_classify_sweep.py 2000 600 2>&1 | tail -2 | tee $O/dense_sweep.log
timeout 1200 python tools/tracker_sweep.py 2>&1 | tail -3 | tee $O/tracker_sweep.log
