#!/bin/bash
# round 5: parity sweeps on the final build -- random stress cases with fresh seeds (all voxel x index types, fused / recorded / separate calls, free views), both mirror forms
cd "$(dirname "$0")/../.."
O=gpurun_out/r5sweeps; rm -rf $O; mkdir -p $O
timeout 2700 python tests/stress_sweep.py ${SEED0:-60000} ${COUNT:-1500} 2>&1 | tail -2 | tee $O/stress_sweep.log
ITM_MIRROR=paged timeout 1500 python tests/stress_sweep.py ${SEED1:-80000} ${COUNT_PAGED:-500} 2>&1 | tail -2 | tee $O/stress_sweep_paged.log
