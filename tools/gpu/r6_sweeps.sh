#!/bin/bash
# round 6: parity sweeps on the final build -- random stress cases with fresh seeds (all voxel x index types, fused / recorded / separate calls,
# free views), the dense mirror at its frustum-sized default, at 256^3 and the paged form
cd "$(dirname "$0")/../.."
O=gpurun_out/r6sweeps; rm -rf $O; mkdir -p $O
timeout 2400 python tests/stress_sweep.py ${SEED0:-100000} ${COUNT:-1500} 2>&1 | tail -2 | tee $O/stress_sweep.log
ITM_MIRROR_BITS=8 timeout 1200 python tests/stress_sweep.py ${SEED2:-120000} ${COUNT_256:-400} 2>&1 | tail -2 | tee $O/stress_sweep_mirror256.log
ITM_MIRROR=paged timeout 1200 python tests/stress_sweep.py ${SEED1:-140000} ${COUNT_PAGED:-400} 2>&1 | tail -2 | tee $O/stress_sweep_paged.log
