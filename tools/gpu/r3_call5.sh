#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3e; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_dense_cull.py tests/test_dense_512_properties.py -m gpu -x -q > $O/pytest_dense.log 2>&1; tail -15 $O/pytest_dense.log
export ITM_REPORT=5,50,110
ITM_DENSE_MODES=11,0 timeout 600 python tools/dense_modes.py 115 > $O/dense_modes.jsonl 2>$O/dense_modes.err
for wg in 1024 3072 4096; do ITM_DENSE_MODES=0 ITM_DEBUG_KV=3:$wg timeout 600 python tools/dense_modes.py 115 >> $O/dense_modes.jsonl 2>>$O/dense_modes.err; done
cat $O/dense_modes.jsonl; tail -3 $O/dense_modes.err
