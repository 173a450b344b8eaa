#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3g; rm -rf $O; mkdir -p $O
export ITM_REPORT=5,50,110 ITM_DENSE_MODES=1
for v in d1s d2s d4s d2d d2s32; do
  for wg in 1024 2048 4096; do
    ITM_LIB=gpurun_variants/lib_$v.so ITM_DEBUG_KV=3:$wg timeout 300 python tools/dense_modes.py 115 >> $O/variants.jsonl 2>>$O/variants.err
  done
done
ITM_DENSE_MODES=1,11 timeout 300 python tools/dense_modes.py 115 >> $O/variants.jsonl 2>>$O/variants.err
cat $O/variants.jsonl; tail -3 $O/variants.err
