#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3h; rm -rf $O; mkdir -p $O
export ITM_REPORT=5,50,110 ITM_DENSE_MODES=11
for sp in 64 32 16 8 4 2; do
  ITM_SPLITS=$sp timeout 300 python tools/dense_modes.py 115 | sed "s/^/splits $sp /" >> $O/splits.jsonl 2>>$O/splits.err
done
cat $O/splits.jsonl; tail -3 $O/splits.err
