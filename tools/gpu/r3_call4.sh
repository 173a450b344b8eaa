#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3d; rm -rf $O; mkdir -p $O
timeout 600 python tools/dense_modes.py 130 > $O/dense_modes.jsonl 2>$O/dense_modes.err; cat $O/dense_modes.jsonl; tail -3 $O/dense_modes.err
