#!/bin/bash
for F in 131 146; do for V in ls_new; do echo "== $V frame $F"; timeout 300 python tools/list_timeline.py gpurun_variants/lib_$V.so c2 $F detail 2>&1 | grep -v "^early" | tail -24; done; done
