#!/bin/bash
for i in 1 2 3; do
  for pin in 0 1; do
    if [ $pin = 0 ]; then export ITM_NO_PIN=1; else unset ITM_NO_PIN; fi
    echo "pin $pin"; timeout 120 python tools/closed_loop_bench.py 100 2>&1 | cut -c1-60,150-260 | head -2
  done
done
unset ITM_NO_PIN; timeout 120 python tools/tracker_bench.py 2>&1 | tail -2
