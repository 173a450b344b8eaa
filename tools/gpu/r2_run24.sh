#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2w; mkdir -p $O
timeout 300 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_main.json
for v in s8 s6 s10 b6 b8 b2 s8b6 la10; do ITM_LIB=gpurun_variants/lib_m$v.so timeout 300 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_$v.json; done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2w/cfg2*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], "raycast %.2f" % d["kernels_us"]["raycast"], "fps %.0f" % d["fps_with_timers"])
PY
