#!/bin/bash
# round 2, GPU call 16: hash integration, one wave per (block, slice group) with next-item prefetch; parity first
cd "$(dirname "$0")/../.."
O=gpurun_out/r2p; mkdir -p $O; rm -f $O/*.json
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_golden.py tests/test_golden_pool40000.py tests/test_golden_widening.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for g in 0 1536 2048 3072 4096 8192; do
  ITM_DEBUG_KV=3:$g timeout 300 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_pf4_g$g.json; ITM_DEBUG_KV=3:$g timeout 300 python tools/config_bench.py 5 60 | tail -1 > $O/cfg5_pf4_g$g.json
done
for v in nopf pfsl2 pfsl8; do
  for g in 0 2048 4096; do
  ITM_DEBUG_KV=3:$g ITM_LIB=gpurun_variants/lib_$v.so timeout 300 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_${v}_g$g.json
  ITM_DEBUG_KV=3:$g ITM_LIB=gpurun_variants/lib_$v.so timeout 300 python tools/config_bench.py 5 60 | tail -1 > $O/cfg5_${v}_g$g.json
  done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2p/cfg*.json")):
    try: d = json.load(open(f))
    except Exception as e: print(f, "unreadable"); continue
    print(f.split("/")[-1], "integrate %.2f" % d["kernels_us"]["integrate"], "fps %.0f" % d["fps_with_timers"])
PY
