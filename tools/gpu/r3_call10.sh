#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3j; rm -rf $O; mkdir -p $O
B="python bench.py --config 3 --no-cpu-baseline --no-extra-legs"
$B > $O/bench_c3.json 2>$O/bench_c3.err
$B --debug-keys 16=1 > $O/bench_c3_noclass.json 2>$O/bench_c3_noclass.err
$B --debug-keys 3=4096 > $O/bench_c3_wg4096.json 2>$O/bench_c3_wg4096.err
for f in $O/bench_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], d['roofline']['frac'])" 2>&1 | tail -1)"; done
timeout 900 python -m pytest tests/test_dense_512_properties.py tests/test_engine_api.py tests/test_random_stress.py -m gpu -x -q 2>&1 | tail -3
ITM_MODES_LIST=0 bash tools/gpu/r3_call9.sh 2>&1 | grep "pmc_m0_[12]"
