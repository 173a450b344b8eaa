#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3probe; rm -rf $O; mkdir -p $O
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extra-legs"
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_golden.py tests/test_random_stress.py tests/test_frame_ahead.py -m gpu -x -q 2>&1 | tail -1
for rep in 1 2; do
$B > $O/bench_new_$rep.json 2>$O/e.err
for V in ${VARIANTS:-noprobe probe1 probe3 probe4}; do
ITM_LIB_OVERRIDE=$PWD/gpurun_variants/lib_$V.so $B > $O/bench_${V}_$rep.json 2>$O/e.err
done
done
python bench.py --config 5 --no-cpu-baseline --no-extra-legs > $O/bench_c5_new.json 2>$O/e.err
ITM_LIB_OVERRIDE=$PWD/gpurun_variants/lib_nofar.so python bench.py --config 5 --no-cpu-baseline --no-extra-legs > $O/bench_c5_nofar.json 2>$O/e.err
for f in $O/bench_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('avg_kernel_us'))" 2>&1 | tail -1)"; done
