#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3ab; rm -rf $O; mkdir -p $O
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extra-legs"
V=${VARIANT:-dirlin}
ITM_LIB_OVERRIDE=$PWD/gpurun_variants/lib_$V.so timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -1
for rep in 1 2 3; do
$B > $O/bench_base_$rep.json 2>$O/e.err
ITM_LIB_OVERRIDE=$PWD/gpurun_variants/lib_$V.so $B > $O/bench_var_$rep.json 2>$O/e.err
done
for c in 5; do
python bench.py --config $c --no-cpu-baseline --no-extra-legs > $O/bench_c${c}_base.json 2>$O/e.err
ITM_LIB_OVERRIDE=$PWD/gpurun_variants/lib_$V.so python bench.py --config $c --no-cpu-baseline --no-extra-legs > $O/bench_c${c}_var.json 2>$O/e.err
done
for f in $O/bench_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('avg_kernel_us'))" 2>&1 | tail -1)"; done
