#!/bin/bash
# A/B of library variants (gpurun_variants/lib_<name>.so, tools/build_variant.sh) against the in-tree build: VARIANTS="a b" CONFIGS="2 5" bash tools/gpu/r3_var.sh
cd "$(dirname "$0")/../.."
O=gpurun_out/r3var; rm -rf $O; mkdir -p $O
for c in ${CONFIGS:-2 5}; do
B="python bench.py --config $c --no-cpu-baseline --no-extra-legs"
if [ $c = 2 ]; then B="$B --steps 400 --warmup 40"; fi
for rep in 1 2; do
$B > $O/bench_c${c}_base_$rep.json 2>$O/e.err
for V in $VARIANTS; do
ITM_LIB_OVERRIDE=$PWD/gpurun_variants/lib_$V.so $B > $O/bench_c${c}_${V}_$rep.json 2>$O/e.err
done
done
done
for f in $O/bench_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('avg_kernel_us'))" 2>&1 | tail -1)"; done
