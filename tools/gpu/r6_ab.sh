#!/bin/bash
# round 6 A/B: parity of the touched paths first, then the in-tree library against measurement builds (VARIANTS) on CONFIGS
cd "$(dirname "$0")/../.."
# VARIANTS: measurement builds (gpurun_variants/lib_<name>.so); KEYS: debug-key settings of the in-tree library (bench.py --debug-keys)
O=gpurun_out/r6ab; rm -rf $O; mkdir -p $O
if [ -n "$TESTS" ]; then timeout 1800 python -m pytest $TESTS -m gpu -q -x > $O/pytest.log 2>&1; grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -8; fi
# parity of measurement builds (TEST_LIBS) on the tests of VTESTS
for V in $TEST_LIBS; do echo "== parity of lib_$V"; ITM_TEST_LIB=$PWD/gpurun_variants/lib_$V.so timeout 1800 python -m pytest $VTESTS -m gpu -q -x > $O/pytest_$V.log 2>&1; grep -E "passed|failed|FAILED|Error" $O/pytest_$V.log | tail -5; done
B="python bench.py --no-cpu-baseline --no-extra-legs --steps ${STEPS:-400} --warmup 40"
for rep in 1 2; do
  for c in ${CONFIGS:-2 3 5}; do
    $B --config $c > $O/c${c}_base_$rep.json 2>$O/e.err
    for V in $VARIANTS; do ITM_LIB_OVERRIDE=$PWD/gpurun_variants/lib_$V.so $B --config $c > $O/c${c}_${V}_$rep.json 2>$O/e.err; done
    for K in $KEYS; do $B --config $c --debug-keys $K > $O/c${c}_key${K//[=,]/_}_$rep.json 2>$O/e.err; done
    # RUNS: "<variant>@<keys>" (variant may be empty = the in-tree library)
    for R in $RUNS; do V=${R%%@*}; K=${R#*@}; L=""; [ -n "$V" ] && L=$PWD/gpurun_variants/lib_$V.so; ITM_LIB_OVERRIDE=$L $B --config $c --debug-keys $K > $O/c${c}_${V:-base}_key${K//[=,]/_}_$rep.json 2>$O/e.err; done
  done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6ab/c*.json')):
    try:
        d = json.load(open(f)); r = d.get('roofline') or {}; o = r.get('other_kernels') or {}
        print("%-28s %9.1f fps  %6.2f us/frame | %s %6.2f us | %s" % (f.split('/')[-1], d['value'], 1e3 * d['ms_per_step'], (r.get('kernel') or '')[:14], r.get('avg_kernel_us') or 0,
              "  ".join("%s %.2f" % (k, v['avg_kernel_us']) for k, v in o.items())))
    except Exception as e:
        print(f, 'ERR', e)
PY
if [ -n "$STAMPS" ]; then for G in ${STAMP_GRIDS:-0}; do echo "== stamps, grid $G"; ITM_STAMP_GRID=$G python tools/fused_stamps.py gpurun_variants/lib_$STAMPS.so; done; fi
