#!/bin/bash
# round 4 A/B: parity of the new paths first, then the in-tree library against measurement builds (VARIANTS) and debug keys (KEYS)
cd "$(dirname "$0")/../.."
O=gpurun_out/r4ab; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest ${TESTS:-tests/test_near_bits.py tests/test_fatal_status.py tests/test_swapping.py tests/test_config4_streams.py::test_bench_with_two_ranks_sharing_the_gpu} -m gpu -q -x > $O/pytest.log 2>&1; grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -8
B="python bench.py --no-cpu-baseline --no-extra-legs --steps 400 --warmup 40"
for rep in 1 2; do
  for c in ${CONFIGS:-2 5}; do
    $B --config $c > $O/c${c}_base_$rep.json 2>$O/e.err
    for V in $VARIANTS; do ITM_LIB_OVERRIDE=$PWD/gpurun_variants/lib_$V.so $B --config $c > $O/c${c}_${V}_$rep.json 2>$O/e.err; done
    for K in $KEYS; do $B --config $c --debug-keys $K > $O/c${c}_key${K}_$rep.json 2>$O/e.err; done
  done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4ab/c*.json')):
    try:
        d = json.load(open(f)); r = d.get('roofline') or {}; o = r.get('other_kernels') or {}
        print("%-40s %9.1f fps  %6.2f us/frame | %s %6.2f us | %s" % (f.split('/')[-1], d['value'], 1e3 * d['ms_per_step'], (r.get('kernel') or '')[:14], r.get('avg_kernel_us') or 0,
              "  ".join("%s %.2f" % (k, v['avg_kernel_us']) for k, v in o.items())))
    except Exception as e:
        print(f, 'ERR', e)
PY
for T in $TIMELINES; do echo "== $T"; python tools/$T 2>&1 | tail -${TAIL:-14}; done
