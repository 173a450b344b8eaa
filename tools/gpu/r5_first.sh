#!/bin/bash
# round 5, first GPU call: the long-run parity tests, the recording contract, the bench line with its parity check
cd "$(dirname "$0")/../.."
O=gpurun_out/r5first; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_long_run_parity.py tests/test_deferred_fusion.py tests/test_fatal_status.py tests/test_frame_ahead.py -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_short.json 2>$O/bench_short.err; echo "bench rc $?"; tail -3 $O/bench_short.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r5first/bench_short.json'))
print('value', d['value'], 'parity', d['parity_check'])
print('roofline', {k: v for k, v in d['roofline'].items() if k != 'other_kernels'})
print('other', d['roofline']['other_kernels'])
print('cpu', d['cpu_baseline'])
PY
