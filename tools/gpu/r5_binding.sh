#!/bin/bash
# round 5: the compiled reference binding (integration/ITMEngines_HIP.h) -- parity under both mirror policies, its frame rate
cd "$(dirname "$0")/../.."
O=gpurun_out/r5binding; mkdir -p $O
timeout 600 oracle/_ref/ref_hip_demo > $O/demo.log 2>$O/demo.err; echo "demo rc $?"; cat $O/demo.log | cut -c1-400; tail -3 $O/demo.err
for i in 1 2 3; do timeout 600 oracle/_ref/ref_hip_demo --bench 300 > $O/bench_$i.log 2>$O/bench_$i.err; echo "bench rc $?"; grep bench $O/bench_$i.log | cut -c1-700; done
grep config $O/bench_1.log | cut -c1-400
timeout 900 python -m pytest tests/test_reference_integration.py tests/test_cpp_adapter.py tests/test_main_engine.py -m gpu -x -q 2>&1 | tail -5
