#!/bin/bash
# round 2, GPU call 20: tracker evaluation session (one resident kernel per TrackCamera call)
cd "$(dirname "$0")/../.."
O=gpurun_out/r2t; mkdir -p $O
timeout 300 python -m pytest tests/test_tracker.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 120 python tools/tracker_bench.py > $O/tracker.txt 2>&1; tail -3 $O/tracker.txt
timeout 200 python tools/closed_loop_bench.py 100 > $O/closed_loop.txt 2>&1; cat $O/closed_loop.txt | cut -c1-400
