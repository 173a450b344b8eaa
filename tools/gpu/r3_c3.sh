#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r3c3; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_dense_cull.py tests/test_dense_512_properties.py -m gpu -q 2>&1 | tail -1
python bench.py --config 3 > $O/bench_c3.json 2>$O/e.err; python -c "import json; d=json.load(open('$O/bench_c3.json')); print(d['value'], d['roofline'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_c3 -o s -- python3 $R/bench.py --config 3 --steps 200 --warmup 10 --no-cpu-baseline --no-extra-legs > $R/$O/stats_c3.log 2>&1
cd $R; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete; cut -c1-100 $O/stats_c3/s_kernel_stats.csv | head -5
