#!/bin/bash
# config 3 after the last dense changes: bench line + rocprofv3 kernel stats
R=$PWD; O=gpurun_out/r2c3; rm -rf $O; mkdir -p $O
python bench.py --config 3 > $O/bench_c3.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_c3 -o s -- python3 $R/bench.py --config 3 --steps 200 --warmup 10 --no-cpu-baseline > $R/$O/stats_c3.log 2>&1
cd $R; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete
cut -c1-200 $O/bench_c3.json; cut -c1-110 $O/stats_c3/*kernel_stats.csv | head -6
