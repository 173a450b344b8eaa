#!/bin/bash
# config 5 after the projection moved beside the integration: bench line + rocprofv3 kernel stats
R=$PWD; O=gpurun_out/r2c5; rm -rf $O; mkdir -p $O
python bench.py --config 5 > $O/bench_c5.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_c5 -o s -- python3 $R/bench.py --config 5 --steps 200 --warmup 10 --no-cpu-baseline > $R/$O/stats_c5.log 2>&1
cd $R; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete
cut -c1-200 $O/bench_c5.json; cut -c1-110 $O/stats_c5/*kernel_stats.csv | head -9
