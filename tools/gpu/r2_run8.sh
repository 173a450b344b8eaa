#!/bin/bash
# round 2, GPU call 8: range reduction inside the ray cast, one-launch visible list; parity; A/B by debug keys; rocprof
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r2h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_default.json
ITM_DEBUG_KEYS=6 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_key6_separate_reduce.json
ITM_DEBUG_KEYS=7 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_key7_two_pass_list.json
ITM_DEBUG_KEYS=4 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_key4_no_fused_projection.json
ITM_DEBUG_KEYS=4,6,7 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_key467.json
python tools/config_bench.py 5 60 | tail -1 > $O/cfg5.json
python tools/config_bench.py 3 60 | tail -1 > $O/cfg3.json
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r2h -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/$O/prof.log 2>&1
export ITM_DEBUG_KEYS=4
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_key4 -o r2h4 -- python3 $R/tools/config_bench.py 2 200 > $R/$O/prof4.log 2>&1
unset ITM_DEBUG_KEYS
cd $R; find $O -name "*kernel_trace*" -delete; find $O -name "*.db" -delete
for f in $O/cfg*.json; do echo "$f $(cut -c1-330 $f)"; done; cut -c1-200 $O/bench_n1.json; echo; cut -c1-150 $O/prof/*kernel_stats.csv | head -10; cut -c1-150 $O/prof_key4/*kernel_stats.csv | head -12
