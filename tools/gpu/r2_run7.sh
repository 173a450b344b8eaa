#!/bin/bash
# round 2, GPU call 7: gated miss look-ahead; rocprof kernel stats of the default build
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r2g; mkdir -p $O
python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_default.json
for v in g6s3 g6s5 g4s3 g10s4 b3 b5; do ITM_LIB=gpurun_variants/lib_$v.so python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_$v.json; done
python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_default2.json
python tools/wave_stats.py gpurun_variants/lib_wt.so > $O/wave.txt 2>&1; cp gpurun_out/wave_stats.npy $O/wave_stats.npy; cp gpurun_out/wave_trace.npy $O/wave_trace.npy
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r2g -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/$O/prof.log 2>&1
cd $R; find $O/prof -name "*kernel_trace*" -delete; find $O/prof -name "*.db" -delete
for f in $O/cfg2_*.json; do echo "$f $(cut -c1-330 $f)"; done; cat $O/wave.txt; cut -c1-200 $O/bench_n1.json; cat $O/prof/*/*kernel_stats.csv 2>/dev/null | cut -c1-200 | head -12 || find $O/prof | head
