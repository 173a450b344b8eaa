#!/bin/bash
# round 2, GPU call 11: two-pass ray cast (parked rays + look-ahead in homogeneous waves)
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r2l; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_default.json
ITM_DEBUG_KEYS=8 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_single_pass.json
for v in ps3 ps6 ps8 la4 la10; do ITM_LIB=gpurun_variants/lib_$v.so python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_$v.json; done
python tools/config_bench.py 5 60 | tail -1 > $O/cfg5.json
ITM_DEBUG_KEYS=8 python tools/config_bench.py 5 60 | tail -1 > $O/cfg5_single_pass.json
python tools/raycast_tune.py infinitam_amd/libitmhip.so > $O/tune.txt 2>&1
python tools/wave_stats.py gpurun_variants/lib_wt.so > $O/wave.txt 2>&1
python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o s -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/$O/prof.log 2>&1
cd $R; find $O -name "*kernel_trace*" -delete; find $O -name "*.db" -delete
for f in $O/cfg*.json; do echo "$f $(cut -c1-330 $f)"; done; cat $O/tune.txt $O/wave.txt; cut -c1-200 $O/bench_n1.json; echo; cut -c1-120 $O/prof/*kernel_stats.csv | head -9
