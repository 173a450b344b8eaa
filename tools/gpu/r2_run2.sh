#!/bin/bash
# round 2, GPU call 2: unconditional (parallel) loads in the ray caster; directory vs table walk; burst variants; wave attribution
cd "$(dirname "$0")/../.."
O=gpurun_out/r2b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for i in 1 2; do
  python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_dir_$i.json
  ITM_NO_DIRECTORY=1 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_table_$i.json
done
for v in b2 b3 b8 d8; do ITM_LIB=gpurun_variants/lib_$v.so python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_$v.json; done
python tools/raycast_tune.py infinitam_amd/libitmhip.so > $O/tune_dir.txt 2>&1
ITM_NO_DIRECTORY=1 python tools/raycast_tune.py infinitam_amd/libitmhip.so > $O/tune_table.txt 2>&1
python tools/wave_stats.py gpurun_variants/lib_wt.so > $O/wave_dir.txt 2>&1; cp gpurun_out/wave_stats.npy $O/wave_stats_dir.npy
ITM_NO_DIRECTORY=1 python tools/wave_stats.py gpurun_variants/lib_wt.so > $O/wave_table.txt 2>&1; cp gpurun_out/wave_stats.npy $O/wave_stats_table.npy
python tools/config_bench.py 3 60 | tail -1 > $O/cfg3.json
python tools/config_bench.py 5 60 | tail -1 > $O/cfg5.json
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r2b -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/$O/prof.log 2>&1
cd $R; find $O/prof -name "*kernel_trace*" -delete; find $O/prof -name "*.db" -delete; ls -R $O/prof | head
cat $O/cfg2_*.json | cut -c1-330; cat $O/tune_*.txt $O/wave_*.txt; cut -c1-300 $O/cfg3.json $O/cfg5.json; cut -c1-200 $O/bench_n1.json
