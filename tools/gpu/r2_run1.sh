#!/bin/bash
# round 2, GPU call 1: parity suite, block-directory A/B, burst variants, launcher check, configs 3/5, rocprof stats
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for i in 1 2; do
  python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_dir_$i.json
  ITM_NO_DIRECTORY=1 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_table_$i.json
done
for v in b2 b8 run16; do ITM_LIB=gpurun_variants/lib_$v.so python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_$v.json; done
python tools/raycast_tune.py infinitam_amd/libitmhip.so > $O/tune_dir.txt 2>&1
ITM_NO_DIRECTORY=1 python tools/raycast_tune.py infinitam_amd/libitmhip.so > $O/tune_table.txt 2>&1
python tools/wave_stats.py gpurun_variants/lib_wt.so > $O/wave_dir.txt 2>&1; cp gpurun_out/wave_stats.npy $O/wave_stats_dir.npy
ITM_NO_DIRECTORY=1 python tools/wave_stats.py gpurun_variants/lib_wt.so > $O/wave_table.txt 2>&1; cp gpurun_out/wave_stats.npy $O/wave_stats_table.npy
python tools/config_bench.py 3 60 | tail -1 > $O/cfg3.json
python tools/config_bench.py 5 60 | tail -1 > $O/cfg5.json
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --force-exchange --no-cpu-baseline > $O/bench_fx1.json 2> $O/bench_fx1.err
python bench.py --force-exchange --exchange-batch 8 --no-cpu-baseline > $O/bench_fx8.json 2> $O/bench_fx8.err
ITM_BENCH_SHARED_GPU=1 python bench.py --gpus 2 --no-cpu-baseline > $O/bench_shared2.json 2> $O/bench_shared2.err
python bench.py --config 3 --steps 60 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --config 5 --steps 60 --warmup 5 > $O/bench_c5.json 2> $O/bench_c5.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o r2a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof -name "*.db" -delete; find $O/prof -name "*kernel_trace*" -delete; ls -R $O | head -50
cat $O/cfg2_*.json $O/tune_*.txt $O/wave_*.txt $O/cfg3.json $O/cfg5.json $O/bench_*.json
