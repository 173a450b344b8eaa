#!/bin/bash
# round 2, GPU call 5: plain directory + predicted-band fused step; parity suite; threshold / burst variants
cd "$(dirname "$0")/../.."
O=gpurun_out/r2f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_default.json
ITM_NO_DIRECTORY=1 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_table.json
for v in la0 la3 la10 la16 b2 b3 b6; do ITM_LIB=gpurun_variants/lib_$v.so python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_$v.json; done
python tools/raycast_tune.py infinitam_amd/libitmhip.so > $O/tune.txt 2>&1
python tools/wave_stats.py gpurun_variants/lib_wt.so > $O/wave.txt 2>&1; cp gpurun_out/wave_stats.npy $O/wave_stats.npy
python tools/config_bench.py 3 60 | tail -1 > $O/cfg3.json
python tools/config_bench.py 5 60 | tail -1 > $O/cfg5.json
ITM_LIB=gpurun_variants/lib_la0.so python tools/config_bench.py 5 60 | tail -1 > $O/cfg5_la0.json
ITM_LIB=gpurun_variants/lib_la0.so python tools/config_bench.py 3 60 | tail -1 > $O/cfg3_la0.json
python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err

for f in $O/cfg2_*.json $O/cfg5*.json $O/cfg3*.json; do echo "$f $(cut -c1-330 $f)"; done; cat $O/tune.txt $O/wave.txt $O/closed_loop.txt; cut -c1-200 $O/bench_n1.json
