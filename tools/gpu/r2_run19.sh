#!/bin/bash
# round 2, GPU call 19: exchange issued from the library (RCCL world 1) vs through torch.distributed, per frame and per 8 frames
cd "$(dirname "$0")/../.."
O=gpurun_out/r2s; mkdir -p $O
timeout 600 python -m pytest tests/test_native_exchange.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench_noex.json 2> $O/bench_noex.err
python bench.py --no-cpu-baseline --force-exchange --exchange-batch 1 > $O/bench_lib_b1.json 2> $O/bench_lib_b1.err
python bench.py --no-cpu-baseline --force-exchange --exchange-batch 8 > $O/bench_lib_b8.json 2> $O/bench_lib_b8.err
python bench.py --no-cpu-baseline --force-exchange --exchange-batch 1 --exchange-impl torch > $O/bench_torch_b1.json 2> $O/bench_torch_b1.err
python bench.py --no-cpu-baseline --force-exchange --exchange-batch 8 --exchange-impl torch > $O/bench_torch_b8.json 2> $O/bench_torch_b8.err
for f in $O/bench_*.json; do echo "$f $(cut -c1-150 $f)"; done
for f in $O/*.err; do echo "== $f"; tail -n 3 $f; done 2>/dev/null | head -30
