#!/bin/bash
# round 2, GPU call 15: SQ counters of the integration kernels, one voxel per lane (HEAD) vs two (packed)
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r2o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM"
P4="SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
run() { # name, pass, counters, config, frames
  rocprofv3 --kernel-trace --pmc $3 --kernel-include-regex "integrate" --output-format csv -d $R/$O/pmc_$1_c$4_$2 -o p -- python3 $R/tools/config_bench.py $4 $5 > $R/$O/pmc_$1_c$4_$2.log 2>&1
}
for lib in main head; do
  if [ $lib = head ]; then export ITM_LIB=$R/gpurun_variants/lib_head.so; else unset ITM_LIB; fi
  for c in 2 3 5; do
    run $lib a "$P1" $c 25; run $lib c "$P3" $c 25; run $lib d "$P4" $c 25
  done
done
unset ITM_LIB
cd $R
python3 - <<'PY'
import csv, glob, os, collections
O="gpurun_out/r2o"
for d in sorted(glob.glob(O+"/pmc_*_?")):
    files = glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not files: print(d, "no counter file"); continue
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for row in csv.DictReader(open(files[0])):
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    print(d, {k: round(acc[k]/max(1,n[k])) for k in acc}, "launches", max(n.values()) if n else 0)
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete
