#!/bin/bash
# SQ counters of the dense integration variants (unsaturated frames 0..29)
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r3f; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_SALU"
P2="SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA"
for m in 11 10 1 0; do
  export ITM_DENSE_MODES=$m
  rocprofv3 --kernel-trace --pmc $P1 --kernel-include-regex "integrate_dense" --output-format csv -d $R/$O/pmc_m${m}_a -o p -- python3 $R/tools/dense_modes.py 30 > $R/$O/pmc_m${m}_a.log 2>&1
  rocprofv3 --kernel-trace --pmc $P2 --kernel-include-regex "integrate_dense" --output-format csv -d $R/$O/pmc_m${m}_b -o p -- python3 $R/tools/dense_modes.py 30 > $R/$O/pmc_m${m}_b.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
O="gpurun_out/r3f"
for d in sorted(glob.glob(O+"/pmc_*_?")):
    files = glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not files: print(d, "no counter file"); continue
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for row in csv.DictReader(open(files[0])):
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    print(d, {k: round(acc[k]/max(1,n[k])) for k in acc}, "launches", max(n.values()) if n else 0)
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete; find $O -name "*counter_collection.csv" -size +1000k -delete
