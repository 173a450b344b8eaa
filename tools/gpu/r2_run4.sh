#!/bin/bash
# round 2, GPU call 4: PMC counters of the ray-cast variants (table walk / directory plain / directory with joint runs)
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r2d; mkdir -p $O
python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_p2.json
ITM_LIB=gpurun_variants/lib_p1b1.so python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_p1b1.json
ITM_LIB=gpurun_variants/lib_dirplain.so python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_dirplain.json
ITM_NO_DIRECTORY=1 python tools/config_bench.py 2 200 | tail -1 > $O/cfg2_table.json
for f in $O/cfg2_*.json; do echo "$f $(cut -c1-330 $f)"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/$O/counters_list.txt 2>&1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
P2="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM"
run() { # name, pass-name, counters
  rocprofv3 --kernel-trace --pmc $3 --kernel-include-regex "raycast" --output-format csv -d $R/$O/pmc_$1_$2 -o p -- python3 $R/tools/config_bench.py 2 25 > $R/$O/pmc_$1_$2.log 2>&1
}
unset ITM_LIB ITM_NO_DIRECTORY
run p2 a "$P1"; run p2 b "$P2"; run p2 c "$P3"
export ITM_NO_DIRECTORY=1
run table a "$P1"; run table b "$P2"; run table c "$P3"
unset ITM_NO_DIRECTORY
export ITM_LIB=$R/gpurun_variants/lib_p1b1.so
run p1b1 a "$P1"; run p1b1 c "$P3"
export ITM_LIB=$R/gpurun_variants/lib_dirplain.so
run dirplain a "$P1"; run dirplain b "$P2"; run dirplain c "$P3"
unset ITM_LIB
cd $R
python3 - <<'PY'
import csv, glob, os, collections
O="gpurun_out/r2d"
for d in sorted(glob.glob(O+"/pmc_*_?")):
    files = glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not files: print(d, "no counter file"); continue
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for row in csv.DictReader(open(files[0])):
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    print(d, {k: round(acc[k]/max(1,n[k])) for k in acc}, "launches", max(n.values()) if n else 0)
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete
