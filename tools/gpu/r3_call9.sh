#!/bin/bash
# SQ counters of the dense integration on SATURATED frames (dispatches 105..129) vs unsaturated (5..29)
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r3i; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_SALU"
P2="SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
export ITM_REPORT=5
for m in ${ITM_MODES_LIST:-11 1}; do
  export ITM_DENSE_MODES=$m
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "integrate_dense|depth_tiles" --output-format csv -d $R/$O/pmc_m${m}_$i -o p -- python3 $R/tools/dense_modes.py 130 > $R/$O/pmc_m${m}_$i.log 2>&1
  done
done
cd $R
python3 - <<'PY'
import csv, glob, collections
O="gpurun_out/r3i"
for d in sorted(glob.glob(O+"/pmc_*_?")):
    files = glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not files: print(d, "no counter file"); continue
    rows = list(csv.DictReader(open(files[0])))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows if "depth_tiles" not in r["Kernel_Name"]})
    order = {d_: i for i, d_ in enumerate(ids)}
    for name, lo, hi in (("unsat 5-29", 5, 29), ("sat 105-129", 105, 129)):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in rows:
            k = order.get(int(r["Dispatch_Id"]), -1)
            if "depth_tiles" in r["Kernel_Name"]: continue
            if lo <= k <= hi:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        print(d.split("/")[-1], name, {k: round(acc[k]/max(1,n[k])) for k in acc})
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete; find $O -name "*counter_collection.csv" -size +1000k -delete
