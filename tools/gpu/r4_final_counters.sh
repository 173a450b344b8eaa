#!/bin/bash
# SQ counters of the three roofline kernels as they are at the end of round 4 (separate --pmc passes, kernel trace only)
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r4fc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM"
declare -A RX=( [2]="raycast_kernel" [3]="integrate_dense" [5]="integrate_hash_kernel" )
for c in 2 3 5; do
  for pass in a c; do
    if [ $pass = a ]; then P="$P1"; else P="$P3"; fi
    rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "${RX[$c]}" --output-format csv -d $R/$O/pmc_c${c}_$pass -o p -- python3 $R/bench.py --config $c --steps 20 --warmup 5 --min-measured-s 0 --no-cpu-baseline --no-extra-legs --timer-frames 1 > $R/$O/pmc_c${c}_$pass.log 2>&1
  done
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json
O="gpurun_out/r4fc"; out={}
for d in sorted(glob.glob(O+"/pmc_c?_?")):
    files = glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not files: print(d, "no counter file"); continue
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for row in csv.DictReader(open(files[0])):
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    cfg = d.split("pmc_")[1].split("_")[0]
    out.setdefault(cfg, {}).update({k: round(acc[k]/max(1,n[k])) for k in acc}); out[cfg]["launches"] = max(n.values()) if n else 0
json.dump(out, open(O+"/counters.json","w"), indent=1); print(json.dumps(out, indent=1))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete; find $O -name "*counter_collection.csv" -delete
