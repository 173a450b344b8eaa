#!/bin/bash
# round 6: what bounds the hash integration launches -- SQ issue / wait, vector-memory path (TA / TCP), instruction cache -- per launch
# (PMC counters in their own runs, kernel trace only; JOBS="<config> <kernel regex> <tag>;..." overrides the default two kernels)
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r6counters; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
S2="SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_ACTIVE_INST_SCA"
S3="TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum"
S4="TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum"
S5="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES GRBM_GUI_ACTIVE TD_TD_BUSY_sum SQ_INSTS_SALU"
IFS=';' read -ra JL <<< "${JOBS:-2 integrate_project_kernel c2int;5 integrate_hash_kernel c5int}"
for job in "${JL[@]}"; do
  set -- $job; c=$1; rx=$2; tag=$3
  for pass in 1 2 3 4 5; do
    eval P=\$S$pass
    timeout 120 rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "$rx" --output-format csv -d $R/$O/pmc_${tag}_$pass -o p -- python3 $R/bench.py --config $c --steps 20 --warmup 5 --min-measured-s 0 --no-cpu-baseline --no-extra-legs --timer-frames 1 > $R/$O/pmc_${tag}_$pass.log 2>&1 || echo "pass $pass of $tag failed: $(tail -2 $R/$O/pmc_${tag}_$pass.log | cut -c1-200)"
  done
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json
O = "gpurun_out/r6counters"; out = {}
for d in sorted(glob.glob(O + "/pmc_*_*")):
    if d.endswith(".log"): continue
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files: print(d, "no counter file"); continue
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for row in csv.DictReader(open(files[0])):
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    tag = d.split("pmc_")[1].rsplit("_", 1)[0]
    e = out.setdefault(tag, {}); e.update({k: round(acc[k] / max(1, n[k]), 1) for k in acc}); e["launches"] = max(n.values()) if n else 0
json.dump(out, open(O + "/counters.json", "w"), indent=1); print(json.dumps(out, indent=1))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete; find $O -name "*counter_collection.csv" -delete
