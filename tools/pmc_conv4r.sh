#!/bin/bash
# Vector-memory path of the two tower kernels side by side: k_conv4w (weights through LDS) and k_conv4r (weights L2 -> registers).
# Separate --pmc passes with --kernel-trace only.  Writes gpurun_out/<tag>_pmc_conv4r.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
cd /tmp; export TMPDIR=/tmp
PASSES=("TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
        "TA_TA_BUSY_sum TD_TD_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
        "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum"
        "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
        "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS")
for K in 4w 4r; do
  i=0
  for P in "${PASSES[@]}"; do
    rm -rf /tmp/pmc_$K_$i
    timeout -k 10 120 rocprofv3 --pmc $P --kernel-trace --kernel-include-regex "k_conv4[wr]" --output-format csv -d /tmp/pmc_${K}_$i -- python3 $R/tools/conv_one.py $K > /tmp/pmc_${K}_$i.log 2>&1
    echo "$K pass $i exit=$?"; tail -1 /tmp/pmc_${K}_$i.log
    i=$((i+1))
  done
done
python3 - <<PY
import csv, glob, json, collections
out = {"what": "per-launch means over the 8 192 x 17 x 17 launches of tools/conv_one.py, one rocprofv3 --pmc pass per counter group", "kernels": {}}
for K in ("4w", "4r"):
    per = collections.defaultdict(list)
    for f in glob.glob("/tmp/pmc_%s_*/*/*counter_collection.csv" % K):
        for r in csv.DictReader(open(f)):
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out["kernels"]["k_conv" + K] = {k: sum(v) / len(v) for k, v in sorted(per.items())}
json.dump(out, open("$R/gpurun_out/${TAG}_pmc_conv4r.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
