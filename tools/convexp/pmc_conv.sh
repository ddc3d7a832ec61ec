#!/bin/bash
# PMC passes over the hand-written tower convolution (separate --pmc runs, --kernel-trace only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-conv}
cd /tmp; export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc_conv_$i
  timeout -k 10 200 rocprofv3 --pmc $SET --kernel-trace --kernel-include-regex "k_conv8" --output-format csv -d /tmp/pmc_conv_$i -- python3 $R/tools/convexp/loop8p.py 8192 > /tmp/pmc_conv_$i.log 2>&1
  echo "pass $i exit=$?"
  i=$((i+1))
done
python3 - <<PY
import csv, glob, json, collections
out = {}
for i in range(5):
    per = collections.defaultdict(list)
    for f in glob.glob("/tmp/pmc_conv_%d/*/*counter_collection.csv" % i):
        for r in csv.DictReader(open(f)):
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        out[k] = {"launches": len(v), "mean": sum(v) / len(v)}
    dur = []
    for f in glob.glob("/tmp/pmc_conv_%d/*/*kernel_trace.csv" % i):
        for r in csv.DictReader(open(f)):
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if dur:
        out["duration_ns_pass%d" % i] = sum(dur) / len(dur)
json.dump(out, open("$R/gpurun_out/${TAG}_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
