#!/bin/bash
# Cycles vs clock of k_conv4r and its timing ablations (variants build): one --pmc pass each.  Writes gpurun_out/<tag>_pmc_conv4r_ablate.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
cd /tmp; export TMPDIR=/tmp
for V in 1 5 65 69; do
  rm -rf /tmp/pmca_$V
  timeout -k 10 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --kernel-include-regex "k_conv4r" --output-format csv -d /tmp/pmca_$V -- python3 $R/tools/conv_one.py 4r $V > /tmp/pmca_$V.log 2>&1
  echo "var $V exit=$?"; grep "ms per launch" /tmp/pmca_$V.log
done
python3 - <<PY
import csv, glob, json, collections
out = {"what": "k_conv4r and its timing ablations under one rocprofv3 --pmc pass each (launches serialised by the counter pass): chip cycles = GRBM_GUI_ACTIVE / 8, clock = cycles / kernel duration", "variants": {}}
names = {1: "full kernel", 5: "no weight loads", 65: "no pixel fragment reads", 69: "neither"}
for V in (1, 5, 65, 69):
    per = collections.defaultdict(list)
    dur = []
    for f in glob.glob("/tmp/pmca_%d/*/*counter_collection.csv" % V):
        for r in csv.DictReader(open(f)):
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob("/tmp/pmca_%d/*/*kernel_trace.csv" % V):
        for r in csv.DictReader(open(f)):
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    d = {k: sum(v) / len(v) for k, v in per.items()}
    if dur and "GRBM_GUI_ACTIVE" in d:
        d["duration_ms"] = sum(dur) / len(dur) / 1e6
        d["chip_cycles"] = d["GRBM_GUI_ACTIVE"] / 8
        d["clock_ghz"] = d["chip_cycles"] / (d["duration_ms"] * 1e6)
        d["mfma_busy_per_simd_cycle"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (d["chip_cycles"] * 1024)
    out["variants"]["%d: %s" % (V, names[V])] = d
json.dump(out, open("$R/gpurun_out/${TAG}_pmc_conv4r_ablate.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
