#!/bin/bash
# MFMA utilisation of the net's dominant kernel (the hand-written tower convolution k_conv4w (or k_conv8w when selected)) inside the headline bench:
# one --pmc pass with --kernel-trace only.  Writes gpurun_out/<tag>_pmc_net.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_net
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "k_conv[48]w" --output-format csv -d /tmp/pmc_net -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-baseline 0 --saturated 0 --steady-state 0 --avg8-leg 0 > /tmp/pmc_net.log 2>&1
echo "exit=$?"
python3 - <<PY
import csv, glob, json, collections
per = collections.defaultdict(list)
dur = []
for f in glob.glob("/tmp/pmc_net/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        per[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("/tmp/pmc_net/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex k_conv[48]w -- python3 bench.py --steps 1 --warmup 0 --cpu-baseline 0 --saturated 0",
       "kernel": "sgo_conv4w::k_conv4w<true/false, 7> of csrc/sgo_conv4w.hpp (the default tower kernel; k_conv8w when selected) (launches with the bigger batch dominate the upper half)"}
for k, v in per.items():
    big = sorted(v)[len(v) // 2:]          # the 8192-batch launches dominate; report their mean
    out[k] = {"launches": len(v), "mean_upper_half": sum(big) / len(big)}
if dur:
    big = sorted(dur)[len(dur) // 2:]
    out["duration_ns_mean_upper_half"] = sum(big) / len(big)
# derived: MFMA busy cycles per SIMD-cycle available.  GRBM_GUI_ACTIVE is summed over the 8 XCDs (guide): /8 = chip cycles.
try:
    chip_cycles = out["GRBM_GUI_ACTIVE"]["mean_upper_half"] / 8.0
    out["chip_cycles"] = chip_cycles
    out["mfma_busy_per_simd_cycle"] = out["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_upper_half"] / (chip_cycles * 256 * 4)
except Exception as e:
    out["derived_error"] = repr(e)
json.dump(out, open("$R/gpurun_out/${TAG}_pmc_net.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
