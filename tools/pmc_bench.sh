#!/bin/bash
# HBM traffic of the board_advance kernels inside the headline bench: separate --pmc passes (FETCH_SIZE, WRITE_SIZE)
# with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md §HBM / §rocprofv3 prescribes.
# Writes gpurun_out/<tag>_pmc_traffic.json (bytes per launch, raw and corrected).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_board_advance|k_history_shift|k_advance_planes|k_stem_packed" --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-baseline 0 --saturated 0 --steady-state 0 --avg8-leg 0 > /tmp/pmc_$c.log 2>&1
  echo "pass $c exit=$?"
done
python3 - <<PY
import csv, glob, json, collections
out = {"command": "rocprofv3 --pmc <C> --kernel-trace --kernel-include-regex 'k_board_advance|k_history_shift|k_advance_planes' -- python3 bench.py --steps 1 --warmup 0 --cpu-baseline 0 --saturated 0",
       "note": "FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 FETCH_SIZE under-counts wide (16 B/lane) reads by 2x (guide): corrected = 2 x raw for the read side"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for f in glob.glob("/tmp/pmc_%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                per[r["Kernel_Name"].split("(")[0][-24:]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        v = [x for x in v if x > 0] or [0.0]
        out["%s_KiB_per_launch[%s]" % (c, k)] = {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)}
fs = sum(v["mean"] for k, v in out.items() if k.startswith("FETCH_SIZE"))
ws = sum(v["mean"] for k, v in out.items() if k.startswith("WRITE_SIZE"))
out["traffic_bytes_per_launch_raw"] = (fs + ws) * 1024
out["traffic_bytes_per_launch_corrected"] = (2 * fs + ws) * 1024
json.dump(out, open("$R/gpurun_out/${TAG}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
