#!/bin/bash
# HBM traffic of the tower convolution inside the headline bench: separate --pmc passes (FETCH_SIZE, WRITE_SIZE) with
# --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Writes gpurun_out/<tag>_pmc_conv.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcc_$c
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_conv[48]w" --output-format csv -d /tmp/pmcc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-baseline 0 --saturated 0 --steady-state 0 --avg8-leg 0 > /tmp/pmcc_$c.log 2>&1
  echo "pass $c exit=$?"
done
python3 - <<PY
import csv, glob, json
out = {"command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --kernel-include-regex k_conv[48]w -- python3 bench.py --steps 1 --warmup 0 --cpu-baseline 0 --saturated 0 --steady-state 0",
       "note": "KiB per launch; the 8192-position launches are the upper half of the sorted values; gfx950 FETCH_SIZE under-counts 16 B/lane reads by 2x (guide): traffic = 2 x FETCH + WRITE; algorithmic x + skip + y + weights = 3.63 GB (no skip: 2.42 GB)"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob("/tmp/pmcc_%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                v.append(float(r["Counter_Value"]))
    big = sorted(v)[len(v) // 2:] or [0.0]
    out[c] = {"launches": len(v), "mean": sum(big) / len(big), "max": max(big)}
out["traffic_bytes_per_launch_corrected"] = (2 * out["FETCH_SIZE"]["mean"] + out["WRITE_SIZE"]["mean"]) * 1024
json.dump(out, open("$R/gpurun_out/${TAG}_pmc_conv.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
