#!/bin/bash
# per-kernel durations of the board_advance pair at a given ply (last 20 dispatches = the timed iterations)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for ply in "$@"; do
  rm -rf /tmp/tr_adv
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_adv -- python3 $R/tools/bench_advance.py --plies $ply --iters 20 > /tmp/tr_adv.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/tr_adv/*/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:48]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("Scratch_Size")))
for k, v in d.items():
    if "sgo::" not in k: continue
    v.sort()
    w = [x[1] for x in v[-43:-23]]   # the split pair's timed iterations come before the fused in-place section
    w2 = [x[1] for x in v[-20:]]
    print("ply $ply", k, "n=%d" % len(v), "avg_us(timed split section)=%.1f" % (sum(w) / max(1, len(w)) / 1e3), "avg_us(last20)=%.1f" % (sum(w2) / len(w2) / 1e3), "vgpr", v[-1][2], "scratch", v[-1][4])
PY
done
