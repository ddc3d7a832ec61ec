#!/bin/bash
# rocprofv3 kernel-trace + stats of the headline bench (1 timed step); writes the stats CSV into gpurun_out/
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
shift
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_bench
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-baseline 0 --saturated 0 --steady-state 0 --avg8-leg 0 "$@" > $R/gpurun_out/prof_${TAG}.log 2>&1
echo exit=$? >> $R/gpurun_out/prof_${TAG}.log
cp /tmp/prof_bench/*/*kernel_stats.csv $R/gpurun_out/${TAG}_bench_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/${TAG}_bench_kernel_stats.csv")))
for r in rows[:12]:
    print("%-60s calls %6s total_ms %9.1f avg_us %9.1f %6s%%" % (r["Name"][:60], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
for r in rows:
    if "sgo::" in r["Name"]:
        print("%-60s calls %6s total_ms %9.2f avg_us %9.1f" % (r["Name"][:60], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
tail -2 $R/gpurun_out/prof_${TAG}.log | cut -c1-1200
