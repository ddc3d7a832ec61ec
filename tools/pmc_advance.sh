#!/bin/bash
# PMC passes for the board_advance kernel (separate runs, --kernel-trace only, as the MI355X guide prescribes).
# usage (on the GPU box): bash tools/pmc_advance.sh <outdir>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-/tmp/pmc_adv}
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --kernel-include-regex "k_advance_planes|k_history_shift|k_advance_legal" --output-format csv -d $OUT/$tag -- python3 $R/tools/bench_advance.py --iters 2 --plies 120 --n 1048576 > $OUT/$tag.log 2>&1 || echo "pass $tag failed" >> $OUT/failed.log
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        w = v[-2:]   # the timed launches (last ones)
        print(f.split("/")[-3], k[0], k[1], "n=%d" % len(v), "avg_last=%.5g" % (sum(w) / len(w)))
PY
