#!/bin/bash
# Kernel-trace stats of the train.py step (tools/train_probe.py: batch 64 of 256^2 frames, 3 timed steps).  usage: tools/profile_train.sh <tag>
set -u
TAG=${1:-train}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/train_probe.py 64 256 3 > $OUT/trace.log 2>&1 || echo "trace pass failed"
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
cd $R
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel sum {tot / 1e6:.1f} ms over the whole process (2 warm-up + 3 timed steps + set-up)")
for r in rows[:40]:
    print(f"{r['Name'][:100]:100s} n {int(r['Calls']):5d} {float(r['TotalDurationNs']) / 1e6:8.2f} ms {float(r['Percentage']):5.1f}%")
PY
tail -3 $OUT/trace.log
