#!/bin/bash
# Kernel-trace profile of the full SmokePhysNet forward (metric M2) on the GPU box:
#   gpurun -- bash tools/profile_inference.sh <tag>
# 10 eager eval forwards at batch 64 and batch 1 (256^2 frames) under rocprofv3 --kernel-trace --stats, plus the
# linear / attention probe outputs.  Summaries land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-infprof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for bs in 64 4 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_b$bs -- python3 $R/tools/inference_profile.py $bs 10 > $OUT/trace_b$bs.log 2>&1 || echo "trace b$bs failed"
  find $OUT/trace_b$bs -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/inference_b${bs}_kernel_stats.csv
done
cd $R
python3 tools/linear_probe.py > $OUT/linear_probe.txt 2>&1
python3 tools/attention_probe.py > $OUT/attention_probe.txt 2>&1
python3 tools/inference_probe.py 2>&1 | grep -v "max |eager" > $OUT/inference_probe.txt
python3 -c "import json,sys; sys.path.insert(0, '$R'); import bench; json.dump(bench.source_stamp(), open('$OUT/inference_STAMP.json', 'w'))"
ls $OUT
