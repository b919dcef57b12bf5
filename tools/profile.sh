#!/bin/bash
# Profile the headline bench on the GPU box (run through gpurun from the repo root):
#   pass 1: kernel trace + stats;  pass 2: --pmc FETCH_SIZE;  pass 3: --pmc WRITE_SIZE  (PMC passes carry no trace domains
#   other than --kernel-trace); plus the same two PMC passes on a calibration kernel with a known byte count.
# Summaries land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-prof}
STEPS=${2:-150}   # >= 100: the first ~20 launches of a fresh process run 10-25 % slower (clock ramp); a short profile overstates the average
DTYPE=${3:-bf16x3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="$R/bench.py --steps $STEPS --warmup 2 --cpu-frames 0 --no-inference --no-alt --no-config1 --no-dataset --no-config4 --no-train-step --encoder-dtype $DTYPE"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || echo "trace pass failed"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1 || echo "write pass failed"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- python3 $R/tools/pmc_calibrate.py > $OUT/cal_fetch.log 2>&1 || echo "cal fetch failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- python3 $R/tools/pmc_calibrate.py > $OUT/cal_write.log 2>&1 || echo "cal write failed"
cd $R
python3 tools/pmc_traffic.py $OUT $STEPS $DTYPE > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err
cat $OUT/pmc_traffic.json
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
