#!/bin/bash
# configs[4] stepper (tools/sim3d_only.py: 8 grids of 512 x 512 x 64, Jacobi-20): kernel trace + stats, then HBM (FETCH_SIZE, WRITE_SIZE)
# and SQ counters in separate PMC passes (kernel-trace only; python3 directly after `--`).
# Summary: gpurun_out/<tag>/sim3d_pmc.json (per kernel: us per launch, launches per step, HBM bytes per launch and per step);
# copy into profiles/ what should be judged.   usage: tools/pmc_sim3d.sh <tag> [steps] [sq]
set -u
TAG=${1:-sim3d}; STEPS=${2:-6}; SQ=${3:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
P="$R/tools/sim3d_only.py $STEPS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $P > $OUT/trace.log 2>&1 || echo "trace pass failed"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $P > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $P > $OUT/write.log 2>&1 || echo "write pass failed"
if [ -n "$SQ" ]; then
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq -- python3 $P > $OUT/sq.log 2>&1 || echo "sq pass failed"
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- python3 $P > $OUT/sq2.log 2>&1 || echo "sq2 pass failed"
fi
cd $R
python3 tools/pmc_sim3d_summary.py $OUT $STEPS > $OUT/sim3d_pmc.json
cat $OUT/sim3d_pmc.json
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/sim3d_kernel_stats.csv
