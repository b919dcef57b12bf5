#!/bin/bash
# SQ counters of the encoder kernel (separate PMC passes, kernel-trace only).
set -u
TAG=${1:-pmcenc}; DTYPE=${2:-bf16x3}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
ARGS="$R/bench.py --steps 6 --warmup 1 --cpu-frames 0 --no-inference --no-alt --encoder-dtype $DTYPE"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1 || echo p1 failed
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1 || echo p2 failed
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/p3 -- python3 $ARGS > $OUT/p3.log 2>&1 || echo p3 failed
cd $R
python3 - <<PY
import csv,glob,collections
for p in ("p1","p2","p3"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%p, recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_encoder" in row["Kernel_Name"] or "k_jacobi_band" in row["Kernel_Name"]:
                acc[row["Kernel_Name"].split("(")[0][-40:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,d in acc.items():
        print(p,k,{c: round(sum(v)/len(v)) for c,v in d.items()})
PY
tail -3 $OUT/p1.log
