#!/bin/bash
# SQ counters of the 3-D advection launch inside the configs[4] step (tools/sim3d_only.py); env (SMK_ADVECT3_*) passes through.
# usage: tools/prof_advect3d.sh <tag>
set -u
TAG=${1:-adv3}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
P="$R/tools/sim3d_only.py 4"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/p1 -- python3 $P > $OUT/p1.log 2>&1 || echo p1 failed
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p2 -- python3 $P > $OUT/p2.log 2>&1 || echo p2 failed
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_WAVES SQ_INSTS_BRANCH --output-format csv -d $OUT/p3 -- python3 $P > $OUT/p3.log 2>&1 || echo p3 failed
cd $R
python3 - <<PY
import csv, glob, collections, json
out = {}
for p in ("p1", "p2", "p3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv" % p, recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row["Kernel_Name"].split("(")[0][-40:]].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for row in csv.DictReader(open(f)):
            if "k3_advect" in row["Kernel_Name"]:
                acc[row["Kernel_Name"].split("(")[0][-40:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        e = out.setdefault(k, {})
        e.update({c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()})
        if dur.get(k): e["us_per_launch_" + p] = round(sum(dur[k]) / len(dur[k]) / 1e3, 1)
json.dump(out, open("$OUT/sq_summary.json", "w"), indent=1, sort_keys=True)
for k, e in out.items(): print(k, json.dumps(e, sort_keys=True))
PY
