#!/bin/bash
# Kernel-trace stats (+ optional SQ counters) of the stencil-only bench on the GPU box.  usage: tools/prof_stencil.sh <tag> [pmc]
set -u
TAG=${1:-sten}; PMC=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
ARGS="$R/bench.py --steps 100 --warmup 3 --cpu-frames 0 --no-inference --no-alt --no-config1 --no-dataset --no-config4 --no-train-step --no-encode"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || echo "trace pass failed"
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
if [ -n "$PMC" ]; then
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1 || echo p1 failed
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1 || echo p2 failed
  cd $R
  python3 - <<PY
import csv, glob, collections, json
out = {}
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv" % p, recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row["Kernel_Name"].split("(")[0][-44:]].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for row in csv.DictReader(open(f)):
            if "smk::" in row["Kernel_Name"]:
                acc[row["Kernel_Name"].split("(")[0][-44:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        e = out.setdefault(k, {})
        e.update({c: round(sum(v) / len(v)) for c, v in d.items()})
        if dur.get(k): e["ns_per_launch_" + p] = round(sum(dur[k]) / len(dur[k])); e["launches"] = len(dur[k])
json.dump(out, open("$OUT/sq_summary.json", "w"), indent=1, sort_keys=True)
for k, e in out.items(): print(k, e)
PY
fi
cd $R; cat $OUT/kernel_stats.csv | cut -c1-160 | head -14
