#!/bin/bash
# configs[4] encoder (tools/enc3d_probe.py: one 512 x 512 x 64 volume, twice): kernel trace + stats, then HBM and SQ counters in separate
# PMC passes (kernel-trace only).  Summary: gpurun_out/<tag>/enc3d_pmc.json ; copy into profiles/ what should be judged.
set -u
TAG=${1:-enc3d}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
P=$R/tools/enc3d_probe.py
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $P > $OUT/trace.log 2>&1 || echo "trace pass failed"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $P > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $P > $OUT/write.log 2>&1 || echo "write pass failed"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -- python3 $P > $OUT/sq.log 2>&1 || echo "sq pass failed"
cd $R
python3 - <<PY
import csv, glob, collections, json
out = collections.defaultdict(dict)
for p in ("trace", "fetch", "write", "sq"):
    dur = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv" % p, recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row["Kernel_Name"].split("(")[0].split("::")[-1][:40]].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0].split("::")[-1][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in dur.items():
        if "march" in k or "pool3d" in k:
            out[k]["launches_" + p] = len(v)
            out[k]["us_per_launch_" + p] = round(sum(v) / len(v) / 1e3, 1)
    for k, d in acc.items():
        if "march" in k or "pool3d" in k:
            out[k].update({c: round(sum(v) / len(v)) for c, v in d.items()})
# FETCH_SIZE / WRITE_SIZE: raw counter units as rocprofv3 reports them (MI355X_MICROARCH.md: KiB-like units with the gfx950 corrections
# applied by tools/pmc_traffic.py for the headline; here the raw averages per launch are kept beside the algorithmic bytes)
json.dump(out, open("$OUT/enc3d_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/enc3d_kernel_stats.csv
