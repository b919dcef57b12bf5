#!/bin/bash
# SQ counters of the inference forward at batch 64 (k_linear_b16, k_attention_x3, k_encoder_b16, k_layernorm): separate PMC passes, kernel-trace only.
set -u
TAG=${1:-pmcbody}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
ARGS="$R/tools/inference_profile.py 64 10"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1 || echo p1 failed
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1 || echo p2 failed
cd $R
python3 - <<PY
import csv, glob, collections, json
out = {}
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv" % p, recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row["Kernel_Name"].split("(")[0][-40:]].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for row in csv.DictReader(open(f)):
            if "smk::" in row["Kernel_Name"]:
                acc[row["Kernel_Name"].split("(")[0][-40:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        e = out.setdefault(k, {}); e.update({c: round(sum(v) / len(v)) for c, v in d.items()})
        if dur.get(k): e["ns_per_launch_" + p] = round(sum(dur[k]) / len(dur[k]))
for k, e in out.items():
    if "GRBM_GUI_ACTIVE" in e and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
        cyc = e["GRBM_GUI_ACTIVE"] / 8.0
        e["mfma_pipe_busy_frac"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)
        e["effective_clock_GHz"] = round(cyc / e["ns_per_launch_p2"], 3)
        e["valu_per_mfma"] = round(e["SQ_INSTS_VALU"] / max(e["SQ_INSTS_MFMA"], 1), 2)
    print(k, e)
json.dump(out, open("$OUT/summary.json", "w"), indent=1, sort_keys=True)
PY
