#!/bin/bash
# SQ counters of the encoder kernel (separate PMC passes, kernel-trace only).
set -u
TAG=${1:-pmcenc}; DTYPE=${2:-bf16x3}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
ARGS="$R/bench.py --steps 40 --warmup 5 --cpu-frames 0 --no-inference --no-alt --no-config1 --no-dataset --no-config4 --no-train-step --encoder-dtype $DTYPE"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1 || echo p1 failed
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1 || echo p2 failed
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/p3 -- python3 $ARGS > $OUT/p3.log 2>&1 || echo p3 failed
cd $R
python3 - <<PY
import csv, glob, collections, json
out = {}
for p in ("p1", "p2", "p3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv" % p, recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row["Kernel_Name"].split("(")[0][-40:]].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_encoder" in row["Kernel_Name"] or "k_jacobi_band" in row["Kernel_Name"]:
                acc[row["Kernel_Name"].split("(")[0][-40:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        e = out.setdefault(k, {})
        e.update({c: round(sum(v) / len(v)) for c, v in d.items()})
        if dur.get(k):
            e["ns_per_launch_" + p] = round(sum(dur[k]) / len(dur[k]))
for k, e in out.items():
    # MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the 1,024 SIMDs; GRBM_GUI_ACTIVE is summed
    # over the 8 XCDs, so kernel cycles = GRBM_GUI_ACTIVE / 8 and the effective clock is that over the launch duration
    if "GRBM_GUI_ACTIVE" in e and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
        cyc = e["GRBM_GUI_ACTIVE"] / 8.0
        e["mfma_pipe_busy_frac"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)
        if "ns_per_launch_p3" in e:
            e["effective_clock_GHz"] = round(cyc / e["ns_per_launch_p3"], 3)
    print(k, e)
import sys
sys.path.insert(0, "$R")
import bench
stamp = bench.source_stamp()
json.dump(dict(out, stamp=stamp), open("$OUT/summary.json", "w"), indent=1, sort_keys=True)
# the headline kernel's extract in the form bench.py passes through as roofline_encoder.pmc (copy to profiles/pmc_mfma.json)
for k, e in out.items():
    if "k_encoder_b16" in k and "mfma_pipe_busy_frac" in e:
        ex = {"kernel": "void smk::k_encoder_b16<8, true>", "stamp": stamp,
              "source": "tools/pmc_encoder.sh (rocprofv3 --kernel-trace --pmc, 3 separate passes, 45 launches each)",
              "ns_per_launch_under_counters": e.get("ns_per_launch_p3")}
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "effective_clock_GHz", "mfma_pipe_busy_frac", "SQ_INSTS_VALU", "SQ_INSTS_MFMA",
                  "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            if c in e: ex[c] = e[c]
        json.dump(ex, open("$OUT/pmc_mfma.json", "w"), indent=1)
PY
tail -3 $OUT/p1.log
