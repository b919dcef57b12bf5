"""Summarise the passes of tools/pmc_sim3d.sh: per k3_* kernel the time per launch, launches per step and HBM bytes.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts 128-byte requests at
64 bytes -- doubled here (the factor tools/pmc_calibrate.py measures on a known dword-per-lane stream: 1.997); WRITE_SIZE is exact."""
import csv, glob, collections, json, os, sys

out_dir, steps = sys.argv[1], int(sys.argv[2])
nsteps = steps + 2                                        # the probe's two warm-up steps are profiled too
FETCH_FACTOR = 2.0


def short(name):
    return name.split("(")[0].replace("void ", "").split("::")[-1]


res = collections.defaultdict(dict)
for p in ("trace", "fetch", "write", "sq", "sq2"):
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out_dir, p, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            dur[short(row["Kernel_Name"])].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out_dir, p, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in dur.items():
        if k.startswith("k3_") and "zero" not in k and "add_sources" not in k:
            res[k]["launches_per_step"] = round(len(v) / nsteps, 3)
            res[k]["us_per_launch_" + p] = round(sum(v) / len(v) / 1e3, 1)
    for k, d in acc.items():
        if k in res:
            for c, v in d.items():
                res[k][c] = sum(v) / len(v)
tot = {"fetch": 0.0, "write": 0.0, "us": 0.0}
for k, e in res.items():
    if "FETCH_SIZE" in e:
        e["hbm_fetch_bytes_per_launch"] = e.pop("FETCH_SIZE") * 1024.0 * FETCH_FACTOR
        tot["fetch"] += e["hbm_fetch_bytes_per_launch"] * e["launches_per_step"]
    if "WRITE_SIZE" in e:
        e["hbm_write_bytes_per_launch"] = e.pop("WRITE_SIZE") * 1024.0
        tot["write"] += e["hbm_write_bytes_per_launch"] * e["launches_per_step"]
    if "us_per_launch_trace" in e:
        tot["us"] += e["us_per_launch_trace"] * e["launches_per_step"]
        b = e.get("hbm_fetch_bytes_per_launch", 0.0) + e.get("hbm_write_bytes_per_launch", 0.0)
        if b:
            e["hbm_GBps"] = round(b / (e["us_per_launch_trace"] * 1e-6) / 1e9, 1)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    stamp = bench.source_stamp()
except Exception as ex:                                   # noqa: BLE001
    stamp = {"error": str(ex)}
cells = 8 * 64 * 512 * 512
print(json.dumps({"workload": "configs[4] stepper: 8 x 512x512x64, Jacobi-20 (tools/sim3d_only.py)", "stamp": stamp,
                  "fetch_factor": FETCH_FACTOR, "cells_per_step": cells,
                  "per_step": {"hbm_fetch_bytes": tot["fetch"], "hbm_write_bytes": tot["write"], "hbm_bytes": tot["fetch"] + tot["write"],
                               "kernel_us": round(tot["us"], 1), "bytes_per_cell": (tot["fetch"] + tot["write"]) / cells,
                               "hbm_GBps_over_kernel_time": round((tot["fetch"] + tot["write"]) / (tot["us"] * 1e-6) / 1e9, 1) if tot["us"] else None,
                               "pass_model_bytes": cells * 4.0 * (37 + 3 * 20)},
                  "kernels": res}, indent=1, sort_keys=True))
