#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes written by tools/profile.sh into per-launch HBM traffic (bytes).

usage: pmc_traffic.py <profile_dir> <bench_steps>   -> JSON on stdout
FETCH_SIZE / WRITE_SIZE are in KiB-units of the TCC_EA request counters; on gfx950 FETCH_SIZE under-reports wide
streaming reads, so both counters are rescaled by the factor measured on the calibration launch (known bytes) in the
same dword-per-lane access pattern, as MI355X_MICROARCH.md (HBM) prescribes.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read_counters(d):
    """kernel name -> list of counter values (one per dispatch)."""
    out = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                out[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return out


def build_stamp():
    """What the counters were taken on: hash of the kernel sources + ABI header (bench.py compares it with its own build)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.source_stamp()


def main():
    root, steps = sys.argv[1], int(sys.argv[2])
    dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
    cal_bytes = 1536 * 256 * 256 * 4.0
    res = {"units": "bytes per launch (stencil: per time step of 64 grids)", "stamp": build_stamp(), "encoder_dtype": dtype,
           "source": "tools/profile.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), fetch x calibration factor",
           "calibration": {}}
    factors = {}
    for name, sub in (("fetch", "cal_fetch"), ("write", "cal_write")):
        c = read_counters(os.path.join(root, sub))
        vals = [v for k, vs in c.items() if "k_apply_fractal" in k for v in vs]
        raw = (sum(vals) / len(vals)) * 1024.0 if vals else None
        factors[name] = cal_bytes / raw if raw else None
        res["calibration"][name] = {"known_bytes": cal_bytes, "counter_bytes": raw, "factor": factors[name]}
    fetch = read_counters(os.path.join(root, "pmc_fetch"))
    write = read_counters(os.path.join(root, "pmc_write"))
    detail = {}
    sten = {"fetch": 0.0, "write": 0.0}
    enc = {"fetch": 0.0, "write": 0.0}
    for kind, table in (("fetch", fetch), ("write", write)):
        f = factors[kind] or 1.0
        for k, vs in table.items():
            if "smk::" not in k:
                continue
            short = k.split("(")[0].replace("void ", "")
            total = sum(vs) * 1024.0 * f
            detail.setdefault(short, {})[kind + "_bytes_per_dispatch"] = total / len(vs)
            detail[short]["dispatches"] = len(vs)
            if "k_encoder" in k:
                enc[kind] += total / len(vs)
            elif any(s in k for s in ("k_jacobi", "k_buoy", "k_advect", "k_grad", "k_divergence", "k_step", "k_copy_cells")):
                sten[kind] += total                      # summed over the run, normalised per step below
    nsteps = steps + 2                                   # bench warm-up steps are profiled too
    res["stencil"] = (sten["fetch"] + sten["write"]) / nsteps
    res["encoder"] = enc["fetch"] + enc["write"]
    res["stencil_split"] = {k: v / nsteps for k, v in sten.items()}
    res["encoder_split"] = enc
    res["detail"] = detail
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
