#!/bin/bash
python -m pytest tests/ -x -q -m gpu > gpurun_out/r4_pytest_full.log 2>&1; tail -4 gpurun_out/r4_pytest_full.log
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_full.json 2> gpurun_out/r4_bench_full.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.load(open("gpurun_out/r4_bench_full.json"))
print({k:d[k] for k in ("value","ms_per_step","ms_sim_per_step","ms_encode_per_step")})
print("dense", d.get("encode_only_dense"))
print("inference", d.get("inference_ms_per_frame"))
c=d.get("config4",{}); print("config4", {k:c.get(k) for k in ("value","ms_per_step","ms_sim_per_step","ms_encode_per_volume","ms_encode_per_volume_dense","error")}, c.get("roofline_stencil",{}).get("frac_measured"))
t=d.get("train_step",{}); print("train", {k:t.get(k) for k in ("ms_per_step","allreduce_flat","direct_exchange_flat","error")})
print("roofline", d["roofline"]["frac"], d["roofline_stencil"].get("frac_measured"), d["cpu_baseline"]["value"] if "cpu_baseline" in d else None)
PY
