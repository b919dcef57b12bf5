#!/bin/bash
# scratch: one GPU call of round 4 (tests + 3-D counters + config4 leg)
python -m pytest tests/test_train_main.py tests/test_rccl_world1.py -x -q > gpurun_out/r4_tests_a.log 2>&1; tail -5 gpurun_out/r4_tests_a.log
bash tools/pmc_sim3d.sh r4_sim3d_v2 6 > gpurun_out/r4_sim3d_v2.log 2>&1
python3 - <<PY
import json
d=json.load(open("gpurun_out/r4_sim3d_v2/sim3d_pmc.json")); print(json.dumps(d["per_step"]))
for k,e in d["kernels"].items(): print(k, e.get("launches_per_step"), e.get("us_per_launch_trace"), round(e.get("hbm_fetch_bytes_per_launch",0)/1e6), round(e.get("hbm_write_bytes_per_launch",0)/1e6), e.get("hbm_GBps"))
PY
python3 tools/sim3d_probe.py 6 > gpurun_out/r4_config4_probe.json 2> gpurun_out/r4_config4_probe.err
python3 - <<PY
import json; d=json.load(open("gpurun_out/r4_config4_probe.json")); print({k:d[k] for k in ("value","ms_per_step","ms_sim_per_step","ms_encode_per_volume","ms_encode_per_volume_dense")}); print(d["roofline_stencil"].get("frac_measured"), d["roofline_encoder"]["frac"], d["roofline_encoder"]["frac_dense"])
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_inf_b4 -- python3 $GRAFT_REPO_ROOT/tools/graph_b1_trace.py 4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r4_inf_b4 -name "*kernel_stats.csv" | xargs head -30 | cut -c1-170
