#!/bin/bash
# copy what tools/_final_r4.sh left under gpurun_out/ into profiles/ and stamp it (build container, after the GPU call)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
cp gpurun_out/r4_final/kernel_stats.csv profiles/r04/final_kernel_stats.csv
cp gpurun_out/r4_final/pmc_traffic.json profiles/r04/final_pmc_traffic.json; cp gpurun_out/r4_final/pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/r4_final_sq/summary.json profiles/r04/final_pmc_sq.json; cp gpurun_out/r4_final_sq/pmc_mfma.json profiles/pmc_mfma.json
cp gpurun_out/r4_final_sim3d/sim3d_pmc.json profiles/r04/sim3d_pmc.json; cp gpurun_out/r4_final_sim3d/sim3d_kernel_stats.csv profiles/r04/sim3d_config4_kernel_stats.csv
cp gpurun_out/r4_final_enc3d/enc3d_kernel_stats.csv gpurun_out/r4_final_enc3d/enc3d_pmc.json profiles/r04/
for f in attention_probe.txt inference_STAMP.json inference_b1_kernel_stats.csv inference_b4_kernel_stats.csv inference_b64_kernel_stats.csv inference_probe.txt linear_probe.txt; do cp gpurun_out/r4_final_inf/$f profiles/r04/$f; done
python3 tools/stamp_profiles.py
