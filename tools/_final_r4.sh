#!/bin/bash
# round 4: every judged artefact from ONE box (copied into profiles/r04 afterwards)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4_final_bench.json 2> gpurun_out/r4_final_bench.err; echo "bench rc=$?"
bash tools/profile.sh r4_final 150 bf16x3 > gpurun_out/r4_final_profile.log 2>&1; echo profile done
bash tools/pmc_encoder.sh r4_final_sq bf16x3 > gpurun_out/r4_final_sq.log 2>&1; echo sq done
bash tools/pmc_sim3d.sh r4_final_sim3d 6 sq > gpurun_out/r4_final_sim3d.log 2>&1; echo sim3d done
bash tools/pmc_enc3d.sh r4_final_enc3d > gpurun_out/r4_final_enc3d.log 2>&1; echo enc3d done
bash tools/profile_inference.sh r4_final_inf > gpurun_out/r4_final_inf.log 2>&1; echo inference done
