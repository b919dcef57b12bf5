#!/bin/bash
# interleaved A/B of the conv1 turns of k_encoder_b16 (SMK_ENC_TURNS), plus the CU census
R=${GRAFT_REPO_ROOT:-$(pwd)}
$R/tools/probes/bin/cu_census
for round in 1 2 3; do
  for tv in 1 0; do
    printf "turns=$tv "; SMK_ENC_TURNS=$tv python3 $R/tools/enc_ablate_probe.py || exit 1
  done
done
