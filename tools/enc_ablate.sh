#!/bin/bash
# Where k_encoder_b16's time goes: the product kernel against builds with parts removed (-DSMK_ENC_ABLATE=bits: 1 conv1 only on a
# workgroup's first tile, 2 no BN/ReLU/pool epilogue, 4 no workgroup barriers, 8 no conv2 MFMAs), interleaved, two rounds.
# Build first (build container): see tools/README.md.  Output: "<library> <median ms> <mean ms>" per run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  python3 $R/tools/enc_ablate_probe.py || exit 1
  for n in 1 2 3 4 7 8 11; do
    SMOKEHIP_LIB=$R/tools/probes/bin/libsmokehip_abl$n.so python3 $R/tools/enc_ablate_probe.py || exit 1
  done
done
