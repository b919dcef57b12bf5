#!/bin/bash
# Where k_attention_x3's time goes: the product kernel against builds with parts removed (-DSMK_ATT_ABLATE=bits: 1 no softmax arithmetic,
# 2 no split arithmetic in the K / V staging (same LDS stores), 4 MFMAs replaced by one vector op each), interleaved, two rounds.
# Build first (build container): tools/README.md.  Output of tools/attention_probe.py per library.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  echo "== product"; python3 $R/tools/attention_probe.py hip || exit 1
  for n in 1 2 3 4 7; do
    echo "== ablate $n"; SMOKEHIP_LIB=$R/tools/probes/bin/libsmokehip_att$n.so python3 $R/tools/attention_probe.py hip || exit 1
  done
done
