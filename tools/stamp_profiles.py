#!/usr/bin/env python3
"""Add the git HEAD (unknown on the GPU box: .git does not travel) to the stamps of profiles/pmc_traffic.json and profiles/pmc_mfma.json,
and report whether their source hash matches this tree.  Run in the build container after copying the extracts from gpurun_out/."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

now = bench.source_stamp()
head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
for name in ("pmc_traffic.json", "pmc_mfma.json"):
    path = os.path.join(ROOT, "profiles", name)
    d = json.load(open(path))
    st = d.setdefault("stamp", {})
    st.setdefault("head_at_collection", head)
    json.dump(d, open(path, "w"), indent=1)
    print(f"{name}: stamp {st} -> {'current' if st.get('csrc_sha256') == now['csrc_sha256'] else 'STALE vs this tree ' + now['csrc_sha256']}")
