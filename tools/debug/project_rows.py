#!/usr/bin/env python3
"""Debug aid: one pressure projection on random (u, v, p) for B grids vs the CPU oracle; prints which rows / columns differ."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from smokephysai_amd import _lib
from smokephysai_amd.physics import NavierStokesSimulator

H, W, J, B = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (256, 256, 100, 64)))
rng = np.random.RandomState(0)
ns = NavierStokesSimulator((H, W), batch_size=B, jacobi_iters=J)
u = (rng.randn(B, H + 1, W) * 0.01).astype(np.float32)
v = (rng.randn(B, H, W + 1) * 0.01).astype(np.float32)
p = (rng.randn(B, H, W) * 0.01).astype(np.float32)
p[:, 0] = 0; p[:, -1] = 0; p[:, :, 0] = 0; p[:, :, -1] = 0
ns.u, ns.v, ns.p = torch.from_numpy(u), torch.from_numpy(v), torch.from_numpy(p)
# STAGE_PROJECT works on the scratch copies (u2, v2): run the diffuse stage with zero viscosity? no -- use the stage chain on a real state:
ns2 = NavierStokesSimulator((H, W), batch_size=B, jacobi_iters=J)
ns2.u, ns2.v, ns2.p = torch.from_numpy(u), torch.from_numpy(v), torch.from_numpy(p)
ns2.density = torch.from_numpy(np.abs(rng.randn(B, H, W)).astype(np.float32))
dens = ns2.density.cpu().numpy().copy()
ns2.run_stage(_lib.STAGE_BUOY_DIFFUSE)
ns2.run_stage(_lib.STAGE_PROJECT)
ns2.run_stage(_lib.STAGE_ADVECT_U); ns2.run_stage(_lib.STAGE_ADVECT_V); ns2.run_stage(_lib.STAGE_ADVECT_D)
torch.cuda.synchronize()
for b in (0, B // 2, B - 1):
    o = oracle.OracleNS((H, W), jacobi_iters=J)
    o.u, o.v, o.p, o.density = u[b].copy(), v[b].copy(), p[b].copy(), dens[b].copy()
    o.step()
    for k in ("p", "u", "v", "density"):
        a = getattr(ns2, k)[b].cpu().numpy(); r = getattr(o, k)
        bad = a != r
        rows = np.where(bad.any(axis=1))[0]; cols = np.where(bad.any(axis=0))[0]
        print(f"grid {b} {k}: {bad.sum()} cells differ; rows {rows[:40].tolist()}{'...' if len(rows) > 40 else ''}; "
              f"cols {cols[:12].tolist()}{'...' if len(cols) > 12 else ''} maxabs {np.abs(a - r).max():.3e}")
# pattern of the p mismatch of the last grid: first / last bad row and rows that are fully right
a = ns2.p[B - 1].cpu().numpy(); r = o.p
bad = (a != r)
good_rows = np.where(~bad[:, 1:-1].any(axis=1))[0]
print("rows of p without any mismatch:", good_rows.tolist()[:80])
print("mismatch count per row (first 70 rows):", bad.sum(axis=1)[:70].tolist())
