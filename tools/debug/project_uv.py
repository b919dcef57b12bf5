#!/usr/bin/env python3
"""Debug aid (SMK_DEBUG_DUMP_PROJECT=1): the projected (u2, v2, p) of the HIP path vs the oracle's, cell by cell."""
import os, sys
import numpy as np, torch
os.environ["SMK_DEBUG_DUMP_PROJECT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from smokephysai_amd import _lib
from smokephysai_amd.physics import NavierStokesSimulator

H, W, J, B = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (256, 256, 100, 64)))
rng = np.random.RandomState(0)
u = (rng.randn(B, H + 1, W) * 0.01).astype(np.float32)
v = (rng.randn(B, H, W + 1) * 0.01).astype(np.float32)
p = (rng.randn(B, H, W) * 0.01).astype(np.float32)
p[:, 0] = 0; p[:, -1] = 0; p[:, :, 0] = 0; p[:, :, -1] = 0
ns = NavierStokesSimulator((H, W), batch_size=B, jacobi_iters=J)
ns.u, ns.v, ns.p = torch.from_numpy(u), torch.from_numpy(v), torch.from_numpy(p)
ns.density = torch.from_numpy(np.abs(rng.randn(B, H, W)).astype(np.float32))
dens = ns.density.cpu().numpy().copy()
ns.run_stage(_lib.STAGE_BUOY_DIFFUSE)
ns.run_stage(_lib.STAGE_PROJECT)
torch.cuda.synchronize()
tot = 0
for b in range(B):
    o = oracle.OracleNS((H, W), jacobi_iters=J)
    o.u, o.v, o.p, o.density = u[b].copy(), v[b].copy(), p[b].copy(), dens[b].copy()
    o.buoyancy()
    o.u = o.diffusion_step(o.u, o.viscosity); o.v = o.diffusion_step(o.v, o.viscosity)
    o.pressure_projection()
    for k in ("p", "u", "v"):
        a = getattr(ns, k)[b].cpu().numpy(); r = getattr(o, k)
        bad = np.argwhere(a != r)
        tot += len(bad)
        if len(bad) and tot < 400:
            print(f"grid {b} {k}: {len(bad)} cells; first: " + "; ".join(f"({i},{j}) got {a[i,j]:.6g} want {r[i,j]:.6g}" for i, j in bad[:6]))
print("total mismatching cells:", tot)
