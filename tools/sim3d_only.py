"""configs[4] stepper alone (no encoder): 8 grids of 512 x 512 x 64, Jacobi-20, `steps` timed steps after 2 warm-up steps.
Target of tools/pmc_sim3d.sh (rocprofv3 kernel trace / FETCH_SIZE / WRITE_SIZE / SQ passes); prints ms per step."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from smokephysai_amd.physics import NavierStokesSimulator3D

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
B, D, H, W, J = int(os.environ.get("SMK_SIM3D_B", "8")), 64, 512, 512, 20          # SMK_SIM3D_B: fewer grids (is one grid's projection Infinity-Cache resident?)
sim = NavierStokesSimulator3D((D, H, W), device=dev, batch_size=B, jacobi_iters=J)
rng = np.random.RandomState(4)
sim.add_smoke_sources([(b, int(rng.randint(40, W - 40)), int(rng.randint(40, H - 40)), int(rng.randint(10, D - 10)), 8,
                        float(rng.uniform(0.5, 2.0))) for b in range(B) for _ in range(3)])
frame = torch.empty(B, D, H, W, device=dev)
for _ in range(2):
    sim.step_into(frame, 1)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
ev[0].record()
for k in range(steps):
    sim.step_into(frame, 1)
    ev[k + 1].record()
torch.cuda.synchronize(dev)
ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]
print(json.dumps({"ms_per_step": float(np.mean(ms)), "min": float(np.min(ms)), "max": float(np.max(ms)), "steps": steps,
                  "checksum": float(frame.double().sum())}))
