"""Time k_encoder_b16 (tokens, bf16x3, 64 frames of 256^2 from the simulator) on whichever libsmokehip SMOKEHIP_LIB names.
Used by tools/enc_ablate.sh with the -DSMK_ENC_ABLATE builds (timing only: their features are wrong by construction)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from smokephysai_amd.models.encoder import HipEncoder          # noqa: E402
from smokephysai_amd.physics import SmokeSimulator             # noqa: E402

dev = torch.device("cuda", 0)
B, N = 64, 256
sim = SmokeSimulator((N, N), device=dev, batch_size=B, jacobi_iters=20)
sim.ns_solver.add_smoke_sources(bench.draw_sources(B, N, seed=0))
frame = torch.empty(B, N, N, device=dev)
for _ in range(10):
    sim.ns_solver.step_into(frame, 1, add_fractal=True, fractal_intensity=0.05)
enc = HipEncoder(bench.encoder_weights(0), device=dev)
for _ in range(30):
    enc.tokens(frame, input_dim=128, dtype="bf16x3")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(81)]
ev[0].record()
for k in range(80):
    enc.tokens(frame, input_dim=128, dtype="bf16x3")
    ev[k + 1].record()
torch.cuda.synchronize()
ms = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(80))
print("%s %.4f %.4f" % (os.environ.get("SMOKEHIP_LIB", "product").split("/")[-1], ms[40], sum(ms) / 80))
