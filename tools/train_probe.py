"""One-GPU time of the train.py step at BASELINE configs[3]'s per-GPU shape (batch 64 of 256x256 frames, full SmokePhysNet).

The batch dictionary is built on the device from the batched simulator exactly as SyntheticSmokeDataset would hand it over
(input / target / chaos_features / sequence); the step is train.py's: zero_grad, batch_losses, backward, clip 1.0, AdamW.
usage: train_probe.py [batch] [grid] [steps]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smokephysai_amd.models import SmokePhysNet
from smokephysai_amd.models.physics_regularizer import PhysicsRegularizer
from smokephysai_amd.physics import SmokeSimulator
from train import batch_losses

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)

sim = SmokeSimulator((N, N), device=dev, batch_size=B, jacobi_iters=20)
sim.ns_solver.add_smoke_sources([(b, N // 2, N // 2, 8, 1.0) for b in range(B)])
seq = torch.empty(B, 20, N, N, device=dev)
t0 = time.perf_counter()
sim.ns_solver.step_into(seq, 20, add_fractal=True, fractal_intensity=0.05)
torch.cuda.synchronize()
print(f"generated {B} x 20 frames of {N}^2 in {(time.perf_counter() - t0) * 1e3:.1f} ms")
batch = {"input": seq[:, 9:10].contiguous(), "target": seq[:, 10:11].contiguous(),
         "chaos_features": torch.rand(B, 3, device=dev), "sequence": seq}

model = SmokePhysNet().to(dev).train()
reg = PhysicsRegularizer()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)


def step():
    opt.zero_grad()
    total, recon, phys, chaos = batch_losses(model, reg, batch, dev)
    total.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
    opt.step()
    return total


for _ in range(2):
    loss = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print(f"train step: {ms:.1f} ms per batch of {B} ({ms / B:.3f} ms/frame, {B / ms * 1e3:.0f} frames/s), loss {loss.item():.5f}, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")

model.eval()
with torch.no_grad():
    for _ in range(2):
        batch_losses(model, reg, batch, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        batch_losses(model, reg, batch, dev)
    torch.cuda.synchronize()
print(f"validation step (HIP eval path): {(time.perf_counter() - t0) / steps * 1e3:.1f} ms per batch of {B}")
