import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smokephysai_amd.models import SmokePhysNet, GraphedSmokePhysNet
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = SmokePhysNet().to(dev).eval()
g = GraphedSmokePhysNet(model)
x64 = torch.rand(64, 1, 256, 256, device=dev)
with torch.no_grad():
    for bs in (8, 16, 32, 64):
        x = x64[:bs].contiguous()
        for _ in range(3): g(x)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps): g(x)
        torch.cuda.synchronize()
        print(f"batch {bs}: {(time.perf_counter() - t0) / reps / bs * 1e3:.4f} ms/frame", flush=True)
