"""Kernel-trace target: hipGraph replays of the batch-1 forward (use under rocprofv3 --kernel-trace); prints nothing itself."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smokephysai_amd.models import SmokePhysNet, GraphedSmokePhysNet
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = SmokePhysNet().to(dev).eval()
g = GraphedSmokePhysNet(model)
x = torch.rand(bs, 1, 256, 256, device=dev)
with torch.no_grad():
    for _ in range(3):
        g(x)
    torch.cuda.synchronize()
    for _ in range(20):
        g(x)
torch.cuda.synchronize()
