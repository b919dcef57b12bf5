"""Kernel-trace target: N eager eval forwards of SmokePhysNet at one batch size (use under rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smokephysai_amd.models import SmokePhysNet
bs = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = SmokePhysNet().to(dev).eval()
x = torch.rand(bs, 1, 256, 256, device=dev)
with torch.no_grad():
    for _ in range(reps):
        model(x)
torch.cuda.synchronize()
