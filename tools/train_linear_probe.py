"""Timing of the three GEMMs of a linear layer's training step at the transformer-body shapes (rows = 64 x 1024 tokens):
libsmokehip (forward, dX through the transposed mirror, dW = smk_linear_wgrad) against PyTorch-ROCm's fp32 GEMMs."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smokephysai_amd.models.linear import HipLinear, hip_linear_wgrad


def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): f()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for K, N in [(512, 512), (512, 2048), (2048, 512), (128, 512), (512, 256)]:
    x = torch.randn(rows, K, device="cuda"); w = torch.randn(N, K, device="cuda") / math.sqrt(K); dy = torch.randn(rows, N, device="cuda")
    fwd = HipLinear(w, None); bwd = HipLinear(w.t().contiguous(), None)
    t = {"fwd": (timeit(lambda: fwd(x)), timeit(lambda: x @ w.t())),
         "dX": (timeit(lambda: bwd(dy)), timeit(lambda: dy @ w)),
         "dW": (timeit(lambda: hip_linear_wgrad(dy, x)), timeit(lambda: dy.t() @ x))}
    print(f"rows={rows} in={K:5d} out={N:5d}: " + "   ".join(f"{k} hip {a:7.1f} us / torch {b:7.1f} us" for k, (a, b) in t.items()), flush=True)
