"""Timing of smk_linear_forward (split-bf16 MFMA) against torch's fp32 GEMM at the transformer-body shapes."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smokephysai_amd.models.linear import HipLinear

def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): f()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps

shapes = [(65536, 128, 512, None), (65536, 512, 512, None), (65536, 512, 2048, "gelu"), (65536, 2048, 512, None),
          (65536, 512, 256, None), (4096, 512, 512, None), (4096, 512, 2048, "gelu"), (4096, 2048, 512, None),
          (1024, 512, 512, None), (1024, 512, 2048, "gelu"), (1024, 2048, 512, None)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[0] == int(sys.argv[1])]
for M, K, N, act in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / math.sqrt(K); b = torch.randn(N, device="cuda")
    lin = HipLinear(w, b)
    y = torch.empty(M, N, device="cuda")
    t_hip = timeit(lambda: lin(x, activation=act, out=y))
    if act:
        t_ref = timeit(lambda: torch.nn.functional.gelu(torch.nn.functional.linear(x, w, b)))
    else:
        t_ref = timeit(lambda: torch.nn.functional.linear(x, w, b))
    fl = 2.0 * M * K * N
    print(f"M={M:6d} K={K:5d} N={N:5d} act={act}: hip {t_hip*1e3:8.1f} us ({fl/t_hip/1e9:7.1f} TF/s counted)   torch fp32 {t_ref*1e3:8.1f} us ({fl/t_ref/1e9:6.1f} TF/s)", flush=True)

# pre-split (SMK_FMT_SPLIT_BF16) activations: the same layers with the split done by the producer
from smokephysai_amd.models.linear import to_split
print("-- split-bf16 input / output")
for M, K, N, act in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / math.sqrt(K); b = torch.randn(N, device="cuda")
    lin = HipLinear(w, b); xs = to_split(x)
    ys = lin(xs, activation=act, x_split=True, out_split=True)
    y = torch.empty(M, N, device="cuda")
    t_in = timeit(lambda: lin(xs, activation=act, x_split=True, out=y))
    t_io = timeit(lambda: lin(xs, activation=act, x_split=True, out_split=True, out=ys))
    fl = 2.0 * M * K * N
    print(f"M={M:6d} K={K:5d} N={N:5d} act={act}: split-in {t_in*1e3:8.1f} us ({fl/t_in/1e9:6.1f} TF/s)   split-in+out {t_io*1e3:8.1f} us ({fl/t_io/1e9:6.1f} TF/s)", flush=True)
