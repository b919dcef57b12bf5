"""Timing of smk_attention (split-bf16 flash attention) against torch SDPA fp32 at the model's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smokephysai_amd.models.attention import hip_attention

def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): f()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps

for B in (1, 4, 64):
    L, H, D = 1024, 8, 512
    q, k, v = (torch.randn(B, L, D, device="cuda") for _ in range(3))
    out = torch.empty_like(q)
    t_hip = timeit(lambda: hip_attention(q, k, v, H, 0.125, out=out))
    fl = 4.0 * B * H * L * L * 64
    if len(sys.argv) > 1 and sys.argv[1] == "hip":          # (tools/att_ablate.sh: the HIP kernel only)
        print(f"B={B:3d}: hip {t_hip*1e3:8.1f} us ({fl/t_hip/1e9:6.1f} TF/s counted)", flush=True)
        continue
    q4, k4, v4 = (t.view(B, L, H, 64).transpose(1, 2) for t in (q, k, v))
    t_ref = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q4, k4, v4, scale=0.125).transpose(1, 2).reshape(B, L, D))
    print(f"B={B:3d}: hip {t_hip*1e3:8.1f} us ({fl/t_hip/1e9:6.1f} TF/s counted)   torch SDPA fp32 + merge-heads copy {t_ref*1e3:8.1f} us ({fl/t_ref/1e9:6.1f} TF/s)", flush=True)
