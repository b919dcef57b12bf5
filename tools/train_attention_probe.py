"""Timing of attention forward + backward in training: libsmokehip (smk_attention_forward_lse / smk_attention_backward) against
PyTorch-ROCm's fp32 scaled_dot_product_attention under autograd, at the model's shape (L = 1024, 8 heads of 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from smokephysai_amd.models.attention import hip_attention_train


def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): f()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


for B in (int(a) for a in (sys.argv[1:] or ["64", "4"])):
    L, H, D = 1024, 8, 512
    q, k, v = [torch.randn(B, L, D, device="cuda", requires_grad=True) for _ in range(3)]
    dout = torch.randn(B, L, D, device="cuda")

    def hip_fwd():
        return hip_attention_train(q, k, v, H, 0.125)

    def hip_fb():
        hip_attention_train(q, k, v, H, 0.125).backward(dout)

    def sdpa(q, k, v):
        qq, kk, vv = [t.view(B, L, H, 64).transpose(1, 2) for t in (q, k, v)]
        return F.scaled_dot_product_attention(qq, kk, vv, scale=0.125).transpose(1, 2).contiguous().view(B, L, D)

    def ref_fb():
        sdpa(q, k, v).backward(dout)
    tf, tfb = timeit(hip_fwd), timeit(hip_fb)
    rf, rfb = timeit(lambda: sdpa(q, k, v)), timeit(ref_fb)
    fl = 4.0 * B * H * L * L * 64
    print(f"B={B}: hip fwd {tf:8.1f} us, fwd+bwd {tfb:8.1f} us (bwd {tfb - tf:8.1f} us = {2.5 * fl / (tfb - tf) / 1e6:6.1f} counted TF/s)   "
          f"torch fp32 fwd {rf:8.1f} us, fwd+bwd {rfb:8.1f} us", flush=True)
