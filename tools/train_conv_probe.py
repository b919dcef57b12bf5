"""Time the training passes of SmokePhysNet.input_encoder's two convolutions on libsmokehip against PyTorch-ROCm (MIOpen) at
BASELINE configs[3]'s per-GPU shape (batch 64 of 256 x 256): conv1 forward / weight gradient, conv2 forward / data gradient / weight gradient.
usage: train_conv_probe.py [batch] [grid]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smokephysai_amd import _lib                              # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
L = _lib.load()
st = _lib.stream_ptr(dev)
torch.manual_seed(0)
x = torch.rand(B, 1, N, N, device=dev)
c1 = torch.nn.Conv2d(1, 64, 7, padding=3).to(dev)
c2 = torch.nn.Conv2d(64, 128, 3, padding=1).to(dev)
a1 = torch.relu(torch.randn(B, 64, N, N, device=dev))
dz2 = torch.randn(B, 128, N, N, device=dev)
dz1 = torch.randn(B, 64, N, N, device=dev)
z1, z2, dx = torch.empty_like(dz1), torch.empty_like(dz2), torch.empty_like(a1)
dw1, db1, dw2, db2 = torch.empty_like(c1.weight), torch.empty(64, device=dev), torch.empty_like(c2.weight), torch.empty(128, device=dev)
ws2 = torch.empty(int(L.smk_conv2_train_workspace()), device=dev, dtype=torch.uint8)
wsw = torch.empty(int(L.smk_conv2_train_wgrad_workspace()), device=dev, dtype=torch.uint8)
ws1 = torch.empty(int(L.smk_conv1_train_wgrad_workspace()), device=dev, dtype=torch.uint8)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


w1, b1, w2, b2 = c1.weight.detach(), c1.bias.detach(), c2.weight.detach(), c2.bias.detach()
rows = [
    ("conv1 forward", lambda: _lib.check(L.smk_conv1_train_forward(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), B, N, N, z1.data_ptr(), st)),
     lambda: torch.ops.aten.convolution(x, w1, b1, [1, 1], [3, 3], [1, 1], False, [0, 0], 1)),
    ("conv1 wgrad", lambda: _lib.check(L.smk_conv1_train_wgrad(dz1.data_ptr(), x.data_ptr(), B, N, N, dw1.data_ptr(), db1.data_ptr(), ws1.data_ptr(), st)),
     lambda: torch.ops.aten.convolution_backward(dz1, x, w1, [64], [1, 1], [3, 3], [1, 1], False, [0, 0], 1, [False, True, True])),
    ("conv2 forward", lambda: _lib.check(L.smk_conv2_train_forward(a1.data_ptr(), w2.data_ptr(), b2.data_ptr(), B, N, N, z2.data_ptr(), ws2.data_ptr(), st)),
     lambda: torch.ops.aten.convolution(a1, w2, b2, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)),
    ("conv2 dgrad", lambda: _lib.check(L.smk_conv2_train_dgrad(dz2.data_ptr(), w2.data_ptr(), B, N, N, dx.data_ptr(), ws2.data_ptr(), st)),
     lambda: torch.ops.aten.convolution_backward(dz2, a1, w2, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])),
    ("conv2 wgrad", lambda: _lib.check(L.smk_conv2_train_wgrad(dz2.data_ptr(), a1.data_ptr(), B, N, N, dw2.data_ptr(), db2.data_ptr(), wsw.data_ptr(), st)),
     lambda: torch.ops.aten.convolution_backward(dz2, a1, w2, [128], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, True])),
]
for name, hip, ref in rows:
    print("%-14s libsmokehip %7.3f ms   PyTorch-ROCm %7.3f ms" % (name, timed(hip), timed(ref)), flush=True)
