"""configs[4] encoder (explicit-GEMM first slice): ms per 512 x 512 x 64 volume, with the split conv1 / im2col+GEMM2 / pool times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smokephysai_amd.models import HipEncoder3D
rng = np.random.RandomState(0)
w = {k: torch.from_numpy(v.astype(np.float32)) for k, v in dict(
    conv1_w=rng.randn(64, 1, 7, 7, 7) * 0.05, conv1_b=rng.randn(64) * 0.1, bn1_w=rng.rand(64) + 0.5, bn1_b=rng.randn(64) * 0.1,
    bn1_mean=rng.randn(64) * 0.2, bn1_var=rng.rand(64) + 0.3, conv2_w=rng.randn(128, 64, 3, 3, 3) * 0.03,
    conv2_b=rng.randn(128) * 0.1, bn2_w=rng.rand(128) + 0.5, bn2_b=rng.randn(128) * 0.1,
    bn2_mean=rng.randn(128) * 0.2, bn2_var=rng.rand(128) + 0.3).items()}
D, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 512, 512)
enc = HipEncoder3D(w, a2_bytes=int(os.environ.get('A2_MB', '128')) << 20)
x = torch.rand(1, D, H, W, device="cuda")
enc(x); torch.cuda.synchronize()
t0 = time.perf_counter(); a1 = enc.conv1_activations(x[0]); torch.cuda.synchronize(); t1 = time.perf_counter()
f = enc(x); torch.cuda.synchronize(); t2 = time.perf_counter()
flop = 2.0 * D * H * W * (343 * 64 + 27 * 64 * 128)
print(f"{D}x{H}x{W}: conv1 {1e3 * (t1 - t0):.1f} ms, whole encoder {1e3 * (t2 - t1):.1f} ms per volume = {flop / (t2 - t1) / 1e12:.1f} counted TFLOP/s; finite {bool(torch.isfinite(f).all())}")
