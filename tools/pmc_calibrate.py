#!/usr/bin/env python3
"""PMC calibration kernel: k_apply_fractal (a pure stream: reads N floats once, writes N floats, plus a 256 KiB
L2-resident constant) over a buffer larger than the 256 MiB Infinity Cache, in the stencil kernels' own access
pattern (one dword per lane, row-coalesced).  FETCH_SIZE/WRITE_SIZE of this launch against the known byte count give
the gfx950 correction factors for this access width (MI355X_MICROARCH.md, HBM section)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smokephysai_amd.physics import FractalGenerator

fg = FractalGenerator()
B, R, C = 1536, 256, 256                     # 402,653,184 bytes in, same out
x = torch.rand(B, R, C, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    y = fg.apply_fractal_perturbation(x, 0.05)
torch.cuda.synchronize()
print("calibration bytes per launch (read, write):", x.numel() * 4, x.numel() * 4)
