"""Smoke simulation system on MI355X -- drop-in for src/physics/smoke_simulator.py:8-139.

simulate_step = one fused stencil step of the batched solver; the fractal perturbation is applied to the emitted
frame only (the solver keeps the unperturbed density, smoke_simulator.py:36-39) inside the density-advect kernel.
The chaos statistics (smoke_simulator.py:47-140) are computed on the device with torch reductions.
"""
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .fractal_generator import FractalGenerator
from .navier_stokes import NavierStokesSimulator


class SmokeSimulator(nn.Module):
    def __init__(self, grid_size: tuple = (128, 128), dt: float = 0.01, viscosity: float = 0.001,
                 device: str = "cuda", batch_size: Optional[int] = None, jacobi_iters: int = 20):
        super().__init__()
        self.ns_solver = NavierStokesSimulator(grid_size, dt, viscosity, device, batch_size=batch_size,
                                               jacobi_iters=jacobi_iters)
        self.fractal_gen = FractalGenerator(device)
        self.device = device
        self.batch_size = batch_size
        self.history = []          # smoke_simulator.py:22-24
        self.max_history = 100

    def add_incense_source(self, positions: list, intensities: list, grid: Optional[int] = None):
        """smoke_simulator.py:26-29 (radius 8).  Batched: `grid` selects the grid, None = every grid."""
        ns = self.ns_solver
        grids = range(ns._B) if grid is None else [grid]
        ns.add_smoke_sources([(g, x, y, 8, inten) for g in grids for (x, y), inten in zip(positions, intensities)])

    def simulate_step(self, add_fractal: bool = True) -> torch.Tensor:
        """smoke_simulator.py:31-45."""
        ns = self.ns_solver
        frame = torch.empty(ns._B, ns.h, ns.w, device=ns._dev)
        ns.step_into(frame, 1, add_fractal=add_fractal, fractal_intensity=0.05)
        density = frame if self.batch_size is not None else frame[0]
        self.history.append(density.clone())
        if len(self.history) > self.max_history:
            self.history.pop(0)
        return density

    def simulate_sequence(self, n_steps: int, add_fractal: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """n_steps frames per grid straight into one [B, n_steps, H, W] tensor (no history bookkeeping)."""
        ns = self.ns_solver
        if out is None:
            out = torch.empty(ns._B, n_steps, ns.h, ns.w, device=ns._dev)
        ns.step_into(out, n_steps, add_fractal=add_fractal, fractal_intensity=0.05)
        return out

    # ---- chaos statistics (smoke_simulator.py:47-140) -------------------------------------------------------
    def get_chaos_features(self) -> dict:
        if len(self.history) < 10:
            return {}
        return {"lyapunov_exponent": self.compute_lyapunov_exponent(),
                "fractal_dimension": self.compute_fractal_dimension(),
                "entropy": self.compute_entropy()}

    def compute_lyapunov_exponent(self) -> float:
        if len(self.history) < 20:
            return 0.0
        return lyapunov_from_frames(torch.stack(self.history[-20:]))

    def compute_fractal_dimension(self) -> float:
        if not self.history:
            return 0.0
        return fractal_dimension(self.history[-1])

    def compute_entropy(self) -> float:
        if not self.history:
            return 0.0
        return histogram_entropy(self.history[-1])


# ---- statistics on device tensors (single grid [.., H, W]) -------------------------------------------------
def lyapunov_from_frames(states: torch.Tensor) -> float:
    """smoke_simulator.py:67-87: mean of diff(log(|s[i+1]-s[i]|_2 + 1e-8)) over 20 frames, clamped at 0."""
    d = torch.linalg.vector_norm((states[1:] - states[:-1]).flatten(1), dim=1)       # fp32 norms, as torch.norm
    distances = d.double().cpu().numpy()
    if len(distances) > 1:
        return max(0, float(np.mean(np.diff(np.log(distances + 1e-8)))))
    return 0.0


def box_counts(frame: torch.Tensor) -> torch.Tensor:
    """smoke_simulator.py:96-115: number of scale x scale boxes holding any cell above the frame mean."""
    binary = frame > frame.mean()
    h, w = binary.shape
    counts = []
    for scale in (2, 4, 8, 16, 32):
        bh, bw = h // scale, w // scale
        boxes = binary[: bh * scale, : bw * scale].reshape(bh, scale, bw, scale)
        counts.append(boxes.any(dim=3).any(dim=1).sum())
    return torch.stack(counts)


def fractal_dimension(frame: torch.Tensor) -> float:
    counts = box_counts(frame).cpu().numpy()
    slope = np.polyfit(np.log([2, 4, 8, 16, 32]), np.log(counts + 1), 1)[0]
    return abs(float(slope))


def hist256(frame: torch.Tensor) -> torch.Tensor:
    """torch.histogram(bins=256, range=(0,1)) semantics (smoke_simulator.py:134-135): values outside [0,1] are
    dropped and exactly 1.0 falls in the last bin; bin = floor(x*256) (exact: power-of-two scale)."""
    x = frame.flatten()
    x = x[(x >= 0) & (x <= 1)]
    idx = torch.clamp((x * 256).floor().long(), max=255)
    return torch.bincount(idx, minlength=256)


def histogram_entropy(frame: torch.Tensor) -> float:
    hist = hist256(frame).float()
    probs = hist / hist.sum()
    return float(-torch.sum(probs * torch.log2(probs + 1e-8)))
