"""Smoke simulation system on MI355X -- drop-in for src/physics/smoke_simulator.py:8-139.

simulate_step = one fused stencil step of the batched solver; the fractal perturbation is applied to the emitted
frame only (the solver keeps the unperturbed density, smoke_simulator.py:36-39) inside the density-advect kernel.
The chaos statistics (smoke_simulator.py:47-140) use the HIP reductions of csrc/chaos.hip (mean, box counts, histogram,
frame-difference norms); only the scalar formulas on their results run on the host, as in the reference.
"""
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
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
        ns.check()                 # all n_steps were enqueued at once: a timed-out projection among them is reported HERE, not a call later
        return out

    # ---- chaos statistics (smoke_simulator.py:47-140) -------------------------------------------------------
    def get_chaos_features(self):
        """Un-batched: the reference's dict (or {} with fewer than 10 frames).  Batched: a list with one dict per grid."""
        if len(self.history) < 10:
            return {} if self.batch_size is None else [{} for _ in range(self.ns_solver._B)]
        self.ns_solver.check()     # the statistics go to the host: the frames behind them must be real
        if self.batch_size is None:
            return {"lyapunov_exponent": self.compute_lyapunov_exponent(),
                    "fractal_dimension": self.compute_fractal_dimension(),
                    "entropy": self.compute_entropy()}
        cur = self.history[-1]                                           # [B,H,W]
        _, box, hist = chaos_stats(cur)
        box, hist = box.cpu().numpy(), hist.cpu().numpy()
        lyap = [0.0] * cur.shape[0]
        if len(self.history) >= 20:
            states = torch.stack(self.history[-20:], dim=1)             # [B,20,H,W]
            # one launch over the flat stream of B x 20 frames and one copy to the host; the pair (last of grid b, first of b+1) is unused
            d = frame_diff_norms(states.view(-1, *states.shape[2:])).cpu().numpy()
            lyap = [lyapunov_from_norms(d[20 * b:20 * b + 19]) for b in range(cur.shape[0])]
        return [{"lyapunov_exponent": lyap[b], "fractal_dimension": fractal_dimension_from_counts(box[b]),
                 "entropy": entropy_from_hist(hist[b])} for b in range(cur.shape[0])]

    def _single_grid(self, what):
        if self.batch_size is not None:
            raise ValueError(f"{what}: batched simulator -- use get_chaos_features(), which returns one dict per grid")

    def compute_lyapunov_exponent(self) -> float:
        self._single_grid("compute_lyapunov_exponent")
        if len(self.history) < 20:
            return 0.0
        return lyapunov_from_frames(torch.stack(self.history[-20:]))

    def compute_fractal_dimension(self) -> float:
        self._single_grid("compute_fractal_dimension")
        if not self.history:
            return 0.0
        return fractal_dimension(self.history[-1])

    def compute_entropy(self) -> float:
        self._single_grid("compute_entropy")
        if not self.history:
            return 0.0
        return histogram_entropy(self.history[-1])


# ---- chaos statistics: HIP reductions (csrc/chaos.hip) + the reference's tiny host-side formulas ---------------
def chaos_stats(frames: torch.Tensor):
    """means [n] fp32, box counts [n,5] int32 (scales 2..32 of frame > mean), histogram [n,256] int32 of n frames
    [n,H,W] on the device (smoke_simulator.py:89-140's reductions, one launch)."""
    dev = _lib.require_cuda(frames.device, "chaos_stats")
    f = frames.to(torch.float32)
    if f.dim() == 2:
        f = f[None]
    if f.stride(2) != 1 or f.stride(1) != f.shape[2]:
        f = f.contiguous()
    n, h, w = f.shape
    means = torch.empty(n, device=dev)
    box = torch.empty(n, 5, dtype=torch.int32, device=dev)
    hist = torch.empty(n, 256, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().smk_chaos_stats(f.data_ptr(), f.stride(0), n, h, w, means.data_ptr(), box.data_ptr(),
                                           hist.data_ptr(), _lib.stream_ptr(dev)))
    return means, box, hist


def frame_diff_norms(frames: torch.Tensor) -> torch.Tensor:
    """||frames[i+1] - frames[i]||_2 for consecutive frames [n,H,W] -> [n-1] fp32 (smoke_simulator.py:73-79)."""
    dev = _lib.require_cuda(frames.device, "frame_diff_norms")
    f = frames.to(torch.float32).contiguous()
    n, h, w = f.shape
    out = torch.empty(n - 1, device=dev)
    _lib.check(_lib.load().smk_frame_diff_norms(f.data_ptr(), f.stride(0), n, h, w, out.data_ptr(), _lib.stream_ptr(dev)))
    return out


def lyapunov_from_norms(distances) -> float:
    """smoke_simulator.py:81-87: mean(diff(log(d + 1e-8))) clamped at 0 (host, float64 like the reference)."""
    d = np.asarray(distances, dtype=np.float64)
    if len(d) > 1:
        return max(0, float(np.mean(np.diff(np.log(d + 1e-8)))))
    return 0.0


def fractal_dimension_from_counts(counts) -> float:
    """smoke_simulator.py:116-122: |slope| of log(count+1) vs log(scale)."""
    slope = np.polyfit(np.log([2, 4, 8, 16, 32]), np.log(np.asarray(counts, dtype=np.float64) + 1), 1)[0]
    return abs(float(slope))


def entropy_from_hist(hist) -> float:
    """smoke_simulator.py:136-140: -sum(p * log2(p + 1e-8)), p = counts / total, fp32 like the reference."""
    h = np.asarray(hist).astype(np.float32)
    probs = h / h.sum(dtype=np.float32)
    return float(-np.sum(probs * np.log2(probs + np.float32(1e-8)), dtype=np.float32))


def lyapunov_from_frames(states: torch.Tensor) -> float:
    return lyapunov_from_norms(frame_diff_norms(states).cpu().numpy())


def fractal_dimension(frame: torch.Tensor) -> float:
    return fractal_dimension_from_counts(chaos_stats(frame)[1][0].cpu().numpy())


def histogram_entropy(frame: torch.Tensor) -> float:
    return entropy_from_hist(chaos_stats(frame)[2][0].cpu().numpy())
