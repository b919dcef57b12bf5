"""Batched 3-D stable-fluids stepper on MI355X (BASELINE configs[4]: 512 x 512 x 64 grids, batch 8).

The reference is 2-D only (src/physics/navier_stokes.py:10,21); this class mirrors NavierStokesSimulator's surface (navier_stokes.py:6-173:
constructor arguments, `u v p density` attributes, setup_grid / add_smoke_source / step) for grids `(D, H, W)`, with the third velocity
component `w`.  Semantics: SPEC_3D.md, the rule-by-rule generalisation of the 2-D code; kernels: csrc/stencil3d.hip through the C ABI
(include/smokehip.h, smk_sim3d_*) on torch's current stream.  No CPU fallback.
"""
import ctypes as C
from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .. import _lib

STAGE3D_BUOY_DIFFUSE, STAGE3D_PROJECT, STAGE3D_ADVECT_U, STAGE3D_ADVECT_V, STAGE3D_ADVECT_W, STAGE3D_ADVECT_D = range(6)


def _round_up(x, m):
    return (x + m - 1) // m * m


class NavierStokesSimulator3D(nn.Module):
    def __init__(self, grid_size: Tuple[int, int, int] = (64, 128, 128), dt: float = 0.01, viscosity: float = 0.001,
                 device: str = "cuda", batch_size: Optional[int] = None, jacobi_iters: int = 20):
        super().__init__()
        self.grid_size = tuple(grid_size)
        if len(self.grid_size) != 3:
            raise ValueError("grid_size must be (D, H, W)")
        self.dt, self.viscosity, self.device = dt, viscosity, device
        self.d, self.h, self.w_ = self.grid_size                      # (depth, height, width); `w` is the third velocity component
        self.batch_size = batch_size
        self.jacobi_iters = jacobi_iters
        self._dev = _lib.require_cuda(device, "NavierStokesSimulator3D")
        self._L = _lib.load()
        self._B = 1 if batch_size is None else int(batch_size)
        D, H, W = self.grid_size
        self._pc, self._pv = _round_up(W, 32), _round_up(W + 1, 32)   # rows padded to 128-byte multiples
        B = self._B
        self._u = torch.zeros(B, D, H + 1, self._pc, device=self._dev)
        self._v = torch.zeros(B, D, H, self._pv, device=self._dev)
        self._w = torch.zeros(B, D + 1, H, self._pc, device=self._dev)
        self._p = torch.zeros(B, D, H, self._pc, device=self._dev)
        self._density = torch.zeros(B, D, H, self._pc, device=self._dev)
        desc = _lib.SmkSim3dDesc(B, D, H, W, jacobi_iters, float(dt), float(viscosity), self._dev.index, self._pc, self._pv,
                                 self._u.data_ptr(), self._v.data_ptr(), self._w.data_ptr(), self._p.data_ptr(), self._density.data_ptr())
        handle = C.c_void_p()
        with torch.cuda.device(self._dev):
            _lib.check(self._L.smk_sim3d_create(C.byref(desc), C.byref(handle)))
        self._handle = handle

    def __del__(self):
        h = self.__dict__.get("_handle")
        if h:
            self.__dict__["_handle"] = None
            self._L.smk_sim3d_destroy(h)

    # ---- state views with the SPEC's shapes ----------------------------------------------------------------
    def _view(self, store, cols):
        v = store[..., :cols]
        return v if self.batch_size is not None else v[0]

    def _assign(self, store, cols, value):
        self._view(store, cols).copy_(torch.as_tensor(value, dtype=torch.float32, device=self._dev))

    u = property(lambda s: s._view(s._u, s.w_), lambda s, x: s._assign(s._u, s.w_, x))
    v = property(lambda s: s._view(s._v, s.w_ + 1), lambda s, x: s._assign(s._v, s.w_ + 1, x))
    w = property(lambda s: s._view(s._w, s.w_), lambda s, x: s._assign(s._w, s.w_, x))
    p = property(lambda s: s._view(s._p, s.w_), lambda s, x: s._assign(s._p, s.w_, x))
    density = property(lambda s: s._view(s._density, s.w_), lambda s, x: s._assign(s._density, s.w_, x))

    def _st(self):
        return _lib.stream_ptr(self._dev)

    # ---- NavierStokesSimulator's API in 3-D ------------------------------------------------------------------
    def setup_grid(self, grids: Optional[Sequence[int]] = None):
        """navier_stokes.py:24-35 -- zero the state (also the reset).  `grids`: subset of batch indices."""
        mask = None
        if grids is not None:
            m = bytearray(self._B)
            for g in grids:
                m[g] = 1
            mask = bytes(m)
        _lib.check(self._L.smk_sim3d_reset(self._handle, mask, self._st()))

    def add_smoke_source(self, x: int, y: int, z: int, radius: int = 10, intensity: float = 1.0, grid: Optional[int] = None):
        """navier_stokes.py:37-48 with the depth index appended: (x, y, z) = (column, row, plane)."""
        grids = range(self._B) if grid is None else [grid]
        self.add_smoke_sources([(g, x, y, z, radius, intensity) for g in grids])

    def add_smoke_sources(self, sources):
        """Many sources in one launch: iterable of (grid, x, y, z, radius, intensity), applied in order per grid."""
        sources = list(sources)
        arr = (_lib.SmkSource3d * len(sources))(*[_lib.SmkSource3d(int(g), int(x), int(y), int(z), int(r), float(i))
                                                  for g, x, y, z, r, i in sources])
        _lib.check(self._L.smk_sim3d_add_sources(self._handle, arr, len(sources), self._st()))

    def run_stage(self, stage: int):
        """One stage of step() (include/smokehip.h smk_stage3d) -- parity-test hook."""
        _lib.check(self._L.smk_sim3d_run_stage(self._handle, stage, self._st()))

    def pressure_projection(self):
        """navier_stokes.py:133-149 in 3-D (in place on u, v, w, p)."""
        self.run_stage(STAGE3D_PROJECT)

    def step(self) -> torch.Tensor:
        """navier_stokes.py:151-173 in 3-D -- one time step; returns a copy of the density."""
        D, H, W = self.grid_size
        out = torch.empty(self._B, D, H, W, device=self._dev)
        self.step_into(out)
        return out if self.batch_size is not None else out[0]

    def step_into(self, frames: Optional[torch.Tensor], n_steps: int = 1):
        """n_steps time steps; frame t of grid b goes to frames[b, t] (frames: [B, n_steps, D, H, W] or [B, D, H, W]; None = no frames)."""
        D, H, W = self.grid_size
        ptr, sb, st_ = None, 0, 0
        if frames is not None:
            if frames.dtype != torch.float32 or frames.device != self._dev:
                raise ValueError("frames must be a float32 tensor on the simulator's device")
            fr = frames[:, None] if frames.dim() == 4 else frames
            if frames.dim() == 4 and n_steps != 1:
                raise ValueError("a 4-D frames buffer holds one step")
            if tuple(fr.shape) != (self._B, n_steps, D, H, W) or not fr[0, 0].is_contiguous():
                raise ValueError(f"frames must be [B={self._B}, n_steps={n_steps}, {D}, {H}, {W}] with dense grids")
            ptr, sb, st_ = fr.data_ptr(), fr.stride(0), fr.stride(1)
        _lib.check(self._L.smk_sim3d_step(self._handle, n_steps, ptr, sb, st_, self._st()))

    def forward(self):
        return self.step()
