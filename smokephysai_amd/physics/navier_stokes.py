"""Batched stable-fluids stepper on MI355X -- drop-in for src/physics/navier_stokes.py:6-173.

Same constructor, attributes (`u v p density h w grid_size dt viscosity device`) and methods as the reference's
NavierStokesSimulator; the state lives in torch tensors on the GPU and every method enqueues hand-written HIP
kernels (smokephysai_amd/csrc/stencil.hip) on torch's current stream through the C ABI.

Extensions over the reference (all optional, defaults reproduce the reference):
  batch_size=None   -> un-batched, fields have the reference's 2-D shapes; an int B adds a leading batch
                       dimension of independent grids (the reference loops B times in Python instead).
  jacobi_iters=20   -> reference hard-codes 20 (navier_stokes.py:139).
"""
import ctypes as C
from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .. import _lib


def _round_up(x, m):
    return (x + m - 1) // m * m


class NavierStokesSimulator(nn.Module):
    def __init__(self, grid_size: Tuple[int, int] = (128, 128), dt: float = 0.01, viscosity: float = 0.001,
                 device: str = "cuda", batch_size: Optional[int] = None, jacobi_iters: int = 20):
        super().__init__()
        self.grid_size = tuple(grid_size)
        self.dt = dt
        self.viscosity = viscosity
        self.device = device
        self.h, self.w = self.grid_size
        self.batch_size = batch_size
        self.jacobi_iters = jacobi_iters
        self._dev = _lib.require_cuda(device, "NavierStokesSimulator")
        self._L = _lib.load()
        self._B = 1 if batch_size is None else int(batch_size)
        # HBM layout: batch-major planes, rows padded to 128-byte multiples (v has W+1 columns)
        self._pc = _round_up(self.w, 32)
        self._pv = _round_up(self.w + 1, 32)
        B, h = self._B, self.h
        self._u = torch.zeros(B, h + 1, self._pc, device=self._dev)
        self._v = torch.zeros(B, h, self._pv, device=self._dev)
        self._p = torch.zeros(B, h, self._pc, device=self._dev)
        self._density = torch.zeros(B, h, self._pc, device=self._dev)
        self.boundary = torch.zeros((B, h, self.w) if batch_size is not None else (h, self.w), device=self._dev)
        desc = _lib.SmkSimDesc(B, h, self.w, jacobi_iters, float(dt), float(viscosity), self._dev.index,
                               self._pc, self._pv, self._u.data_ptr(), self._v.data_ptr(), self._p.data_ptr(),
                               self._density.data_ptr())
        handle = C.c_void_p()
        with torch.cuda.device(self._dev):
            _lib.check(self._L.smk_sim_create(C.byref(desc), C.byref(handle)))
        self._handle = handle

    def __del__(self):
        h = self.__dict__.get("_handle")
        if h:
            self.__dict__["_handle"] = None       # not nn.Module.__setattr__: torch's globals may be gone at interpreter exit
            if self._L.smk_sim_destroy(h) != 0:   # (an exception raised in __del__ is only printed: warn in words)
                import warnings
                warnings.warn("NavierStokesSimulator: " + self._L.smk_last_error().decode(), RuntimeWarning)

    def close(self):
        """Free the library-side scratch now; raises if a persistent projection of this simulator timed out unreported."""
        h = self.__dict__.get("_handle")
        if h:
            self.__dict__["_handle"] = None
            _lib.check(self._L.smk_sim_destroy(h))

    def check(self):
        """Wait for the work enqueued on the current stream, then raise if a persistent projection among it ran into its bounded
        wait (the affected grids were set to NaN by the kernel).  The reference is synchronous -- results are right or an exception
        is raised (navier_stokes.py:133-149); this is the point where the asynchronous path gives the same guarantee.  Called by
        every method that hands finished frames on (simulate_sequence, the dataset generator, the chaos statistics); not inside a
        stream capture."""
        if torch.cuda.is_current_stream_capturing():
            return
        torch.cuda.current_stream(self._dev).synchronize()
        _lib.check(self._L.smk_sim_status(self._handle))

    # ---- state views with the reference's shapes (navier_stokes.py:27-32) ---------------------------------
    def _view(self, store, cols):
        v = store[:, :, :cols]
        return v if self.batch_size is not None else v[0]

    def _assign(self, store, cols, value):
        self._view(store, cols).copy_(torch.as_tensor(value, dtype=torch.float32, device=self._dev))

    u = property(lambda s: s._view(s._u, s.w), lambda s, x: s._assign(s._u, s.w, x))
    v = property(lambda s: s._view(s._v, s.w + 1), lambda s, x: s._assign(s._v, s.w + 1, x))
    p = property(lambda s: s._view(s._p, s.w), lambda s, x: s._assign(s._p, s.w, x))
    density = property(lambda s: s._view(s._density, s.w), lambda s, x: s._assign(s._density, s.w, x))

    def _st(self):
        return _lib.stream_ptr(self._dev)

    # ---- reference API ---------------------------------------------------------------------------------
    def setup_grid(self, grids: Optional[Sequence[int]] = None):
        """navier_stokes.py:24-35 -- zero the state (also the reset).  `grids`: subset of batch indices."""
        mask = None
        if grids is not None:
            m = bytearray(self._B)
            for g in grids:
                m[g] = 1
            mask = bytes(m)
        _lib.check(self._L.smk_sim_reset(self._handle, mask, self._st()))
        self.boundary.zero_()

    def add_smoke_source(self, x: int, y: int, radius: int = 10, intensity: float = 1.0, grid: Optional[int] = None):
        """navier_stokes.py:37-48.  (x, y) = (column, row).  Batched: `grid` picks the grid, None = every grid."""
        grids = range(self._B) if grid is None else [grid]
        self.add_smoke_sources([(g, x, y, radius, intensity) for g in grids])

    def add_smoke_sources(self, sources):
        """Many sources in one launch: iterable of (grid, x, y, radius, intensity), applied in order per grid."""
        sources = list(sources)
        arr = (_lib.SmkSource * len(sources))(*[_lib.SmkSource(int(g), int(x), int(y), int(r), float(i))
                                                for g, x, y, r, i in sources])
        _lib.check(self._L.smk_sim_add_sources(self._handle, arr, len(sources), self._st()))

    def diffusion_step(self, field: torch.Tensor, viscosity: float) -> torch.Tensor:
        """navier_stokes.py:50-72 -- pure function of `field` ([R,C] or [B,R,C])."""
        f = torch.as_tensor(field, dtype=torch.float32, device=self._dev).contiguous()
        f3 = f if f.dim() == 3 else f[None]
        out = torch.empty_like(f3)
        _lib.check(self._L.smk_diffuse(f3.data_ptr(), out.data_ptr(), f3.shape[0], f3.shape[1], f3.shape[2],
                                       f3.shape[2], float(self.dt), float(viscosity), self._st()))
        return out if f.dim() == 3 else out[0]

    def advection_step(self, field: torch.Tensor, u: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        """navier_stokes.py:74-95 -- pure function; `field` must have the shape of u, v or density."""
        batched = field.dim() == 3
        f, uu, vv = [torch.as_tensor(t, dtype=torch.float32, device=self._dev) for t in (field, u, v)]
        if not batched:
            f, uu, vv = f[None], uu[None], vv[None]
        h, w = self.h, self.w
        shapes = {(h + 1, w): 0, (h, w + 1): 1, (h, w): 2}
        which = shapes.get(tuple(f.shape[1:]))
        if which is None or tuple(uu.shape[1:]) != (h + 1, w) or tuple(vv.shape[1:]) != (h, w + 1):
            raise ValueError("advection_step: field must be u-, v- or density-shaped for this grid")
        f, uu, vv = f.contiguous(), uu.contiguous(), vv.contiguous()
        out = torch.empty_like(f)
        # contiguous reference shapes: pitch_c = W, pitch_v = W+1
        _lib.check(self._L.smk_advect(f.data_ptr(), out.data_ptr(), which, uu.data_ptr(), vv.data_ptr(), f.shape[0],
                                      h, w, w, w + 1, float(self.dt), self._st()))
        return out if batched else out[0]

    # ---- the three public interpolation helpers (navier_stokes.py:97-131) as pure gathers ---------------------------------
    def _interp(self, mode: int, field: torch.Tensor, y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        f = torch.as_tensor(field, dtype=torch.float32, device=self._dev)
        yy = torch.as_tensor(y, dtype=torch.float32, device=self._dev)
        xx = torch.as_tensor(x, dtype=torch.float32, device=self._dev)
        if f.dim() not in (2, 3):
            raise ValueError("field must be [h, w] or a batch [B, h, w]")
        if yy.shape != xx.shape:
            yy, xx = torch.broadcast_tensors(yy, xx)
        batched = f.dim() == 3
        f3 = (f if batched else f[None]).contiguous()
        B, h, w = f3.shape
        # batched field [B, h, w]: coordinates of 3+ dimensions with the same leading B are one list per field; anything else
        # (the reference's 2-D meshgrids) is one list shared by all fields
        shared = not (batched and yy.dim() >= 3 and yy.shape[0] == B)
        yc, xc = yy.contiguous(), xx.contiguous()
        n = yc.numel() if shared else yc[0].numel()
        out = torch.empty((B,) + (tuple(yc.shape) if shared else tuple(yc.shape[1:])), device=self._dev, dtype=torch.float32)
        _lib.check(self._L.smk_interpolate(mode, f3.data_ptr(), B, h, w, w, h * w, yc.data_ptr(), xc.data_ptr(),
                                           0 if shared else n, n, out.data_ptr(), self._st()))
        return out if batched else out[0]

    def bilinear_interpolate(self, field: torch.Tensor, y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """navier_stokes.py:111-131 -- indices are clamped BEFORE the weights are formed, so a coordinate exactly on the upper
        edge (x == w-1 or y == h-1) returns 0, not the edge value (the reference's quirk, reproduced)."""
        return self._interp(0, field, y, x)

    def interpolate_velocity_u(self, u: torch.Tensor, y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """navier_stokes.py:97-102 -- u sampled at (y, clamp(x + 0.5, 0, u.shape[-1] - 1))."""
        return self._interp(1, u, y, x)

    def interpolate_velocity_v(self, v: torch.Tensor, y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """navier_stokes.py:104-109 -- v sampled at (clamp(y + 0.5, 0, v.shape[-2] - 1), x)."""
        return self._interp(2, v, y, x)

    def pressure_projection(self):
        """navier_stokes.py:133-149 (in place on u, v, p)."""
        _lib.check(self._L.smk_sim_run_stage(self._handle, _lib.STAGE_PROJECT, self._st()))

    def run_stage(self, stage: int):
        """One stage of step() (see include/smokehip.h smk_stage) -- parity-test hook."""
        _lib.check(self._L.smk_sim_run_stage(self._handle, stage, self._st()))

    def divergence(self) -> torch.Tensor:
        """navier_stokes.py:136."""
        out = torch.empty(self._B, self.h, self.w, device=self._dev)
        _lib.check(self._L.smk_sim_divergence(self._handle, out.data_ptr(), self._st()))
        return out if self.batch_size is not None else out[0]

    def backtrace_indices(self, which: int):
        """x0, y0 (int32) of advection_step's final gather for field u (0), v (1), density (2)."""
        R = self.h + 1 if which == 0 else self.h
        Cc = self.w + 1 if which == 1 else self.w
        x0 = torch.empty(self._B, R, Cc, dtype=torch.int32, device=self._dev)
        y0 = torch.empty_like(x0)
        _lib.check(self._L.smk_sim_backtrace(self._handle, which, x0.data_ptr(), y0.data_ptr(), self._st()))
        return (x0, y0) if self.batch_size is not None else (x0[0], y0[0])

    def jacobi_plan(self) -> dict:
        """How libsmokehip launches this simulator's pressure projection (kernel, bands per grid, launches, sweeps per launch)."""
        import json
        buf = C.create_string_buffer(4096)
        _lib.check(self._L.smk_sim_describe(self._handle, buf, 4096))
        return json.loads(buf.value.decode())

    def step(self) -> torch.Tensor:
        """navier_stokes.py:151-173 -- one time step; returns a copy of the density."""
        out = torch.empty(self._B, self.h, self.w, device=self._dev)
        self.step_into(out)
        return out if self.batch_size is not None else out[0]

    def step_into(self, frames: Optional[torch.Tensor], n_steps: int = 1, add_fractal: bool = False,
                  fractal_intensity: float = 0.05):
        """n_steps time steps; frame t of grid b goes to frames[b, t] (frames: [B, n_steps, H, W] or [B, H, W])."""
        ptr, sb, st_ = None, 0, 0
        if frames is not None:
            if frames.dtype != torch.float32 or frames.device != self._dev:
                raise ValueError("frames must be a float32 tensor on the simulator's device")
            if frames.dim() == 3:
                if n_steps != 1:
                    raise ValueError("3-D frames buffer holds one step")
                fr = frames[:, None]
            else:
                fr = frames
            if tuple(fr.shape) != (self._B, n_steps, self.h, self.w) or fr.stride(3) != 1 or fr.stride(2) != self.w:
                raise ValueError(f"frames must be [B={self._B}, n_steps={n_steps}, {self.h}, {self.w}] with dense rows")
            ptr, sb, st_ = fr.data_ptr(), fr.stride(0), fr.stride(1)
        _lib.check(self._L.smk_sim_step(self._handle, n_steps, ptr, sb, st_, int(add_fractal), float(fractal_intensity),
                                        self._st()))

    def forward(self):
        return self.step()
