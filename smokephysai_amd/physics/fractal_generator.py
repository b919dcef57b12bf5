"""Fractal frame perturbation on MI355X -- drop-in for src/physics/fractal_generator.py:5-62.

The perlin and mandelbrot fields depend only on the grid shape; the reference recomputes them for every frame
(100 masked complex iterations), here they are computed once per shape by a HIP kernel and kept resident.
"""
import torch
import torch.nn as nn

from .. import _lib


class FractalGenerator(nn.Module):
    def __init__(self, device="cuda"):
        super().__init__()
        self.device = device
        self._dev = _lib.require_cuda(device, "FractalGenerator")
        self._L = _lib.load()
        self._cache = {}

    def _constants(self, shape):
        h, w = shape
        if h != w:
            # fractal_generator.py:44,49: a [w,h] mask indexes an [h,w] buffer -> IndexError in the reference
            raise IndexError(f"fractal fields need a square grid (got {h}x{w}); the reference fails the same way")
        if h not in self._cache:
            per, man, fld = (torch.empty(h, h, device=self._dev) for _ in range(3))
            _lib.check(self._L.smk_fractal_constants(h, per.data_ptr(), man.data_ptr(), fld.data_ptr(),
                                                     _lib.stream_ptr(self._dev)))
            self._cache[h] = (per, man, fld)
        return self._cache[h]

    def generate_perlin_noise(self, shape: tuple, scale: float = 10.0) -> torch.Tensor:
        """fractal_generator.py:12-31 (scale is fixed at the reference's only call value, 10.0)."""
        if scale != 10.0:
            raise NotImplementedError("only scale=10.0 (the reference's call site, fractal_generator.py:55)")
        return self._constants(tuple(shape))[0].clone()

    def generate_mandelbrot_field(self, shape: tuple, iterations: int = 100) -> torch.Tensor:
        """fractal_generator.py:33-51."""
        if iterations != 100:
            raise NotImplementedError("only iterations=100 (the reference's call site, fractal_generator.py:56)")
        return self._constants(tuple(shape))[1].clone()

    def apply_fractal_perturbation(self, field: torch.Tensor, intensity: float = 0.1) -> torch.Tensor:
        """fractal_generator.py:53-62: field + intensity * (0.7*perlin + 0.3*mandelbrot) * field."""
        f = torch.as_tensor(field, dtype=torch.float32, device=self._dev).contiguous()
        h, w = f.shape[-2:]
        if h != w:
            raise IndexError(f"fractal fields need a square grid (got {h}x{w}); the reference fails the same way")
        out = torch.empty_like(f)
        _lib.check(self._L.smk_apply_fractal(f.data_ptr(), out.data_ptr(), f.numel() // (h * w), h, float(intensity),
                                             _lib.stream_ptr(self._dev)))
        return out
