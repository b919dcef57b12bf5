from .navier_stokes import NavierStokesSimulator
from .navier_stokes3d import NavierStokesSimulator3D
from .smoke_simulator import SmokeSimulator
from .fractal_generator import FractalGenerator

__all__ = ["NavierStokesSimulator", "NavierStokesSimulator3D", "SmokeSimulator", "FractalGenerator"]
