from .navier_stokes import NavierStokesSimulator
from .smoke_simulator import SmokeSimulator
from .fractal_generator import FractalGenerator

__all__ = ["NavierStokesSimulator", "SmokeSimulator", "FractalGenerator"]
