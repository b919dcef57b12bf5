"""CPU oracle for the SmokePhysAI hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (smokephysai_amd) must never import it.
"""
from .oracle import *  # noqa: F401,F403
