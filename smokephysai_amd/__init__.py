"""smokephysai_amd -- MI355X-native drop-in for the SmokePhysAI hot path.

Package layout mirrors the reference's `src/` tree (physics/, models/, utils/) so that a user switches by
replacing `from src.` with `from smokephysai_amd.`; the compute lives in csrc/ (hand-written HIP for gfx950)
behind the C ABI declared in include/smokehip.h.  There is NO CPU fallback: every op here needs
libsmokehip.so and a ROCm device, and fails loudly otherwise.
"""
__version__ = "0.1.0"
