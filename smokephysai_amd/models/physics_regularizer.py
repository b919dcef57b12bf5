"""Physics-motivated loss terms for SmokePhysNet training.

Drop-in for the reference's PhysicsRegularizer (/root/reference/src/models/physics_regularizer.py:5-109): same
constructor weights, same four public term methods, same forward(predictions, targets) -> dict contract (keys
`mass_conservation`, `continuity`, `energy_conservation`, `divergence`, `total_physics_loss`).
These are a handful of scalar reductions under autograd, so they stay PyTorch-ROCm tensor ops.
"""
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

Tensor = torch.Tensor


def _zero_like_device(t: Tensor) -> Tensor:
    return torch.zeros((), device=t.device)


class PhysicsRegularizer(nn.Module):
    def __init__(self, conservation_weight: float = 1.0, continuity_weight: float = 1.0, energy_weight: float = 0.5):
        super().__init__()
        self.conservation_weight = conservation_weight
        self.continuity_weight = continuity_weight
        self.energy_weight = energy_weight

    # ---- individual terms ------------------------------------------------------------------------------
    def mass_conservation_loss(self, density_pred: Tensor, density_target: Tensor) -> Tensor:
        """MSE between the total mass (spatial sum) of prediction and target  (reference :18-24)."""
        total = lambda d: d.sum(dim=(-2, -1))
        return F.mse_loss(total(density_pred), total(density_target))

    def continuity_loss(self, density_sequence: Tensor) -> Tensor:
        """Mean absolute frame-to-frame change of a [B, T, H, W] sequence; 0 for T < 2  (reference :26-35)."""
        if density_sequence.shape[1] < 2:
            return _zero_like_device(density_sequence)
        return density_sequence.diff(dim=1).abs().mean()

    def energy_conservation_loss(self, velocity_pred: Tensor) -> Tensor:
        """Penalise kinetic-energy growth along dim 0  (reference :37-49)."""
        energy = (velocity_pred.square().sum(dim=1)) * 0.5
        if energy.shape[0] <= 1:
            return _zero_like_device(velocity_pred)
        return F.relu(energy.diff(dim=0)).mean()

    def divergence_loss(self, velocity: Tensor) -> Tensor:
        """Mean squared discrete divergence of a 2-component field  (reference :51-71)."""
        if velocity.shape[1] != 2:
            return _zero_like_device(velocity)
        du = velocity[:, 0].diff(dim=2)
        dv = velocity[:, 1].diff(dim=1)
        rows, cols = min(du.shape[1], dv.shape[1]), min(du.shape[2], dv.shape[2])
        return (du[:, :rows, :cols] + dv[:, :rows, :cols]).square().mean()

    # ---- weighted sum ----------------------------------------------------------------------------------
    def forward(self, predictions: Dict[str, Tensor], targets: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
        """Every term whose inputs are present is evaluated and added with its weight (reference :73-109)."""
        terms: Dict[str, tuple] = {}
        if "density" in predictions and targets and "density" in targets:
            terms["mass_conservation"] = (self.conservation_weight,
                                          lambda: self.mass_conservation_loss(predictions["density"], targets["density"]))
        if "density_sequence" in predictions:
            terms["continuity"] = (self.continuity_weight, lambda: self.continuity_loss(predictions["density_sequence"]))
        if "velocity" in predictions:
            terms["energy_conservation"] = (self.energy_weight,
                                            lambda: self.energy_conservation_loss(predictions["velocity"]))
            terms["divergence"] = (0.5, lambda: self.divergence_loss(predictions["velocity"]))
        out: Dict[str, Tensor] = {}
        total = 0.0
        for name, (weight, fn) in terms.items():
            out[name] = fn()
            total = total + weight * out[name]
        out["total_physics_loss"] = total
        return out
