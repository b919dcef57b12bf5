"""Physics regularisation losses -- drop-in for src/models/physics_regularizer.py:5-109.
Tiny reductions with an autograd path: kept as PyTorch-ROCm tensor ops (out of scope for HIP)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class PhysicsRegularizer(nn.Module):
    def __init__(self, conservation_weight: float = 1.0, continuity_weight: float = 1.0, energy_weight: float = 0.5):
        super().__init__()
        self.conservation_weight = conservation_weight
        self.continuity_weight = continuity_weight
        self.energy_weight = energy_weight

    def mass_conservation_loss(self, density_pred, density_target):          # physics_regularizer.py:18-24
        return F.mse_loss(density_pred.sum(dim=(-2, -1)), density_target.sum(dim=(-2, -1)))

    def continuity_loss(self, density_sequence):                              # physics_regularizer.py:26-35
        if density_sequence.shape[1] < 2:
            return torch.tensor(0.0, device=density_sequence.device)
        return torch.mean(torch.abs(density_sequence[:, 1:] - density_sequence[:, :-1]))

    def energy_conservation_loss(self, velocity_pred):                        # physics_regularizer.py:37-49
        kinetic = 0.5 * (velocity_pred ** 2).sum(dim=1)
        if kinetic.shape[0] > 1:
            return torch.relu(kinetic[1:] - kinetic[:-1]).mean()
        return torch.tensor(0.0, device=velocity_pred.device)

    def divergence_loss(self, velocity):                                      # physics_regularizer.py:51-71
        if velocity.shape[1] != 2:
            return torch.tensor(0.0, device=velocity.device)
        u, v = velocity[:, 0], velocity[:, 1]
        du_dx = u[:, :, 1:] - u[:, :, :-1]
        dv_dy = v[:, 1:, :] - v[:, :-1, :]
        mh, mw = min(du_dx.shape[1], dv_dy.shape[1]), min(du_dx.shape[2], dv_dy.shape[2])
        return torch.mean((du_dx[:, :mh, :mw] + dv_dy[:, :mh, :mw]) ** 2)

    def forward(self, predictions: dict, targets: dict = None) -> dict:       # physics_regularizer.py:73-109
        losses = {}
        total = 0.0
        if "density" in predictions and targets and "density" in targets:
            m = self.mass_conservation_loss(predictions["density"], targets["density"])
            losses["mass_conservation"] = m
            total = total + self.conservation_weight * m
        if "density_sequence" in predictions:
            c = self.continuity_loss(predictions["density_sequence"])
            losses["continuity"] = c
            total = total + self.continuity_weight * c
        if "velocity" in predictions:
            e = self.energy_conservation_loss(predictions["velocity"])
            losses["energy_conservation"] = e
            total = total + self.energy_weight * e
            dv = self.divergence_loss(predictions["velocity"])
            losses["divergence"] = dv
            total = total + 0.5 * dv
        losses["total_physics_loss"] = total
        return losses
