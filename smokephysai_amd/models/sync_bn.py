"""Cross-rank BatchNorm2d for the data-parallel training step.

The reference trains in ONE process: its BatchNorm layers (smokephys_net.py:26,29,59,62) see the whole batch.  Under DDP every rank
normalises with its own shard's statistics unless the statistics are exchanged.  `SyncBatchNorm2d` restores the reference's semantics:
per channel the ranks all-reduce (sum, sum of squares, count) in the forward and (sum dy, sum dy * xhat) in the backward -- 2 x C
floats each way (C <= 128 here), tiny beside the 111 MB gradient all-reduce -- so outputs, input gradients, weight / bias gradients and
running statistics equal the single-process full-batch BatchNorm2d (up to fp32 reduction order).

Unlike torch.nn.SyncBatchNorm this module runs on any device and any torch.distributed backend (gloo on CPU for the tests,
RCCL on MI355X) and keeps nn.BatchNorm2d's parameters, buffers and state_dict keys, so a reference checkpoint loads unchanged.
"""
import torch
import torch.distributed as dist
from torch import nn


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class _SyncBatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        C = x.shape[1]
        dims = (0, 2, 3)
        n_local = x.numel() // C
        # local sums in fp64 on [C] vectors (the reduction over B*H*W itself runs in fp32 inside torch's kernels)
        s1 = x.sum(dims, dtype=torch.float32).double()
        s2 = (x * x).sum(dims, dtype=torch.float32).double()
        packed = torch.cat([s1, s2, torch.tensor([float(n_local)], dtype=torch.float64, device=x.device)])
        if _world() > 1:
            dist.all_reduce(packed, group=group)
        n = packed[-1]
        mean = packed[:C] / n
        var = (packed[C:2 * C] / n - mean * mean).clamp_min_(0.0)                 # biased variance of the GLOBAL batch
        rstd = torch.rsqrt(var + eps)
        mean32, rstd32 = mean.float(), rstd.float()
        xhat = (x - mean32[None, :, None, None]) * rstd32[None, :, None, None]
        ctx.save_for_backward(xhat, weight, rstd32)
        ctx.group, ctx.n = group, float(n)
        ctx.mark_non_differentiable(mean32, var)
        return xhat * weight[None, :, None, None] + bias[None, :, None, None], mean32, var.float(), n.float()

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar, _dn):
        xhat, weight, rstd = ctx.saved_tensors
        dims = (0, 2, 3)
        C = xhat.shape[1]
        sum_dy = dy.sum(dims, dtype=torch.float32)
        sum_dy_xhat = (dy * xhat).sum(dims, dtype=torch.float32)
        dweight, dbias = sum_dy_xhat.clone(), sum_dy.clone()                         # LOCAL sums: DDP averages parameter gradients itself
        packed = torch.cat([sum_dy, sum_dy_xhat]).double()
        if _world() > 1:
            dist.all_reduce(packed, group=ctx.group)
        k1 = (packed[:C] / ctx.n).float()
        k2 = (packed[C:] / ctx.n).float()
        dx = (weight * rstd)[None, :, None, None] * (dy - k1[None, :, None, None] - xhat * k2[None, :, None, None])
        return dx, dweight, dbias, None, None


class SyncBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d whose training statistics span every rank of the process group (eval mode: plain running statistics)."""

    process_group = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.training:
            return super().forward(x)
        if x.dim() != 4:
            raise ValueError("SyncBatchNorm2d expects [B, C, H, W]")
        if not self.affine:
            raise ValueError("SyncBatchNorm2d needs affine=True (the reference's BatchNorm layers are affine)")
        y, mean, var, n = _SyncBatchNormFn.apply(x, self.weight, self.bias, self.eps, self.process_group)
        if self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                nn_ = float(n)
                self.running_mean.mul_(1 - m).add_(mean, alpha=m)
                self.running_var.mul_(1 - m).add_(var, alpha=m * nn_ / max(nn_ - 1.0, 1.0))      # unbiased, like nn.BatchNorm2d
        return y


def convert_sync_batchnorm(module: nn.Module, process_group=None) -> nn.Module:
    """Replace every nn.BatchNorm2d below `module` (in place) by a SyncBatchNorm2d sharing its parameters and buffers."""
    for name, child in list(module.named_children()):
        if type(child) is nn.BatchNorm2d:
            sb = SyncBatchNorm2d(child.num_features, eps=child.eps, momentum=child.momentum, affine=child.affine,
                                 track_running_stats=child.track_running_stats)
            if child.affine:
                sb.weight, sb.bias = child.weight, child.bias
            if child.track_running_stats:
                sb.running_mean, sb.running_var, sb.num_batches_tracked = child.running_mean, child.running_var, child.num_batches_tracked
            sb.process_group = process_group
            sb.train(child.training)
            setattr(module, name, sb)
        else:
            convert_sync_batchnorm(child, process_group)
    return module
