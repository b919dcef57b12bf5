"""Host side of libsmokehip's training-mode BatchNorm2d + ReLU + mean-pool kernels (smk_bn_relu_pool_*): the norm / activation /
pool blocks of SmokePhysNet.input_encoder under autograd (smokephys_net.py:24-32,87-91)."""
import torch
from torch import nn

from .. import _lib


def hip_bn_relu_pool_supported(z: torch.Tensor, pool: int) -> bool:
    if z.dim() != 4 or not z.is_cuda or z.dtype != torch.float32:
        return False
    H, W = z.shape[-2:]
    chunk = 16384 if pool == 8 else 4096
    return pool in (1, 4, 8) and (H * W) % chunk == 0 and (pool == 1 or (W == 32 * pool and H % pool == 0))


class _HipBnReluPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, weight, bias, eps, pool):
        dev = _lib.require_cuda(z.device, "hip_bn_relu_pool")
        L = _lib.load()
        z = z.contiguous()
        B, C, H, W = z.shape
        out = torch.empty(B, C, H // pool, W // pool, device=dev, dtype=torch.float32)
        stats = torch.empty(3, C, device=dev, dtype=torch.float32)          # mean | biased var | rstd
        ws = torch.empty(int(L.smk_bn_train_workspace(B, C, H, W, pool)), device=dev, dtype=torch.uint8)
        w, b = weight.detach().contiguous(), bias.detach().contiguous()
        _lib.check(L.smk_bn_relu_pool_forward(z.data_ptr(), B, C, H, W, w.data_ptr(), b.data_ptr(), float(eps), pool, out.data_ptr(),
                                              stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), ws.data_ptr(),
                                              _lib.stream_ptr(dev)))
        ctx.save_for_backward(z, w, b, stats)
        ctx.pool = pool
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, dout, _dstats):
        z, w, b, stats = ctx.saved_tensors
        L = _lib.load()
        dev = z.device
        B, C, H, W = z.shape
        dout = dout.contiguous()
        dz = torch.empty_like(z)
        dwb = torch.empty(2, C, device=dev, dtype=torch.float32)
        ws = torch.empty(int(L.smk_bn_train_workspace(B, C, H, W, ctx.pool)), device=dev, dtype=torch.uint8)
        _lib.check(L.smk_bn_relu_pool_backward(z.data_ptr(), dout.data_ptr(), B, C, H, W, w.data_ptr(), b.data_ptr(),
                                               stats[0].data_ptr(), stats[2].data_ptr(), ctx.pool, dz.data_ptr(), dwb[0].data_ptr(),
                                               dwb[1].data_ptr(), ws.data_ptr(), _lib.stream_ptr(dev)))
        return dz, dwb[0], dwb[1], None, None


def hip_bn_relu_pool(z: torch.Tensor, bn: nn.BatchNorm2d, pool: int = 1) -> torch.Tensor:
    """blockmean_pool(relu(bn(z))) with bn in training mode (batch statistics, running statistics updated exactly like
    nn.BatchNorm2d: momentum, unbiased variance, num_batches_tracked); z [B, C, H, W] float32 on a ROCm device."""
    if not (bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None):
        raise ValueError("hip_bn_relu_pool: a training-mode affine BatchNorm2d with running statistics and a fixed momentum")
    out, stats = _HipBnReluPoolFn.apply(z, bn.weight, bn.bias, bn.eps, pool)
    with torch.no_grad():
        n = z.numel() // z.shape[1]
        m = bn.momentum
        bn.running_mean.mul_(1 - m).add_(stats[0], alpha=m)
        bn.running_var.mul_(1 - m).add_(stats[1], alpha=m * n / max(n - 1, 1))
        bn.num_batches_tracked += 1
    return out
