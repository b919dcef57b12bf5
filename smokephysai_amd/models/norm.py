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


# ---------------------------------------------------------------- cross-rank statistics (SyncBatchNorm2d) on the same kernels
BN_STATS, BN_APPLY, BN_BWD_SUMS, BN_BWD_DZ = 0, 1, 2, 3


def _phase(L, phase, z, dout, gamma, beta, eps, mean, var, rstd, pool, out, dz, dgamma, dbeta, count, ws, dev):
    B, C, H, W = z.shape
    ptr = lambda t: None if t is None else t.data_ptr()
    _lib.check(L.smk_bn_relu_pool_phase(phase, z.data_ptr(), ptr(dout), B, C, H, W, gamma.data_ptr(), beta.data_ptr(), float(eps),
                                        mean.data_ptr(), ptr(var), rstd.data_ptr(), pool, ptr(out), ptr(dz), ptr(dgamma), ptr(dbeta),
                                        float(count), ptr(ws), _lib.stream_ptr(dev)))


def _combine_stats(mean, var, n_local, group):
    """Per-rank (mean, biased var, count) -> the GLOBAL batch's (mean, biased var, count): one all-gather of 2C + 1 floats per rank,
    combined in fp64 with the shifted form var = sum n_r (var_r + (mean_r - mean)^2) / N (no cancellation)."""
    import torch.distributed as dist
    C = mean.numel()
    mine = torch.cat([mean.double(), var.double(), torch.tensor([float(n_local)], dtype=torch.float64, device=mean.device)])
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        return mean, var, float(n_local)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    allp = torch.stack(parts)                                   # [world, 2C + 1]
    n_r = allp[:, -1:]
    N = n_r.sum()
    gmean = (allp[:, :C] * n_r).sum(0) / N
    gvar = ((allp[:, C:2 * C] + (allp[:, :C] - gmean) ** 2) * n_r).sum(0) / N
    return gmean.float(), gvar.float(), float(N)


class _HipSyncBnReluPoolFn(torch.autograd.Function):
    """_HipBnReluPoolFn with the statistics all-reduced over the process group between the kernel passes."""

    @staticmethod
    def forward(ctx, z, weight, bias, eps, pool, group):
        dev = _lib.require_cuda(z.device, "hip_sync_bn_relu_pool")
        L = _lib.load()
        z = z.contiguous()
        B, C, H, W = z.shape
        w, b = weight.detach().contiguous(), bias.detach().contiguous()
        stats = torch.empty(3, C, device=dev, dtype=torch.float32)
        ws = torch.empty(int(L.smk_bn_train_workspace(B, C, H, W, pool)), device=dev, dtype=torch.uint8)
        _phase(L, BN_STATS, z, None, w, b, eps, stats[0], stats[1], stats[2], pool, None, None, None, None, 0.0, ws, dev)
        gmean, gvar, N = _combine_stats(stats[0], stats[1], B * H * W, group)
        # one process: the kernel's own 1 / sqrtf(var + eps) (bit-identical to the fused call); several: the same expression on the combined variance
        grstd = stats[2] if gvar is stats[1] else 1.0 / torch.sqrt(gvar + eps)
        out = torch.empty(B, C, H // pool, W // pool, device=dev, dtype=torch.float32)
        gm, gr = gmean.contiguous(), grstd.contiguous()
        _phase(L, BN_APPLY, z, None, w, b, eps, gm, None, gr, pool, out, None, None, None, 0.0, None, dev)
        ctx.save_for_backward(z, w, b, gm, gr)
        ctx.pool, ctx.group, ctx.N = pool, group, N
        gstats = torch.stack([gm, gvar.contiguous()])
        ctx.mark_non_differentiable(gstats)
        return out, gstats, torch.tensor(N)

    @staticmethod
    def backward(ctx, dout, _ds, _dn):
        import torch.distributed as dist
        z, w, b, gm, gr = ctx.saved_tensors
        L = _lib.load()
        dev = z.device
        B, C, H, W = z.shape
        dout = dout.contiguous()
        dwb = torch.empty(2, C, device=dev, dtype=torch.float32)
        ws = torch.empty(int(L.smk_bn_train_workspace(B, C, H, W, ctx.pool)), device=dev, dtype=torch.uint8)
        _phase(L, BN_BWD_SUMS, z, dout, w, b, 0.0, gm, None, gr, ctx.pool, None, None, dwb[0], dwb[1], 0.0, ws, dev)
        tot = dwb.clone()                                        # dgamma / dbeta returned to autograd stay LOCAL (DDP averages them)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(ctx.group) > 1:
            dist.all_reduce(tot, group=ctx.group)
        dz = torch.empty_like(z)
        _phase(L, BN_BWD_DZ, z, dout, w, b, 0.0, gm, None, gr, ctx.pool, None, dz, tot[0], tot[1], ctx.N, None, dev)
        return dz, dwb[0], dwb[1], None, None, None


def hip_sync_bn_relu_pool(z: torch.Tensor, bn, pool: int = 1) -> torch.Tensor:
    """hip_bn_relu_pool for a SyncBatchNorm2d: batch statistics over every rank of bn.process_group, running statistics updated from
    the global batch exactly like nn.BatchNorm2d does from a single-process one."""
    if not (bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None):
        raise ValueError("hip_sync_bn_relu_pool: a training-mode affine SyncBatchNorm2d with running statistics and a fixed momentum")
    out, gstats, n = _HipSyncBnReluPoolFn.apply(z, bn.weight, bn.bias, bn.eps, pool, bn.process_group)
    with torch.no_grad():
        nn_ = float(n)
        m = bn.momentum
        bn.running_mean.mul_(1 - m).add_(gstats[0], alpha=m)
        bn.running_var.mul_(1 - m).add_(gstats[1], alpha=m * nn_ / max(nn_ - 1.0, 1.0))
        bn.num_batches_tracked += 1
    return out
