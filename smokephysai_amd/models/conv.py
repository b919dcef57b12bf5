"""Host side of libsmokehip's training convolution (smk_conv2_train_forward): SmokePhysNet.input_encoder's second convolution
(smokephys_net.py:28, Conv2d(64, 128, 3, padding=1)) under autograd.  The data gradient (csrc/encoder.hip: k_conv2_dgrad_b16) and the weight / bias gradients (k_conv2_wgrad_b16) run on
split-bf16 MFMA kernels, and so does the forward (k_conv2_fwd_b16, three bf16 terms per operand: see hip_conv2_train)."""
import torch
from torch import nn

from .. import _lib


def hip_conv2_train_supported(x: torch.Tensor, conv: nn.Conv2d) -> bool:
    return (x.dim() == 4 and x.is_cuda and x.dtype == torch.float32 and conv.in_channels == 64 and conv.out_channels == 128
            and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros" and x.shape[1] == 64 and x.shape[2] % 8 == 0 and x.shape[3] % 16 == 0)


class _HipConv2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, hip_forward, hip_wgrad):
        dev = _lib.require_cuda(x.device, "hip_conv2_train")
        L = _lib.load()
        x = x.contiguous()
        B, _, H, W = x.shape
        if hip_forward:
            z = torch.empty(B, 128, H, W, device=dev, dtype=torch.float32)
            ws = torch.empty(int(L.smk_conv2_train_workspace()), device=dev, dtype=torch.uint8)
            w = weight.detach().contiguous()
            b = None if bias is None else bias.detach().contiguous()
            _lib.check(L.smk_conv2_train_forward(x.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(), B, H, W, z.data_ptr(),
                                                 ws.data_ptr(), _lib.stream_ptr(dev)))
        else:                                               # PyTorch-ROCm's fp32 convolution
            z = torch.ops.aten.convolution(x, weight.detach(), None if bias is None else bias.detach(), [1, 1], [1, 1], [1, 1], False, [0, 0], 1)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.hip_wgrad = hip_wgrad
        return z

    @staticmethod
    def backward(ctx, dz):
        x, weight = ctx.saved_tensors
        dz = dz.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:                         # the data gradient on libsmokehip (k_conv2_dgrad_b16)
            L = _lib.load()
            B, _, H, W = x.shape
            dx = torch.empty_like(x)
            ws = torch.empty(int(L.smk_conv2_train_workspace()), device=x.device, dtype=torch.uint8)
            _lib.check(L.smk_conv2_train_dgrad(dz.data_ptr(), weight.detach().contiguous().data_ptr(), B, H, W, dx.data_ptr(), ws.data_ptr(),
                                               _lib.stream_ptr(x.device)))
        mask = [False, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]]
        dw = db = None
        if (mask[1] or mask[2]) and ctx.hip_wgrad:          # weight / bias gradients on libsmokehip (k_conv2_wgrad_b16)
            L = _lib.load()
            B, _, H, W = x.shape
            dw = torch.empty_like(weight)
            db = torch.empty(128, device=x.device, dtype=torch.float32) if mask[2] else None
            ws = torch.empty(int(L.smk_conv2_train_wgrad_workspace()), device=x.device, dtype=torch.uint8)
            _lib.check(L.smk_conv2_train_wgrad(dz.data_ptr(), x.data_ptr(), B, H, W, dw.data_ptr(), None if db is None else db.data_ptr(),
                                               ws.data_ptr(), _lib.stream_ptr(x.device)))
            if not mask[1]:
                dw = None
        elif mask[1] or mask[2]:                            # ... or PyTorch-ROCm's
            _, dw, db = torch.ops.aten.convolution_backward(dz, x, weight, [128] if ctx.has_bias else None, [1, 1], [1, 1], [1, 1],
                                                            False, [0, 0], 1, mask)
        return dx, dw, db, None, None


def hip_conv2_train(x: torch.Tensor, conv: nn.Conv2d, hip_forward: bool = True, hip_wgrad: bool = True) -> torch.Tensor:
    """conv(x) for the encoder's 64 -> 128 3x3 convolution under autograd on libsmokehip: the forward (k_conv2_fwd_b16), the data gradient
    (k_conv2_dgrad_b16) and the weight / bias gradients (k_conv2_wgrad_b16); hip_forward / hip_wgrad = False leave that pass to
    PyTorch-ROCm.  Raises off a ROCm device: no CPU fallback.

    The forward uses THREE bf16 terms per operand and six products (9e-7 max-norm from an fp64 convolution; MIOpen's fp32 Winograd: 4e-7),
    not the two terms / three products of the gradients and of the eval encoder (5e-6): this output feeds train-mode BatchNorm + ReLU, and
    in fp64, noise of relative size 5e-7 / 5e-6 on it moves conv2.weight.grad by 4e-3 / 1.7e-2 (ReLU masks flip) -- a three-product forward
    took the full-loss gradients out of the band PyTorch's own fp32 run stays in (tests/test_hip_pipeline.py).  The gradients have no
    such amplifier: their 5e-6 propagates linearly."""
    if not hip_conv2_train_supported(x, conv):
        raise ValueError("hip_conv2_train: a float32 ROCm tensor [B, 64, H, W] with H % 8 == 0, W % 16 == 0 and Conv2d(64, 128, 3, padding=1)")
    return _HipConv2Fn.apply(x, conv.weight, conv.bias, bool(hip_forward), bool(hip_wgrad))


# ---------------------------------------------------------------- the first convolution: Conv2d(1, 64, 7, padding=3) (smokephys_net.py:25)
def hip_conv1_train_supported(x: torch.Tensor, conv: nn.Conv2d) -> bool:
    return (x.dim() == 4 and x.is_cuda and x.dtype == torch.float32 and conv.in_channels == 1 and conv.out_channels == 64
            and conv.kernel_size == (7, 7) and conv.stride == (1, 1) and conv.padding == (3, 3) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros" and x.shape[1] == 1 and x.shape[2] % 4 == 0 and x.shape[3] % 64 == 0
            and x.shape[0] <= 65535)


class _HipConv1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        dev = _lib.require_cuda(x.device, "hip_conv1_train")
        L = _lib.load()
        x = x.contiguous()
        B, _, H, W = x.shape
        z = torch.empty(B, 64, H, W, device=dev, dtype=torch.float32)
        w = weight.detach().contiguous()
        b = None if bias is None else bias.detach().contiguous()
        _lib.check(L.smk_conv1_train_forward(x.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(), B, H, W, z.data_ptr(),
                                             _lib.stream_ptr(dev)))
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return z

    @staticmethod
    def backward(ctx, dz):
        x, weight = ctx.saved_tensors
        dz = dz.contiguous()
        L = _lib.load()
        B, _, H, W = x.shape
        dx = dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(weight)
            db = torch.empty(64, device=x.device, dtype=torch.float32) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
            ws = torch.empty(int(L.smk_conv1_train_wgrad_workspace()), device=x.device, dtype=torch.uint8)
            _lib.check(L.smk_conv1_train_wgrad(dz.data_ptr(), x.data_ptr(), B, H, W, dw.data_ptr(), None if db is None else db.data_ptr(),
                                               ws.data_ptr(), _lib.stream_ptr(x.device)))
            if not ctx.needs_input_grad[1]:
                dw = None
        if ctx.needs_input_grad[0]:                         # (train.py's frames carry no gradient; a caller that wants dX gets PyTorch-ROCm's)
            dx = torch.ops.aten.convolution_backward(dz, x, weight, None, [1, 1], [3, 3], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        return dx, dw, db


def hip_conv1_train(x: torch.Tensor, conv: nn.Conv2d) -> torch.Tensor:
    """conv(x) for the encoder's 1 -> 64 7x7 convolution under autograd on libsmokehip (k_conv1_train_fwd / k_conv1_train_wgrad: plain fp32 on
    the vector ALUs).  Raises off a ROCm device: no CPU fallback."""
    if not hip_conv1_train_supported(x, conv):
        raise ValueError("hip_conv1_train: a float32 ROCm tensor [B, 1, H, W] with H % 4 == 0, W % 64 == 0 and Conv2d(1, 64, 7, padding=3)")
    return _HipConv1Fn.apply(x, conv.weight, conv.bias)

