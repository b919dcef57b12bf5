"""Host side of libsmokehip's fused element-wise FFN kernels (smk_ffn_elementwise) under autograd: GELU + dropout and dropout + residual
add of ChaosTransformerLayer (smokephys_net.py:153-159,165-167), one read and one write per tensor, no mask tensors (the keep mask is
recomputed in the backward from the call's seed)."""
import torch

from .. import _lib

GELU_DROPOUT_FWD, GELU_DROPOUT_BWD, DROPOUT_ADD_FWD, DROPOUT_BWD = 0, 1, 2, 3

def _next_seed(p: float) -> int:
    """A fresh 62-bit seed per call from torch's CPU generator: deterministic under torch.manual_seed, a host-side draw (no device
    synchronisation).  p == 0 keeps every element whatever the seed: no draw, so deterministic runs leave the RNG stream untouched."""
    if p == 0.0:
        return 0
    return int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64).item())


def hip_ffn_elementwise_supported(t: torch.Tensor) -> bool:
    return t.is_cuda and t.dtype == torch.float32 and t.numel() % 4 == 0


def _run(op, a, b, p, seed):
    dev = _lib.require_cuda(a.device, "hip_ffn_elementwise")
    out = torch.empty_like(a)
    _lib.check(_lib.load().smk_ffn_elementwise(op, a.data_ptr(), None if b is None else b.data_ptr(), out.data_ptr(), a.numel(), float(p),
                                               seed, _lib.stream_ptr(dev)))
    return out


class _GeluDropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, p, seed):
        h = h.contiguous()
        ctx.save_for_backward(h)
        ctx.p, ctx.seed = p, seed
        return _run(GELU_DROPOUT_FWD, h, None, p, seed)

    @staticmethod
    def backward(ctx, dout):
        (h,) = ctx.saved_tensors
        return _run(GELU_DROPOUT_BWD, h, dout.contiguous(), ctx.p, ctx.seed), None, None


class _DropoutAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, residual, p, seed):
        ctx.p, ctx.seed = p, seed
        return _run(DROPOUT_ADD_FWD, y.contiguous(), residual.contiguous(), p, seed)

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        dy = _run(DROPOUT_BWD, dout, None, ctx.p, ctx.seed) if ctx.needs_input_grad[0] else None
        return dy, (dout if ctx.needs_input_grad[1] else None), None, None


def hip_gelu_dropout(h: torch.Tensor, p: float, training: bool = True) -> torch.Tensor:
    """dropout_p(gelu(h)) (exact-erf GELU); p is ignored (0) when not training."""
    p = float(p) if training else 0.0
    return _GeluDropoutFn.apply(h, p, _next_seed(p))


def hip_dropout_add(y: torch.Tensor, residual: torch.Tensor, p: float, training: bool = True) -> torch.Tensor:
    """residual + dropout_p(y)."""
    p = float(p) if training else 0.0
    return _DropoutAddFn.apply(y, residual, p, _next_seed(p))
