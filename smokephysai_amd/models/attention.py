"""Host side of libsmokehip's flash attention (smk_attention): softmax(Q K^T * scale) V over token-major tensors."""
from typing import Optional

import torch

from .. import _lib
from .linear import split_empty


MAX_QKV_ELEMS = 1 << 29      # smk_attention: B * L * ld < 2^29 floats per tensor and launch (32-bit buffer offsets)


def hip_attention_supported(L: int, head_dim: int) -> bool:
    return head_dim == 64 and L % 128 == 0 and L >= 128


def hip_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int, scale: float,
                  out: Optional[torch.Tensor] = None, out_split: bool = False, kv_split: bool = False) -> torch.Tensor:
    """q, k, v: [B, L, H*64] float32 on a ROCm device (row pitch may exceed H*64, e.g. slices of a fused qkv tensor);
    returns [B, L, H*64] -- the layout `out.transpose(1, 2).contiguous().view(B, L, D)` has in the reference
    (chaos_attention.py:111-112).  kv_split: the BITS of k and v are SMK_FMT_SPLIT4_INPLACE (what HipLinearLN.forward_ln(split_from=...)
    wrote), not fp32 values; the result is bit for bit the one of the fp32 k and v they encode."""
    dev = _lib.require_cuda(q.device, "hip_attention")
    B, L, D = q.shape
    d = D // num_heads
    for t in (q, k, v):
        if t.dtype != torch.float32 or t.shape != q.shape or t.device != q.device or t.stride(2) != 1 or t.stride(0) != L * t.stride(1):
            raise ValueError("hip_attention: q, k, v must be float32 [B, L, H*d] with unit inner stride and dense batches")
    if out is None:
        out = split_empty(B, L, D, device=dev) if out_split else torch.empty(B, L, D, device=dev, dtype=torch.float32)
    ldo = D if out_split else out.stride(1)
    ld_max = max(q.stride(1), k.stride(1), v.stride(1), ldo)
    bmax = max(1, (MAX_QKV_ELEMS - 1) // (L * ld_max))       # larger batches go in batch chunks (independent problems)
    o_batch_bytes = L * D * 4 if out_split else out.stride(0) * 4
    L_ = _lib.load()
    # small batches (one frame: 64 workgroups for 256 CUs): the keys are split over workgroups and merged by a second launch; the library
    # says how much workspace that takes for this problem (0: it would not split)
    ws_bytes = 0 if out_split else int(L_.smk_attention_workspace_bytes(min(B, bmax), L, num_heads, d))
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8) if ws_bytes > 0 else None
    b0 = 0
    while b0 < B:
        nb = min(bmax, B - b0)
        _lib.check(L_.smk_attention_kv(q.data_ptr() + b0 * q.stride(0) * 4, k.data_ptr() + b0 * k.stride(0) * 4,
                                       v.data_ptr() + b0 * v.stride(0) * 4, out.data_ptr() + b0 * o_batch_bytes, nb, L,
                                       num_heads, d, q.stride(1), k.stride(1), v.stride(1), ldo, float(scale),
                                       int(out_split), _lib.SMK_FMT_SPLIT4_INPLACE if kv_split else _lib.SMK_FMT_F32,
                                       ws.data_ptr() if ws is not None and nb == min(B, bmax) else None, ws_bytes, _lib.stream_ptr(dev)))
        b0 += nb
    return out


def hip_attention_delta(dout: torch.Tensor, out: torch.Tensor, num_heads: int) -> torch.Tensor:
    """delta [B, L, H] = (dout * out).view(B, L, H, 64).sum(-1) as one libsmokehip pass (smk_attention_delta)."""
    dev = _lib.require_cuda(dout.device, "hip_attention_delta")
    B, L, D = out.shape
    d2, o2 = dout.reshape(B * L, D), out.reshape(B * L, D)
    if d2.stride(1) != 1 or d2.stride(0) % 4 != 0 or d2.data_ptr() % 16 != 0:
        d2 = d2.contiguous()
    if o2.stride(1) != 1 or o2.stride(0) % 4 != 0 or o2.data_ptr() % 16 != 0:
        o2 = o2.contiguous()
    delta = torch.empty(B, L, num_heads, device=dev, dtype=torch.float32)
    _lib.check(_lib.load().smk_attention_delta(d2.data_ptr(), o2.data_ptr(), B * L, num_heads, D // num_heads, d2.stride(0), o2.stride(0),
                                               delta.data_ptr(), _lib.stream_ptr(dev)))
    return delta


class _HipAttentionFn(torch.autograd.Function):
    """softmax(q k^T * scale) v over token-major [B, L, H*64] tensors with forward AND backward on libsmokehip
    (smk_attention_forward_lse / smk_attention_backward)."""

    @staticmethod
    def forward(ctx, q, k, v, num_heads, scale):
        dev = _lib.require_cuda(q.device, "hip_attention_train")
        B, L, D = q.shape
        q, k, v = [t if (t.stride(2) == 1 and t.stride(0) == L * t.stride(1) and t.stride(1) % 4 == 0 and t.data_ptr() % 16 == 0)
                   else t.contiguous() for t in (q, k, v)]
        out = torch.empty(B, L, D, device=dev, dtype=torch.float32)
        lse = torch.empty(B, L, num_heads, device=dev, dtype=torch.float32)
        bmax = max(1, (MAX_QKV_ELEMS - 1) // (L * max(q.stride(1), k.stride(1), v.stride(1), D)))
        for b0 in range(0, B, bmax):
            nb = min(bmax, B - b0)
            _lib.check(_lib.load().smk_attention_forward_lse(
                q[b0:].data_ptr(), k[b0:].data_ptr(), v[b0:].data_ptr(), out[b0:].data_ptr(), lse[b0:].data_ptr(), nb, L, num_heads,
                D // num_heads, q.stride(1), k.stride(1), v.stride(1), D, float(scale), _lib.stream_ptr(dev)))
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.num_heads, ctx.scale = num_heads, float(scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        B, L, D = q.shape
        H = ctx.num_heads
        dout = dout.contiguous()
        delta = hip_attention_delta(dout, out, H)                               # [B, L, H] = rowsum(dout * out) per head
        dq, dk, dv = torch.empty_like(out), torch.empty_like(out), torch.empty_like(out)
        dev = q.device
        bmax = max(1, (MAX_QKV_ELEMS - 1) // (L * max(q.stride(1), k.stride(1), v.stride(1), D)))
        for b0 in range(0, B, bmax):
            nb = min(bmax, B - b0)
            _lib.check(_lib.load().smk_attention_backward(
                q[b0:].data_ptr(), k[b0:].data_ptr(), v[b0:].data_ptr(), dout[b0:].data_ptr(), lse[b0:].data_ptr(),
                delta[b0:].data_ptr(), dq[b0:].data_ptr(), dk[b0:].data_ptr(), dv[b0:].data_ptr(), nb, L, H, D // H,
                q.stride(1), k.stride(1), v.stride(1), D, D, D, D, ctx.scale, _lib.stream_ptr(dev)))
        return dq, dk, dv, None, None


def hip_attention_train(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int, scale: float) -> torch.Tensor:
    """Differentiable hip_attention: q, k, v [B, L, H*64] float32 on a ROCm device -> [B, L, H*64]."""
    if q.dtype != torch.float32 or q.shape != k.shape or q.shape != v.shape:
        raise ValueError("hip_attention_train: q, k, v float32 of one shape [B, L, H*64]")
    return _HipAttentionFn.apply(q, k, v, num_heads, scale)


def hip_layernorm_supported(D: int) -> bool:
    return D % 4 == 0 and 4 <= D <= 2048


def hip_layernorm(x: torch.Tensor, ln: torch.nn.LayerNorm, out: Optional[torch.Tensor] = None,
                  out_split: bool = False) -> torch.Tensor:
    """nn.LayerNorm over the last dimension as one libsmokehip launch (smk_layernorm); x [..., D] float32, dense rows."""
    dev = _lib.require_cuda(x.device, "hip_layernorm")
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    if x2.stride(1) != 1 or x.dtype != torch.float32:
        raise ValueError("hip_layernorm: float32 rows with unit inner stride")
    if out is None:
        out = split_empty(*x.shape, device=dev) if out_split else torch.empty(x.shape, device=dev, dtype=torch.float32)
    ldy = D if out_split else out.view(-1, D).stride(0)
    _lib.check(_lib.load().smk_layernorm(x2.data_ptr(), x2.shape[0], D, x2.stride(0), ln.weight.data_ptr(), ln.bias.data_ptr(),
                                        float(ln.eps), out.data_ptr(), ldy, int(out_split), _lib.stream_ptr(dev)))
    return out


class _HipLayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension with forward and backward on libsmokehip (smk_layernorm / smk_layernorm_backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        dev = _lib.require_cuda(x.device, "hip_layernorm_train")
        D = x.shape[-1]
        x2 = x.reshape(-1, D)
        if x2.stride(1) != 1 or x2.stride(0) % 4 != 0 or x2.data_ptr() % 16 != 0:
            x2 = x2.contiguous()
        w, b = weight.detach().contiguous(), bias.detach().contiguous()
        y = torch.empty(x.shape, device=dev, dtype=torch.float32)
        _lib.check(_lib.load().smk_layernorm(x2.data_ptr(), x2.shape[0], D, x2.stride(0), w.data_ptr(), b.data_ptr(), float(eps),
                                            y.data_ptr(), D, 0, _lib.stream_ptr(dev)))
        ctx.save_for_backward(x2, w)
        ctx.eps = float(eps)
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        L = _lib.load()
        dev = x2.device
        rows, D = x2.shape
        dy2 = dy.reshape(rows, D)
        if dy2.stride(1) != 1 or dy2.stride(0) % 4 != 0 or dy2.data_ptr() % 16 != 0:
            dy2 = dy2.contiguous()
        dx = torch.empty(rows, D, device=dev, dtype=torch.float32)
        dwb = torch.empty(2, D, device=dev, dtype=torch.float32)
        ws = torch.empty(int(L.smk_layernorm_bwd_workspace(D)), device=dev, dtype=torch.uint8)
        _lib.check(L.smk_layernorm_backward(x2.data_ptr(), dy2.data_ptr(), rows, D, x2.stride(0), dy2.stride(0), w.data_ptr(), ctx.eps,
                                            dx.data_ptr(), D, dwb[0].data_ptr(), dwb[1].data_ptr(), ws.data_ptr(), _lib.stream_ptr(dev)))
        return dx.view(ctx.shape), dwb[0], dwb[1], None


def hip_layernorm_train(x: torch.Tensor, ln: torch.nn.LayerNorm) -> torch.Tensor:
    """Differentiable hip_layernorm (x [..., D] float32 on a ROCm device; ln with elementwise affine over the last dimension)."""
    if ln.weight is None or ln.bias is None or tuple(ln.normalized_shape) != (x.shape[-1],):
        raise ValueError("hip_layernorm_train: an affine LayerNorm over the last dimension")
    return _HipLayerNormFn.apply(x, ln.weight, ln.bias, ln.eps)



# ---------------------------------------------------------------- training: fused q|k|v projection + attention as ONE autograd node
class _HipQKVAttentionFn(torch.autograd.Function):
    """ChaosAttention's q / k / v projections, the chaos term folded into q, and the softmax attention (chaos_attention.py:77-112) as one
    node.  Forward: ONE [3D, D] linear launch (x read once; the [B, 5, D] chaos addend rides in the epilogue on the q columns, as in the
    eval body) + the flash attention on strided slices of its output.  Backward: the attention backward writes dq | dk | dv into ONE
    [B, L, 3D] buffer, so dX is one GEMM over K = 3D (instead of three GEMMs and two adds), dW / db one weight-gradient call with
    out = 3D (x split once instead of three times), and the addend's gradient five strided row sums of the dq columns."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, add5, mod, num_heads, scale):
        dev = _lib.require_cuda(x.device, "hip_qkv_attention_train")
        L_ = _lib.load()
        B, L, D = x.shape
        x = x.contiguous()
        fwd, _ = mod._hip_qkv_handles()
        add15 = None
        if add5 is not None:
            add15 = torch.zeros(B, 5, 3 * D, device=dev, dtype=torch.float32)
            add15[:, :, :D] = add5
        qkv = fwd(x, periodic_add=add15, rows_per_group=L)                      # [B, L, 3D]
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        out = torch.empty(B, L, D, device=dev, dtype=torch.float32)
        lse = torch.empty(B, L, num_heads, device=dev, dtype=torch.float32)
        bmax = max(1, (MAX_QKV_ELEMS - 1) // (L * 3 * D))
        for b0 in range(0, B, bmax):
            nb = min(bmax, B - b0)
            _lib.check(L_.smk_attention_forward_lse(q[b0:].data_ptr(), k[b0:].data_ptr(), v[b0:].data_ptr(), out[b0:].data_ptr(),
                                                    lse[b0:].data_ptr(), nb, L, num_heads, D // num_heads, 3 * D, 3 * D, 3 * D, D,
                                                    float(scale), _lib.stream_ptr(dev)))
        ctx.save_for_backward(x, qkv, out, lse)
        ctx.mod, ctx.num_heads, ctx.scale, ctx.has_add = mod, num_heads, float(scale), add5 is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        from .linear import hip_linear_wgrad
        x, qkv, out, lse = ctx.saved_tensors
        L_ = _lib.load()
        B, L, D = x.shape
        H = ctx.num_heads
        dev = x.device
        dout = dout.contiguous()
        delta = hip_attention_delta(dout, out, H)                               # [B, L, H] = rowsum(dout * out) per head
        dqkv = torch.empty(B, L, 3 * D, device=dev, dtype=torch.float32)
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
        bmax = max(1, (MAX_QKV_ELEMS - 1) // (L * 3 * D))
        for b0 in range(0, B, bmax):
            nb = min(bmax, B - b0)
            _lib.check(L_.smk_attention_backward(q[b0:].data_ptr(), k[b0:].data_ptr(), v[b0:].data_ptr(), dout[b0:].data_ptr(),
                                                 lse[b0:].data_ptr(), delta[b0:].data_ptr(), dq[b0:].data_ptr(), dk[b0:].data_ptr(),
                                                 dv[b0:].data_ptr(), nb, L, H, D // H, 3 * D, 3 * D, 3 * D, D, 3 * D, 3 * D, 3 * D,
                                                 ctx.scale, _lib.stream_ptr(dev)))
        dy2, x2 = dqkv.view(-1, 3 * D), x.view(-1, D)
        need = ctx.needs_input_grad
        dx = None
        if need[0]:
            _, bwd = ctx.mod._hip_qkv_handles()
            dx = bwd(dy2).view(B, L, D)
        dw = db = None
        if any(need[1:7]):
            dw, db = hip_linear_wgrad(dy2, x2, want_db=True)                    # [3D, D], [3D]
        dadd5 = None
        if ctx.has_add and need[7]:
            dadd5 = torch.stack([dq[:, r::5].sum(1) for r in range(5)], dim=1)  # row l of the sequence received addend row l % 5
        g = lambda t, i: None if t is None else t[i * D:(i + 1) * D]
        return (dx, g(dw, 0), g(db, 0), g(dw, 1), g(db, 1), g(dw, 2), g(db, 2), dadd5, None, None, None)


def hip_qkv_attention_train(mod, x: torch.Tensor, add5: Optional[torch.Tensor], num_heads: int, scale: float) -> torch.Tensor:
    """`mod`: a ChaosAttention (q_proj / k_proj / v_proj with biases).  x [B, L, D] float32 on a ROCm device; add5 [B, 5, D] or None."""
    return _HipQKVAttentionFn.apply(x, mod.q_proj.weight, mod.q_proj.bias, mod.k_proj.weight, mod.k_proj.bias, mod.v_proj.weight,
                                    mod.v_proj.bias, add5, mod, num_heads, scale)
