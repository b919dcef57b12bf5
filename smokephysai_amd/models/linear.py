"""Host side of libsmokehip's split-bf16 linear kernel: the nn.Linear layers of SmokePhysNet's token path
(smokephys_net.py:38,50-54,153-158; chaos_attention.py:25-28) as one fused launch each --
``y = act(x W^T + b + periodic_add) + residual`` -- with fp32-class accuracy on the bf16 matrix cores."""
import ctypes as C
from typing import Optional

import torch
from torch import nn

from .. import _lib


def split_empty(*shape, device) -> torch.Tensor:
    """Storage of an activation [..., F] in SMK_FMT_SPLIT_BF16 (x = hi + lo): a bfloat16 tensor [..., F/8, 2, 8]."""
    if shape[-1] % 8:
        raise ValueError("split-bf16 activations need a feature count that is a multiple of 8")
    return torch.empty(*shape[:-1], shape[-1] // 8, 2, 8, device=device, dtype=torch.bfloat16)


def to_split(x: torch.Tensor) -> torch.Tensor:
    """fp32 [..., F] -> SMK_FMT_SPLIT_BF16 storage (host-side helper for tests / callers; the kernels emit it themselves)."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.stack([hi.reshape(*x.shape[:-1], -1, 8), lo.reshape(*x.shape[:-1], -1, 8)], dim=-2).contiguous()


def from_split(xs: torch.Tensor) -> torch.Tensor:
    """SMK_FMT_SPLIT_BF16 storage [..., F/8, 2, 8] -> fp32 [..., F] (hi + lo)."""
    v = xs.float()
    return (v[..., 0, :] + v[..., 1, :]).reshape(*xs.shape[:-3], -1)


def hip_linear_supported(in_features: int, out_features: int) -> bool:
    """Shapes the HIP kernel is built for (include/smokehip.h: smk_linear_create)."""
    return in_features % 64 == 0 and out_features % 32 == 0


MAX_X_ELEMS = 1 << 30        # smk_linear_forward: (rows + 256) * ldx < 2^30 floats per launch (32-bit buffer offsets)


class HipLinear:
    """Device-resident, re-laid-out copy of one nn.Linear's weights + the launch wrapper."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, device=None):
        self._dev = _lib.require_cuda(device if device is not None else weight.device, "HipLinear")
        self._L = _lib.load()
        self._handle = None
        w = weight.detach().to(self._dev, torch.float32).contiguous()
        b = None if bias is None else bias.detach().to(self._dev, torch.float32).contiguous()
        self.out_features, self.in_features = w.shape
        handle = C.c_void_p()
        _lib.check(self._L.smk_linear_create(w.data_ptr(), 0 if b is None else b.data_ptr(), self.out_features,
                                             self.in_features, self._dev.index, _lib.stream_ptr(self._dev),
                                             C.byref(handle)))
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(self._dev).synchronize()      # the split kernel has read w / b
        else:
            self._keep = (w, b)
        self._handle = handle

    @classmethod
    def from_module(cls, lin: nn.Linear) -> "HipLinear":
        return cls(lin.weight, lin.bias)

    def update(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, transposed: bool = False):
        """Re-split new parameter values into this handle (smk_linear_update; no allocation).  transposed: `weight` is
        [in_features, out_features] -- the handle then computes ``x @ weight``."""
        want = (self.in_features, self.out_features) if transposed else (self.out_features, self.in_features)
        w = weight.detach()
        if tuple(w.shape) != want or w.dtype != torch.float32 or w.device != self._dev or not w.is_contiguous():
            raise ValueError(f"HipLinear.update: weight must be contiguous float32 {want} on {self._dev}")
        b = None if bias is None else bias.detach().to(self._dev, torch.float32).contiguous()
        _lib.check(self._L.smk_linear_update(self._handle, w.data_ptr(), int(transposed), 0 if b is None else b.data_ptr(),
                                             _lib.stream_ptr(self._dev)))
        self._keep = (w, b)                                     # alive until the enqueued split kernel has read them

    def close(self):
        if getattr(self, "_handle", None):
            self._L.smk_linear_destroy(self._handle)
            self._handle = None

    __del__ = close

    def __call__(self, x: torch.Tensor, activation: Optional[str] = None, residual: Optional[torch.Tensor] = None,
                 periodic_add: Optional[torch.Tensor] = None, rows_per_group: int = 0,
                 out: Optional[torch.Tensor] = None, x_split: bool = False, out_split: bool = False) -> torch.Tensor:
        """x [..., in_features] -> [..., out_features].  residual: same shape as the result, added after the activation.
        periodic_add [groups, period, out_features]: row i of group g gets periodic_add[g, i % period] before the
        activation (rows_per_group rows per group, default = x.shape[-2]).
        x_split: x is SMK_FMT_SPLIT_BF16 storage [..., in/8, 2, 8] (split_empty / to_split); out_split: return that format."""
        if x_split:
            if x.device != self._dev or x.dtype != torch.bfloat16 or x.shape[-3:] != (self.in_features // 8, 2, 8) or not x.is_contiguous():
                raise ValueError("HipLinear: x_split wants contiguous bfloat16 [..., in_features/8, 2, 8]")
            lead = x.shape[:-3]
            rows = x.numel() // (2 * self.in_features)
            x_ptr, ldx, seq = x.data_ptr(), self.in_features, (lead[-1] if lead else rows)
        else:
            if x.device != self._dev or x.dtype != torch.float32:
                raise ValueError(f"HipLinear: x must be float32 on {self._dev}")
            if x.shape[-1] != self.in_features:
                raise ValueError(f"HipLinear: last dim {x.shape[-1]} != in_features {self.in_features}")
            lead = x.shape[:-1]
            x2 = x.reshape(-1, self.in_features)
            if x2.stride(1) != 1 or x2.stride(0) % 4 != 0 or x2.data_ptr() % 16 != 0:
                x2 = x2.contiguous()
            rows = x2.shape[0]
            x_ptr, ldx, seq = x2.data_ptr(), x2.stride(0), (x.shape[-2] if x.dim() >= 2 else rows)
        if out_split:
            y = out if out is not None else split_empty(*lead, self.out_features, device=self._dev)
            y_ptr, ldy = y.data_ptr(), self.out_features
        else:
            y = out if out is not None else torch.empty(*lead, self.out_features, device=self._dev, dtype=torch.float32)
            y2 = y.view(-1, self.out_features)
            y_ptr, ldy = y2.data_ptr(), y2.stride(0)
        res_ptr, ldr = 0, 0
        if residual is not None:
            r2 = residual.reshape(-1, self.out_features)
            if r2.stride(1) != 1:
                r2 = r2.contiguous()
            if r2.shape[0] != rows or r2.dtype != torch.float32 or r2.device != self._dev:
                raise ValueError("HipLinear: residual must match the output")
            res_ptr, ldr = r2.data_ptr(), r2.stride(0)
        pa_ptr, period, rpg = 0, 1, 1
        if periodic_add is not None:
            pa = periodic_add.to(torch.float32).contiguous()
            rpg = rows_per_group or seq
            if pa.dim() != 3 or pa.shape[2] != self.out_features or pa.shape[0] * rpg != rows:
                raise ValueError("HipLinear: periodic_add must be [rows / rows_per_group, period, out_features]")
            pa_ptr, period = pa.data_ptr(), pa.shape[1]
        act = {None: _lib.SMK_ACT_NONE, "none": _lib.SMK_ACT_NONE, "gelu": _lib.SMK_ACT_GELU, "relu": _lib.SMK_ACT_RELU}[activation]
        # one launch addresses x with 32-bit offsets: larger inputs go in row chunks (whole groups when a periodic addend is set)
        unit = rpg if pa_ptr else 128
        max_rows = max(unit, ((MAX_X_ELEMS // max(ldx, 1) - 256) // unit) * unit)
        x_row_bytes = (2 * self.in_features * 2) if x_split else ldx * 4
        y_row_bytes = (2 * self.out_features * 2) if out_split else ldy * 4
        r0 = 0
        while r0 < rows:
            n = min(max_rows, rows - r0)
            _lib.check(self._L.smk_linear_forward(
                self._handle, x_ptr + r0 * x_row_bytes, n, ldx, y_ptr + r0 * y_row_bytes, ldy,
                res_ptr + r0 * ldr * 4 if res_ptr else 0, ldr,
                pa_ptr + (r0 // rpg) * period * self.out_features * 4 if pa_ptr else 0, rpg, period,
                act, int(x_split), int(out_split), _lib.stream_ptr(self._dev)))
            r0 += n
        return y


class HipLinearLN(HipLinear):
    """LayerNorm + Linear as ONE launch (smk_linear_forward_ln): the handle holds the folded parameters W' = W diag(gamma),
    b' = b + W beta; the kernel normalises inside (row statistics gathered while it stages the raw rows, the mean / rstd correction in
    its epilogue).  `max_rows`: the largest row count the handle serves (the library's own answer, smk_linear_ln_max_rows)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], ln_weight: torch.Tensor, ln_bias: torch.Tensor, eps: float,
                 device=None):
        w = weight.detach().double()
        g, be = ln_weight.detach().double().to(w.device), ln_bias.detach().double().to(w.device)
        wf = w * g[None, :]
        bf = w @ be + (0.0 if bias is None else bias.detach().double())
        super().__init__(wf.float(), bf.float(), device=device)
        self.eps = float(eps)
        self.wsum = wf.sum(dim=1).float().to(self._dev).contiguous()
        self.max_rows = int(self._L.smk_linear_ln_max_rows(self._handle))

    def forward_ln(self, x: torch.Tensor, activation: Optional[str] = None, periodic_add: Optional[torch.Tensor] = None,
                   rows_per_group: int = 0, out: Optional[torch.Tensor] = None, split_from: Optional[int] = None) -> torch.Tensor:
        """x [..., in_features] RAW (not normalised) -> act(LayerNorm(x) W^T + b + periodic_add).
        split_from (a multiple of 32): the output columns from there on are written as SMK_FMT_SPLIT4_INPLACE -- per 4 columns the 16 bytes
        {hi[0..3], lo[0..3]} (bf16) instead of 4 floats -- for hip_attention(kv_split=True); unsplit4_inplace() decodes them."""
        if x.device != self._dev or x.dtype != torch.float32 or x.shape[-1] != self.in_features:
            raise ValueError(f"HipLinearLN: x must be float32 [..., {self.in_features}] on {self._dev}")
        x2 = x.reshape(-1, self.in_features)
        if x2.stride(1) != 1 or x2.stride(0) % 4 != 0 or x2.data_ptr() % 16 != 0:
            x2 = x2.contiguous()
        rows = x2.shape[0]
        if rows > self.max_rows:
            raise ValueError(f"HipLinearLN: {rows} rows > max_rows {self.max_rows} (use the LayerNorm kernel + HipLinear)")
        y = out if out is not None else torch.empty(*x.shape[:-1], self.out_features, device=self._dev, dtype=torch.float32)
        y2 = y.view(-1, self.out_features)
        pa_ptr, period, rpg = 0, 1, 1
        if periodic_add is not None:
            pa = periodic_add.to(torch.float32).contiguous()
            rpg = rows_per_group or (x.shape[-2] if x.dim() >= 2 else rows)
            if pa.dim() != 3 or pa.shape[2] != self.out_features or pa.shape[0] * rpg != rows:
                raise ValueError("HipLinearLN: periodic_add must be [rows / rows_per_group, period, out_features]")
            pa_ptr, period = pa.data_ptr(), pa.shape[1]
        act = {None: _lib.SMK_ACT_NONE, "none": _lib.SMK_ACT_NONE, "gelu": _lib.SMK_ACT_GELU, "relu": _lib.SMK_ACT_RELU}[activation]
        _lib.check(self._L.smk_linear_forward_ln_split(self._handle, x2.data_ptr(), rows, x2.stride(0), y2.data_ptr(), y2.stride(0),
                                                       self.wsum.data_ptr(), self.eps, pa_ptr, rpg, period, act,
                                                       -1 if split_from is None else int(split_from), _lib.stream_ptr(self._dev)))
        return y


def split4_inplace(x: torch.Tensor) -> torch.Tensor:
    """fp32 [..., F] (F % 4 == 0) -> the same storage size holding SMK_FMT_SPLIT4_INPLACE: per 4 features {hi[0..3], lo[0..3]} bf16, returned
    as a float32 tensor of x's shape whose BITS are that encoding (host-side helper for tests; the q | k | v epilogue writes it itself)."""
    g = x.contiguous().view(*x.shape[:-1], x.shape[-1] // 4, 4)
    hi = g.to(torch.bfloat16)
    lo = (g - hi.to(torch.float32)).to(torch.bfloat16)
    return torch.cat([hi, lo], dim=-1).contiguous().view(torch.float32).view(x.shape)


def unsplit4_inplace(y: torch.Tensor) -> torch.Tensor:
    """Decode SMK_FMT_SPLIT4_INPLACE bits (a float32-typed tensor [..., F]) back to fp32 values hi + lo."""
    g = y.contiguous().view(torch.bfloat16).view(*y.shape[:-1], y.shape[-1] // 4, 8).to(torch.float32)
    return (g[..., :4] + g[..., 4:]).reshape(y.shape)


def hip_linear_wgrad_supported(in_features: int, out_features: int) -> bool:
    return in_features % 32 == 0 and out_features % 4 == 0


MAX_WGRAD_ROWS = 1 << 17     # rows per smk_linear_wgrad call ((out + 256) * (rows + 4096) < 2^30 holds up to out = 7,680); more: chunk + add


def hip_linear_wgrad(dy: torch.Tensor, x: torch.Tensor, want_db: bool = False):
    """dW [out, in] = dy^T x over the token rows (dy [rows, out], x [rows, in], fp32 on one ROCm device) -- smk_linear_wgrad.
    want_db: also return the bias gradient db [out] = dy.sum(0) (from the transposed copy of dy the call makes anyway)."""
    dev = _lib.require_cuda(dy.device, "hip_linear_wgrad")
    L = _lib.load()
    if dy.dim() != 2 or x.dim() != 2 or dy.shape[0] != x.shape[0] or dy.dtype != torch.float32 or x.dtype != torch.float32 or x.device != dev:
        raise ValueError("hip_linear_wgrad: dy [rows, out] and x [rows, in] float32 on the same device")
    if dy.stride(1) != 1:
        dy = dy.contiguous()
    if x.stride(1) != 1:
        x = x.contiguous()
    rows, out_f, in_f = dy.shape[0], dy.shape[1], x.shape[1]
    dw = torch.empty(out_f, in_f, device=dev, dtype=torch.float32)
    db = torch.empty(out_f, device=dev, dtype=torch.float32) if want_db else None
    r0 = 0
    while r0 < rows:
        n = min(MAX_WGRAD_ROWS, rows - r0)
        nbytes = int(L.smk_linear_wgrad_workspace(n, out_f, in_f))
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)      # torch's caching allocator: stream-ordered reuse
        tgt = dw if r0 == 0 else torch.empty_like(dw)
        tgb = None if db is None else (db if r0 == 0 else torch.empty_like(db))
        _lib.check(L.smk_linear_wgrad(dy[r0:].data_ptr(), dy.stride(0), x[r0:].data_ptr(), x.stride(0), n, out_f, in_f,
                                      tgt.data_ptr(), None if tgb is None else tgb.data_ptr(), ws.data_ptr(), nbytes, _lib.stream_ptr(dev)))
        if r0:
            dw += tgt
            if db is not None:
                db += tgb
        r0 += n
    return (dw, db) if want_db else dw


class _HipLinearFn(torch.autograd.Function):
    """y = x W^T + b with all three GEMMs of the layer on libsmokehip's split-bf16 kernel: the forward, the input gradient
    dX = dY W (a handle filled from W read transposed) and the weight gradient dW = dY^T X (smk_linear_wgrad: the reduction over
    the token rows as K-segments of one launch)."""

    @staticmethod
    def forward(ctx, x, weight, bias, mod, residual=None):
        ctx.mod = mod
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return mod._hip_forward_handle()(x, residual=residual)          # residual (the pre-LN block's `x + sublayer(x)`) rides in the epilogue

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        mod = ctx.mod
        dx = dw = db = None
        dy2 = dy.reshape(-1, mod.out_features)
        if dy2.stride(1) != 1:
            dy2 = dy2.contiguous()
        if ctx.needs_input_grad[0]:
            dx = mod._hip_backward_handle()(dy2).view(x.shape)
        if ctx.needs_input_grad[1]:
            x2 = x.reshape(-1, mod.in_features)
            if mod.hip_wgrad and hip_linear_wgrad_supported(mod.in_features, mod.out_features):
                if ctx.has_bias and ctx.needs_input_grad[2]:
                    dw, db = hip_linear_wgrad(dy2, x2, want_db=True)      # db rides on the transposed copy of dy
                else:
                    dw = hip_linear_wgrad(dy2, x2)
            else:
                dw = dy2.t().mm(x2)
        if db is None and ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(0)
        dres = dy if len(ctx.needs_input_grad) > 4 and ctx.needs_input_grad[4] else None      # d(y + residual) / d residual = identity
        return dx, dw, db, None, dres


class TrainableHipLinear(nn.Linear):
    """nn.Linear (same parameters, init and state_dict keys) whose training forward and input gradient run on libsmokehip when
    `hip_train` is set, the tensors live on a ROCm device and the shape is one the kernel is built for; otherwise F.linear.
    The two device mirrors of the weight (W for the forward, W^T for dX) are re-split when the parameter changes."""

    hip_train = False
    hip_wgrad = True         # dW on libsmokehip too (False: PyTorch-ROCm fp32 GEMM for the weight gradient only)

    def __getstate__(self):            # deepcopy / pickling: the device mirrors are per instance, rebuilt on first use
        d = self.__dict__.copy()
        for k in ("_hip_fwd", "_hip_bwd", "_hip_fwd_fp", "_hip_bwd_fp"):
            d.pop(k, None)
        return d

    def _hip_ok(self, x: torch.Tensor) -> bool:
        return (self.hip_train and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
                and (x.requires_grad or self.weight.requires_grad)
                and hip_linear_supported(self.in_features, self.out_features)
                and hip_linear_supported(self.out_features, self.in_features))

    def _fingerprint(self):
        b = self.bias
        return (self.weight.data_ptr(), self.weight._version, None if b is None else (b.data_ptr(), b._version))

    def _hip_forward_handle(self) -> HipLinear:
        fp = self._fingerprint()
        h = self.__dict__.get("_hip_fwd")
        if h is None:
            h = self.__dict__["_hip_fwd"] = HipLinear(self.weight, self.bias)
        elif self.__dict__.get("_hip_fwd_fp") != fp:
            h.update(self.weight, self.bias)
        self.__dict__["_hip_fwd_fp"] = fp
        return h

    def _hip_backward_handle(self) -> HipLinear:
        fp = (self.weight.data_ptr(), self.weight._version)
        h = self.__dict__.get("_hip_bwd")
        if h is None:       # a handle with out = in_features, in = out_features, filled from W read as [in' = out][out' = in]
            h = self.__dict__["_hip_bwd"] = HipLinear(self.weight.detach().t().contiguous(), None)
        elif self.__dict__.get("_hip_bwd_fp") != fp:
            h.update(self.weight, None, transposed=True)
        self.__dict__["_hip_bwd_fp"] = fp
        return h

    def forward(self, x: torch.Tensor, residual: torch.Tensor = None) -> torch.Tensor:
        """residual (optional, same shape as the result): returns linear(x) + residual -- on the libsmokehip route as the GEMM's epilogue."""
        if self._hip_ok(x):
            return _HipLinearFn.apply(x, self.weight, self.bias, self, residual)
        y = nn.functional.linear(x, self.weight, self.bias)
        return y if residual is None else y + residual
