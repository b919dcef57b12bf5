"""The eval-mode transformer body on libsmokehip: caches the re-laid-out weights of every token-wise nn.Linear and runs
one pre-LN ChaosTransformerLayer (smokephys_net.py:136-168; chaos_attention.py:68-114) as 7 launches --
LayerNorm, chaos addend, fused q|k|v projection (+ chaos term on the q columns), flash attention, out_proj (+ residual),
LayerNorm, FFN up (+ GELU), FFN down (+ residual)."""
import math
import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .attention import hip_attention, hip_attention_supported, hip_layernorm, hip_layernorm_supported
from .linear import HipLinear, HipLinearLN, hip_linear_supported, to_split


def _fp(lin: nn.Linear):
    return (lin.weight.data_ptr(), lin.weight._version, None if lin.bias is None else (lin.bias.data_ptr(), lin.bias._version))


class HipBody:
    def __init__(self):
        self.linears: Dict[str, Tuple[HipLinear, tuple]] = {}      # name -> (handle, fingerprint of the source tensors)
        self.sources: Dict[str, tuple] = {}                        # name -> the nn.Linear modules the handle mirrors
        self.addend_bufs: Dict[tuple, torch.Tensor] = {}           # (name, batch) -> [B,5,3D] addend of the fused q|k|v layer
        self.split_activations = os.environ.get("SMK_BODY_SPLIT", "0") == "1"
        self.fuse_layernorm = os.environ.get("SMK_BODY_FUSE_LN", "1") == "1"     # each LayerNorm inside the layer behind it (HipLinearLN)
        # the fused q | k | v layer writes k and v as in-place split-bf16 and the attention stages them without split arithmetic (once per
        # element instead of once per 128-query block; same bits, so same output).  SMK_BODY_KV_SPLIT=0: fp32 k and v
        self.kv_presplit = os.environ.get("SMK_BODY_KV_SPLIT", "1") == "1"

    # ---- weight mirrors ----------------------------------------------------------------------------------------
    def linear(self, name: str, lin: nn.Linear) -> HipLinear:
        """Device copy of one nn.Linear in the kernel's layout, rebuilt when its tensors change."""
        fp = _fp(lin)
        hit = self.linears.get(name)
        if hit is None or hit[1] != fp:
            if hit is not None:
                hit[0].close()
            hit = self.linears[name] = (HipLinear.from_module(lin), fp)
            self.sources[name] = (lin,)
        return hit[0]

    def qkv(self, name: str, att) -> HipLinear:
        """q_proj | k_proj | v_proj as ONE [3D, D] layer: x is read once and the three projections are a single launch."""
        mods = (att.q_proj, att.k_proj, att.v_proj)
        fp = tuple(_fp(m) for m in mods)
        hit = self.linears.get(name)
        if hit is None or hit[1] != fp:
            if hit is not None:
                hit[0].close()
            hit = self.linears[name] = (HipLinear(torch.cat([m.weight for m in mods]), torch.cat([m.bias for m in mods])), fp)
            self.sources[name] = mods
        return hit[0]

    def linear_ln(self, name: str, lins, ln: nn.LayerNorm) -> HipLinearLN:
        """LayerNorm `ln` folded into the layer(s) `lins` (one nn.Linear, or several stacked along the outputs like q | k | v): the fused
        single-frame form (HipLinearLN), rebuilt when any source tensor changes."""
        mods = tuple(lins) + (ln,)
        fp = tuple(_fp(m) for m in mods)
        key = name + "+ln"
        hit = self.linears.get(key)
        if hit is None or hit[1] != fp:
            if hit is not None:
                hit[0].close()
            w = torch.cat([m.weight for m in lins]) if len(lins) > 1 else lins[0].weight
            b = torch.cat([m.bias for m in lins]) if len(lins) > 1 else lins[0].bias
            hit = self.linears[key] = (HipLinearLN(w, b, ln.weight, ln.bias, ln.eps), fp)
            self.sources[key] = mods
        return hit[0]

    def fingerprint(self) -> tuple:
        return tuple(_fp(m) for mods in self.sources.values() for m in mods)

    # ---- ops ------------------------------------------------------------------------------------------------------
    @staticmethod
    def layernorm(x: torch.Tensor, ln: nn.LayerNorm, out_split: bool = False) -> torch.Tensor:
        if hip_layernorm_supported(x.shape[-1]) and ln.elementwise_affine and ln.bias is not None:
            return hip_layernorm(x, ln, out_split=out_split)
        y = F.layer_norm(x, (x.shape[-1],), ln.weight, ln.bias, ln.eps)
        return to_split(y) if out_split else y

    @staticmethod
    def layer_supported(layer, L: int) -> bool:
        a = layer.chaos_attention
        if a.q_proj.bias is None or a.k_proj.bias is None or a.v_proj.bias is None or L % 32 != 0:
            return False
        lins = [a.q_proj, a.k_proj, a.v_proj, a.out_proj, layer.ffn[0], layer.ffn[3]]
        return all(hip_linear_supported(m.in_features, m.out_features) for m in lins)

    def _addend_buffer(self, name: str, B: int, D: int, device) -> torch.Tensor:
        add15 = self.addend_bufs.get((name, B))
        if add15 is None or add15.device != device:         # the k|v columns stay zero, the q columns are rewritten per call
            add15 = self.addend_bufs[(name, B)] = torch.zeros(B, 5, 3 * D, device=device)
        return add15

    def chaos_addends(self, names, layers, noise_all: torch.Tensor, B: int, device) -> bool:
        """Every layer's chaos addend (chaos_attention.py:39-66, 85-100) in ONE launch (smk_chaos_addend_batched) into the layers' addend
        buffers; noise_all [num_layers, 3, B, 1].  False (nothing done) when the layers do not share dim / Lorenz constants or are more than 8."""
        from .. import _lib
        atts = [l.chaos_attention for l in layers]
        if not (1 <= len(atts) <= 8) or noise_all.shape[0] != len(atts):
            return False
        D = atts[0].dim
        consts = [(float(a.lorenz_sigma), float(a.lorenz_rho), float(a.lorenz_beta)) if a._lorenz_host is None else a._lorenz_host for a in atts]
        for a, c in zip(atts, consts):
            a._lorenz_host = c
        if any(a.dim != D for a in atts) or any(c != consts[0] for c in consts):
            return False
        n = noise_all.to(device, torch.float32).reshape(len(atts), 3, B).contiguous()
        arr = (_lib.SmkChaosLayer * len(atts))()
        for i, (name, a) in enumerate(zip(names, atts)):
            w, g = a.chaos_proj.weight, a.chaos_gate.weight
            if not (w.is_contiguous() and g.is_contiguous()):
                return False
            buf = self._addend_buffer(name, B, D, device)
            arr[i] = _lib.SmkChaosLayer(n[i].data_ptr(), w.data_ptr(), a.chaos_proj.bias.data_ptr(), g.data_ptr(), a.chaos_gate.bias.data_ptr(),
                                        buf.data_ptr(), buf.shape[2], float(a.chaos_strength))
        import ctypes
        sg, rh, bt = consts[0]
        _lib.check(_lib.load().smk_chaos_addend_batched(len(atts), ctypes.cast(arr, ctypes.c_void_p), B, D, sg, rh, bt, 0.01, _lib.stream_ptr(device)))
        return True

    def layer(self, name: str, layer, x: torch.Tensor, noise: Optional[torch.Tensor] = None, addend_ready: bool = False) -> torch.Tensor:
        """One ChaosTransformerLayer, eval mode, IN PLACE on x [B,L,D] (the residual stream).  noise: the layer's three
        randn(B,1) draws [3,B,1] or None to draw them like the reference does.  addend_ready: chaos_addends() has filled this layer's buffer."""
        B, L, D = x.shape
        att = layer.chaos_attention
        H, d = att.num_heads, att.head_dim
        # Optional (SMK_BODY_SPLIT=1): activations between the kernels travel as split-bf16 pairs (SMK_FMT_SPLIT_BF16, 4 bytes
        # per element like fp32) written by the producers' epilogues, so no linear layer splits its input inside its K loop.
        # Measured neutral on MI355X (the K loop is not bound by the split arithmetic: DESIGN.md 3.3), so it is off by default.
        sp = self.split_activations and D % 8 == 0 and layer.ffn[0].out_features % 8 == 0
        # each LayerNorm runs inside the layer that follows it (HipLinearLN: statistics gathered while the kernel stages the raw rows) -- 12
        # launches and their round trips less per forward, at every batch size.  The gate is the HANDLE's own limit (max_rows)
        fuse_ln = (self.fuse_layernorm and not sp and all(ln.elementwise_affine and ln.bias is not None for ln in (layer.norm1, layer.norm2)))
        add15 = self._addend_buffer(name, B, D, x.device)
        if not addend_ready:
            att.chaos_addend_hip(B, x.device, noise, out=add15)
        qkv_ln = self.linear_ln(name + "chaos_attention.qkv", (att.q_proj, att.k_proj, att.v_proj), layer.norm1) if fuse_ln else None
        kvs = False
        if qkv_ln is not None and B * L <= qkv_ln.max_rows:
            kvs = self.kv_presplit and hip_attention_supported(L, d) and D % 32 == 0
            qkv = qkv_ln.forward_ln(x, periodic_add=add15, rows_per_group=L, split_from=D if kvs else None)
        else:
            h = self.layernorm(x, layer.norm1, out_split=sp)
            qkv = self.qkv(name + "chaos_attention.qkv", att)(h, periodic_add=add15, rows_per_group=L, x_split=sp)
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        scale = 1.0 / (math.sqrt(d) * att.temperature)
        if hip_attention_supported(L, d):
            o = hip_attention(q, k, v, H, scale, out_split=sp, kv_split=kvs)      # [B, L, D]: heads already merged
        else:
            o = F.scaled_dot_product_attention(q.view(B, L, H, d).transpose(1, 2), k.view(B, L, H, d).transpose(1, 2),
                                               v.view(B, L, H, d).transpose(1, 2), scale=scale)
            o = o.transpose(1, 2).reshape(B, L, D)
            o = to_split(o) if sp else o
        self.linear(name + "chaos_attention.out_proj", att.out_proj)(o, residual=x, out=x, x_split=sp)     # x += attn
        ffn_ln = self.linear_ln(name + "ffn.0", (layer.ffn[0],), layer.norm2) if fuse_ln else None
        if ffn_ln is not None and B * L <= ffn_ln.max_rows:
            f = ffn_ln.forward_ln(x, activation="gelu")
        else:
            h = self.layernorm(x, layer.norm2, out_split=sp)
            f = self.linear(name + "ffn.0", layer.ffn[0])(h, activation="gelu", x_split=sp, out_split=sp)
        self.linear(name + "ffn.3", layer.ffn[3])(f, residual=x, out=x, x_split=sp)                        # x += ffn
        return x
