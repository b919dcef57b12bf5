"""Chaos-aware attention -- drop-in for src/models/chaos_attention.py:6-114 (same parameter/buffer names).

The reference forms two [B,heads,L,L] score tensors, QK^T/sqrt(d) and strength*gate*(C_h K^T)/sqrt(d), and adds them
(chaos_attention.py:82-100).  Both are linear in the left operand, so  S = ((Q + strength*gate*C_h) K^T)/sqrt(d)
exactly; the chaos term is folded into Q and one fused scaled-dot-product attention is used.
The Lorenz chaos field has period 5 along the sequence (chaos_attention.py:61-65), so it is projected once per
distinct row and tiled.
"""
import math
import os
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .attention import hip_attention_supported, hip_attention_train, hip_qkv_attention_train
from .linear import HipLinear, TrainableHipLinear, hip_linear_supported, hip_linear_wgrad_supported


class ChaosAttention(nn.Module):
    def __init__(self, dim: int, num_heads: int = 8, chaos_strength: float = 0.1, temperature: float = 1.0):
        super().__init__()
        self.dim = dim
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.chaos_strength = chaos_strength
        self.temperature = temperature
        assert dim % num_heads == 0
        self.q_proj = TrainableHipLinear(dim, dim)
        self.k_proj = TrainableHipLinear(dim, dim)
        self.v_proj = TrainableHipLinear(dim, dim)
        self.out_proj = TrainableHipLinear(dim, dim)
        self.chaos_proj = nn.Linear(3, dim)
        self.chaos_gate = nn.Linear(dim, 1)
        self.register_buffer("lorenz_sigma", torch.tensor(10.0))
        self.register_buffer("lorenz_rho", torch.tensor(28.0))
        self.register_buffer("lorenz_beta", torch.tensor(8.0 / 3.0))
        self._lorenz_host = None
        self.hip_train = False      # set by SmokePhysNet(linear_dtype="bf16x3"): attention forward + backward on libsmokehip in training

    # ---- training: q | k | v as one [3D, D] layer (device mirrors of the concatenated weights, re-split when a parameter changes)
    def __getstate__(self):
        d = self.__dict__.copy()
        for k in ("_hip_qkv", "_hip_qkv_fp"):
            d.pop(k, None)
        return d

    def _hip_qkv_handles(self):
        mods = (self.q_proj, self.k_proj, self.v_proj)
        fp = tuple((m.weight.data_ptr(), m.weight._version, m.bias.data_ptr(), m.bias._version) for m in mods)
        h = self.__dict__.get("_hip_qkv")
        if h is None or self.__dict__.get("_hip_qkv_fp") != fp:
            w = torch.cat([m.weight.detach() for m in mods])                    # [3D, D]
            b = torch.cat([m.bias.detach() for m in mods])
            if h is None:
                h = (HipLinear(w, b), HipLinear(w.t().contiguous(), None))      # forward; dX = dY W (out = D, in = 3D)
                self.__dict__["_hip_qkv"] = h
            else:
                h[0].update(w, b)
                h[1].update(w, None, transposed=True)
            self.__dict__["_hip_qkv_fp"] = fp
        return h

    def _hip_qkv_ok(self, x: torch.Tensor) -> bool:
        D = self.dim
        return (all(m.bias is not None and m.weight.requires_grad == self.q_proj.weight.requires_grad for m in (self.q_proj, self.k_proj, self.v_proj))
                and hip_linear_supported(D, 3 * D) and hip_linear_supported(3 * D, D) and hip_linear_wgrad_supported(D, 3 * D)
                and x.shape[1] % 32 == 0 and os.environ.get("SMK_TRAIN_QKV_FUSED", "1") == "1")

    def lorenz_system(self, x, y, z, dt: float = 0.01):
        """chaos_attention.py:39-45 (explicit Euler)."""
        dx = self.lorenz_sigma * (y - x)
        dy = x * (self.lorenz_rho - z) - y
        dz = x * y - self.lorenz_beta * z
        return (x + dt * dx, y + dt * dy, z + dt * dz)

    def chaos_states(self, batch_size: int, device, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The 5 Lorenz states [B,5,3] of generate_chaos_field (chaos_attention.py:47-59).
        noise: the three randn(B,1) draws stacked as [3,B,1]; None draws them like the reference does."""
        if noise is None:
            x = torch.randn(batch_size, 1, device=device) * 0.1
            y = torch.randn(batch_size, 1, device=device) * 0.1
            z = torch.randn(batch_size, 1, device=device) * 0.1
        else:
            noise = noise.to(device)
            x, y, z = noise[0] * 0.1, noise[1] * 0.1, noise[2] * 0.1
        seq = []
        for _ in range(5):
            x, y, z = self.lorenz_system(x, y, z)
            seq.append(torch.cat([x, y, z], dim=-1))
        return torch.stack(seq, dim=1)

    def chaos_states_hip(self, batch_size: int, device, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """chaos_states as ONE libsmokehip launch (smk_lorenz_states): the reference's three randn(B,1) draws in its order, then the five
        Euler steps on the device -- the states carry no gradient (noise and the Lorenz constants are not parameters)."""
        from .. import _lib
        dev = _lib.require_cuda(device, "ChaosAttention.chaos_states_hip")
        if noise is None:
            noise = torch.empty(3, batch_size, 1, device=dev, dtype=torch.float32)
            for i in range(3):
                torch.randn(batch_size, 1, device=dev, out=noise[i])
        n3 = noise.detach().to(dev, torch.float32).reshape(3, batch_size).contiguous()
        if self._lorenz_host is None:
            self._lorenz_host = (float(self.lorenz_sigma), float(self.lorenz_rho), float(self.lorenz_beta))
        sg, rh, bt = self._lorenz_host
        states = torch.empty(batch_size, 5, 3, device=dev, dtype=torch.float32)
        _lib.check(_lib.load().smk_lorenz_states(n3.data_ptr(), batch_size, sg, rh, bt, 0.01, states.data_ptr(), _lib.stream_ptr(dev)))
        return states

    def generate_chaos_field(self, seq_len: int, batch_size: int, device, noise=None) -> torch.Tensor:
        """chaos_attention.py:47-66."""
        field = self.chaos_states(batch_size, device, noise)
        reps = (seq_len + field.size(1) - 1) // field.size(1)
        return field.repeat(1, reps, 1)[:, :seq_len, :]

    def chaos_addend(self, batch_size: int, device, dtype, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The chaos term folded into Q, on the 5 distinct rows of the tiled Lorenz field: chaos_strength * gate(C) * C with
        C = chaos_proj(field) (chaos_attention.py:85-100).  [B,5,D]; row l of the sequence receives row l % 5."""
        if self.hip_train and torch.device(device).type == "cuda" and dtype == torch.float32 and torch.is_grad_enabled():
            states = self.chaos_states_hip(batch_size, device, noise)            # one launch instead of ~75 one-element launches
        else:
            states = self.chaos_states(batch_size, device, noise).to(dtype)
        c5 = self.chaos_proj(states)
        return self.chaos_strength * torch.sigmoid(self.chaos_gate(c5)) * c5

    def chaos_addend_hip(self, batch_size: int, device, noise: Optional[torch.Tensor] = None,
                         out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """chaos_addend as ONE libsmokehip launch (smk_chaos_addend) instead of ~90 elementwise launches; the three
        randn(B,1) draws are made exactly like the reference makes them (same generator calls, same order)."""
        from .. import _lib
        dev = _lib.require_cuda(device, "ChaosAttention.chaos_addend_hip")
        if noise is None:
            # the reference's three randn(B,1) generator calls, in its order, written straight into one [3,B,1] buffer
            # (torch.stack of three tiny tensors costs three device memcpys per layer)
            noise = torch.empty(3, batch_size, 1, device=dev, dtype=torch.float32)
            for i in range(3):
                torch.randn(batch_size, 1, device=dev, out=noise[i])
        n3 = noise.to(dev, torch.float32).reshape(3, batch_size).contiguous()
        if out is None:                      # out: [B, 5, >= dim] float32, columns 0..dim-1 are written
            out = torch.empty(batch_size, 5, self.dim, device=dev, dtype=torch.float32)
        elif out.shape[:2] != (batch_size, 5) or out.shape[2] < self.dim or not out.is_contiguous() or out.dtype != torch.float32:
            raise ValueError("chaos_addend_hip: out must be a contiguous float32 [B, 5, >= dim] tensor")
        if self._lorenz_host is None:        # buffers are constants of the model; read them once (no sync per forward)
            self._lorenz_host = (float(self.lorenz_sigma), float(self.lorenz_rho), float(self.lorenz_beta))
        sg, rh, bt = self._lorenz_host
        w = self.chaos_proj.weight
        if not (w.is_contiguous() and self.chaos_gate.weight.is_contiguous()):
            raise ValueError("chaos_proj / chaos_gate weights must be contiguous")
        _lib.check(_lib.load().smk_chaos_addend(n3.data_ptr(), batch_size, self.dim, w.data_ptr(),
                                               self.chaos_proj.bias.data_ptr(), self.chaos_gate.weight.data_ptr(),
                                               self.chaos_gate.bias.data_ptr(), float(self.chaos_strength), sg, rh, bt,
                                               0.01, out.data_ptr(), out.shape[2], _lib.stream_ptr(dev)))
        return out

    def forward(self, x: torch.Tensor, mask: torch.Tensor = None, noise: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """residual (optional): returns attention(x) + residual (the block's `x + attn(norm1(x))`, smokephys_net.py:161-163) with the add
        in out_proj's epilogue."""
        B, L, D = x.shape
        H, d = self.num_heads, self.head_dim
        if (self.hip_train and mask is None and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
                and hip_attention_supported(L, d) and self._hip_qkv_ok(x)):
            # training on a ROCm device: q | k | v projection (+ chaos addend on the q columns) and the flash attention as one autograd node
            add5 = self.chaos_addend(B, x.device, x.dtype, noise)                    # [B,5,D]: through chaos_proj / chaos_gate (autograd)
            out = hip_qkv_attention_train(self, x, add5, H, 1.0 / (math.sqrt(d) * self.temperature))
            return self.out_proj(out, residual=residual)
        q = self.q_proj(x)
        add5 = self.chaos_addend(B, x.device, x.dtype, noise)                        # [B,5,D]
        reps = (L + 4) // 5
        q = q + add5.repeat(1, reps, 1)[:, :L]
        scale = 1.0 / (math.sqrt(d) * self.temperature)
        if (self.hip_train and mask is None and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
                and hip_attention_supported(L, d)):
            # training on a ROCm device: flash attention forward + backward on libsmokehip, token-major in and out
            # (no head transposes, no merge-heads copy)
            out = hip_attention_train(q, self.k_proj(x), self.v_proj(x), H, scale)
            return self.out_proj(out, residual=residual)
        k = self.k_proj(x).view(B, L, H, d).transpose(1, 2)
        v = self.v_proj(x).view(B, L, H, d).transpose(1, 2)
        q = q.view(B, L, H, d).transpose(1, 2)
        attn_mask = None
        if mask is not None:
            attn_mask = (mask != 0)[:, None, None, :]
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, scale=scale)
        out = out.transpose(1, 2).contiguous().view(B, L, D)
        return self.out_proj(out, residual=residual)
