"""HIP front-end of SmokePhysNet.input_encoder + pooling (src/models/smokephys_net.py:24-32,87-91).

`HipEncoder` folds eval-mode BatchNorm into per-channel scale/shift once, re-lays the conv weights out for the
MFMA kernel (csrc/encoder.hip) and maps frames [B,1,H,W] -> features [B,128,32,32] in one fused launch.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib

_KEYS = ("conv1_w", "conv1_b", "bn1_w", "bn1_b", "bn1_mean", "bn1_var",
         "conv2_w", "conv2_b", "bn2_w", "bn2_b", "bn2_mean", "bn2_var")


def encoder_weight_dict(input_encoder: nn.Sequential) -> dict:
    """The 12 eval-mode tensors of input_encoder (indices 0,1,3,4 of the Sequential, smokephys_net.py:24-32)."""
    c1, b1, c2, b2 = input_encoder[0], input_encoder[1], input_encoder[3], input_encoder[4]
    return dict(conv1_w=c1.weight, conv1_b=c1.bias, bn1_w=b1.weight, bn1_b=b1.bias, bn1_mean=b1.running_mean,
                bn1_var=b1.running_var, conv2_w=c2.weight, conv2_b=c2.bias, bn2_w=b2.weight, bn2_b=b2.bias,
                bn2_mean=b2.running_mean, bn2_var=b2.running_var)


class HipEncoder:
    def __init__(self, weights: dict, device="cuda"):
        self._dev = _lib.require_cuda(device, "HipEncoder")
        self._L = _lib.load()
        self._handle = None
        self.load_weights(weights)

    def load_weights(self, weights: dict):
        ws = {k: torch.as_tensor(weights[k]).detach().to(self._dev, torch.float32).contiguous() for k in _KEYS}
        if ws["conv1_w"].numel() != 64 * 49 or ws["conv2_w"].numel() != 128 * 64 * 9:
            raise ValueError("input_encoder must be Conv2d(1,64,7) / Conv2d(64,128,3) (smokephys_net.py:25,28)")
        packed = _lib.SmkEncoderWeights(*[ws[k].data_ptr() for k in _KEYS])
        handle = C.c_void_p()
        _lib.check(self._L.smk_encoder_create(C.byref(packed), self._dev.index, _lib.stream_ptr(self._dev), C.byref(handle)))
        torch.cuda.current_stream(self._dev).synchronize()      # fold kernel has read `ws`
        self.close()
        self._handle = handle

    def close(self):
        if getattr(self, "_handle", None):
            self._L.smk_encoder_destroy(self._handle)
            self._handle = None

    __del__ = close

    def _frames(self, x):
        if x.dim() == 4:
            if x.shape[1] != 1:
                raise ValueError("input_encoder takes one channel (smokephys_net.py:25)")
            x = x[:, 0]
        x = x.to(self._dev, torch.float32)
        if x.stride(2) != 1 or x.stride(1) != x.shape[2]:
            x = x.contiguous()
        return x

    def __call__(self, x: torch.Tensor, input_dim: int = 128, dtype: str = "f32") -> torch.Tensor:
        x = self._frames(x)
        B, H, W = x.shape
        out = torch.empty(B, 128, 32, 32, device=self._dev)
        _lib.check(self._L.smk_encoder_forward(self._handle, x.data_ptr(), x.stride(0), B, H, W, int(input_dim),
                                               out.data_ptr(), _lib.DTYPES[dtype], _lib.stream_ptr(self._dev)))
        return out

    def tokens(self, x: torch.Tensor, input_dim: int = 128, dtype: str = "bf16x3") -> torch.Tensor:
        """Same features, token-major [B, 1024, 128] (= encoded.flatten(2).transpose(1, 2), smokephys_net.py:95)."""
        x = self._frames(x)
        B, H, W = x.shape
        out = torch.empty(B, 1024, 128, device=self._dev)
        _lib.check(self._L.smk_encoder_forward_tokens(self._handle, x.data_ptr(), x.stride(0), B, H, W, int(input_dim),
                                                      out.data_ptr(), _lib.DTYPES[dtype], _lib.stream_ptr(self._dev)))
        return out

    def conv1_activations(self, x: torch.Tensor) -> torch.Tensor:
        x = self._frames(x)
        B, H, W = x.shape
        out = torch.empty(B, 64, H, W, device=self._dev)
        _lib.check(self._L.smk_encoder_conv1(self._handle, x.data_ptr(), x.stride(0), B, H, W, out.data_ptr(),
                                             _lib.stream_ptr(self._dev)))
        return out
