"""Host side of libsmokehip's reconstruction head (smk_decoder_*): SmokePhysNet.reconstruction_head in eval mode as three
direct fp32 kernels with the BatchNorms folded in (smokephys_net.py:57-66,117-118)."""
import ctypes as C

import torch
from torch import nn

from .. import _lib

_KEYS = ("ct1_w", "ct1_b", "bn1_w", "bn1_b", "bn1_mean", "bn1_var", "ct2_w", "ct2_b", "bn2_w", "bn2_b", "bn2_mean", "bn2_var",
         "conv_w", "conv_b")


def decoder_weight_dict(head: nn.Sequential) -> dict:
    """The 14 eval-mode tensors of reconstruction_head (indices 0,1,3,4,6 of the Sequential)."""
    c1, b1, c2, b2, c3 = head[0], head[1], head[3], head[4], head[6]
    return dict(ct1_w=c1.weight, ct1_b=c1.bias, bn1_w=b1.weight, bn1_b=b1.bias, bn1_mean=b1.running_mean, bn1_var=b1.running_var,
                ct2_w=c2.weight, ct2_b=c2.bias, bn2_w=b2.weight, bn2_b=b2.bias, bn2_mean=b2.running_mean, bn2_var=b2.running_var,
                conv_w=c3.weight, conv_b=c3.bias)


def hip_decoder_supported(head: nn.Sequential, S: int) -> bool:
    """Shapes the HIP kernels are built for: the default head (64 -> 32 -> 16 -> 1 channels), token grid side % 16 == 0."""
    try:
        c1, c2, c3 = head[0], head[3], head[6]
        ok = (isinstance(c1, nn.ConvTranspose2d) and isinstance(c2, nn.ConvTranspose2d) and isinstance(c3, nn.Conv2d)
              and (c1.in_channels, c1.out_channels, c2.in_channels, c2.out_channels, c3.in_channels, c3.out_channels) == (64, 32, 32, 16, 16, 1)
              and c1.kernel_size == (4, 4) and c1.stride == (2, 2) and c1.padding == (1, 1) and c1.output_padding == (0, 0)
              and c2.kernel_size == (4, 4) and c2.stride == (2, 2) and c2.padding == (1, 1) and c2.output_padding == (0, 0)
              and c3.kernel_size == (3, 3) and c3.padding == (1, 1) and c1.bias is not None and c2.bias is not None
              and c3.bias is not None and isinstance(head[7], nn.Sigmoid))
    except (IndexError, AttributeError):
        return False
    return ok and S >= 16 and S % 16 == 0


class HipDecoder:
    def __init__(self, weights: dict, device="cuda"):
        self._dev = _lib.require_cuda(device, "HipDecoder")
        self._L = _lib.load()
        ws = {k: torch.as_tensor(weights[k]).detach().to(self._dev, torch.float32).contiguous() for k in _KEYS}
        packed = _lib.SmkDecoderWeights(*[ws[k].data_ptr() for k in _KEYS])
        handle = C.c_void_p()
        _lib.check(self._L.smk_decoder_create(C.byref(packed), self._dev.index, _lib.stream_ptr(self._dev), C.byref(handle)))
        torch.cuda.current_stream(self._dev).synchronize()      # the fold kernel has read `ws`
        self._handle = handle
        self._tmp = {}

    def close(self):
        if getattr(self, "_handle", None):
            self._L.smk_decoder_destroy(self._handle)
            self._handle = None

    __del__ = close

    def __call__(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens [B, S*S, 64] float32 (output_decoder's result) -> reconstructed [B, 1, 4S, 4S]."""
        if tokens.device != self._dev or tokens.dtype != torch.float32 or tokens.dim() != 3 or tokens.shape[2] != 64:
            raise ValueError(f"HipDecoder: tokens must be float32 [B, S*S, 64] on {self._dev}")
        B, L, _ = tokens.shape
        S = int(round(L ** 0.5))
        if S * S != L:
            raise ValueError("HipDecoder: the token count must be a square")
        tokens = tokens.contiguous()
        tmp = self._tmp.get(B * 4096 + S)
        if tmp is None:                                          # scratch, reused across calls (stream-ordered)
            tmp = self._tmp[B * 4096 + S] = (torch.empty(B, 32, 2 * S, 2 * S, device=self._dev),
                                             torch.empty(B, 16, 4 * S, 4 * S, device=self._dev))
        out = torch.empty(B, 1, 4 * S, 4 * S, device=self._dev)
        _lib.check(self._L.smk_decoder_forward(self._handle, tokens.data_ptr(), B, S, tmp[0].data_ptr(), tmp[1].data_ptr(),
                                               out.data_ptr(), _lib.stream_ptr(self._dev)))
        return out
