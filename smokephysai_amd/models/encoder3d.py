"""3-D input encoder on MI355X (BASELINE configs[4]: "MFMA conv3d encoder").

The reference's encoder is 2-D (src/models/smokephys_net.py:24-32,87-91); SPEC_3D.md section 8 generalises it axis by axis:
Conv3d(1, 64, 7, padding 3) + BatchNorm3d + ReLU -> Conv3d(64, 128, 3, padding 1) + BatchNorm3d + ReLU -> the two adaptive average
pools with the depth axis pooled to 1 -> features [B, 128, 32, 32], i.e. exactly the tensor SmokePhysNet tokenises (smokephys_net.py:95),
so the rest of the network applies unchanged.  Eval mode (running statistics folded into the weights, as HipEncoder does in 2-D).

Execution (csrc/conv3d_march.hip): both convolutions on split-bf16 MFMAs (fp32-class accuracy), activations channels-last [D, H, W, C].
conv1 (`smk_conv3d_s7_march_forward`): a workgroup marches an 8 x 16 column of voxels along z, weights in registers, each input plane expanded
once into a table of ready MFMA fragments in LDS.  conv2 (`smk_conv3d_cl_zsum_forward`): the same march with three input planes in LDS, all
27 taps read from there, `relu(conv + bias)` summed over z in registers -- the depth half of the pooling -- so the activated conv2 output is
never written; `smk_pool3d_accumulate` reduces the depth sums to the 32 x 32 token sums.
`conv2_mode="implicit"` keeps the first form built (both convolutions as implicit GEMMs on the layer kernel: `smk_conv3d_cl_forward`,
`smk_conv3d_s7_forward`), `"im2col"` the explicit one (`smk_conv3d_im2col` -> [voxels, taps x channels] -> `smk_linear_forward`), for A/B runs.
"""
import torch

from .. import _lib
from .linear import HipLinear

_KEYS = ("conv1_w", "conv1_b", "bn1_w", "bn1_b", "bn1_mean", "bn1_var",
         "conv2_w", "conv2_b", "bn2_w", "bn2_b", "bn2_mean", "bn2_var")


class HipEncoder3D:
    def __init__(self, weights: dict, device="cuda", eps: float = 1e-5, slab_bytes: int = 2 << 30, conv2_mode: str = "march",
                 a2_bytes: int = 128 << 20):
        self._dev = _lib.require_cuda(device, "HipEncoder3D")
        self._L = _lib.load()
        w = {k: torch.as_tensor(weights[k]).detach().to(self._dev, torch.float64) for k in _KEYS}
        if tuple(w["conv1_w"].shape) != (64, 1, 7, 7, 7) or tuple(w["conv2_w"].shape) != (128, 64, 3, 3, 3):
            raise ValueError("HipEncoder3D: weights must be Conv3d(1,64,7) / Conv3d(64,128,3) (SPEC_3D.md section 8)")
        s1 = w["bn1_w"] / torch.sqrt(w["bn1_var"] + eps)
        s2 = w["bn2_w"] / torch.sqrt(w["bn2_var"] + eps)
        # column order of smk_conv3d_im2col: tap * C + c, tap = (kz * k + ky) * k + kx
        w1 = torch.zeros(64, 384, dtype=torch.float64, device=self._dev)
        w1[:, :343] = w["conv1_w"].reshape(64, 343) * s1[:, None]
        w2 = (w["conv2_w"].permute(0, 2, 3, 4, 1).reshape(128, 27 * 64) * s2[:, None]).contiguous()
        b1 = (w["conv1_b"] - w["bn1_mean"]) * s1 + w["bn1_b"]
        b2 = (w["conv2_b"] - w["bn2_mean"]) * s2 + w["bn2_b"]
        self._lin1 = HipLinear(w1.float(), b1.float(), device=self._dev)
        # implicit-GEMM layout of conv1: 56 window rows (kz * 7 + ky; 49 used) x 8 kx slots (7 used)
        w1i = torch.zeros(64, 56, 8, dtype=torch.float64, device=self._dev)
        w1i[:, :49, :7] = (w["conv1_w"].reshape(64, 49, 7) * s1[:, None, None])
        self._lin1i = HipLinear(w1i.reshape(64, 448).float(), b1.float(), device=self._dev)
        # marched form of conv1 (smk_conv3d_s7_march_forward): 7 kz x 8 ky slots x 8 kx slots
        w1m = torch.zeros(64, 7, 8, 8, dtype=torch.float64, device=self._dev)
        w1m[:, :, :7, :7] = w["conv1_w"].reshape(64, 7, 7, 7) * s1[:, None, None, None]
        self._lin1m = HipLinear(w1m.reshape(64, 448).float(), b1.float(), device=self._dev)
        self._lin2 = HipLinear(w2.float(), b2.float(), device=self._dev)
        self.slab_bytes = int(slab_bytes)
        self._bufs = {}
        # the activated conv2 slab is written by the GEMM and read back by the pooling launch right behind it: kept within the 256 MB
        # Infinity Cache it is read from there (512 x 512 planes: one plane per launch; measured 34.4 -> 29.8 ms per 64-plane volume)
        self.a2_bytes = int(a2_bytes)
        if conv2_mode not in ("march", "implicit", "im2col"):
            raise ValueError("conv2_mode: 'march' (smk_conv3d_cl_zsum_forward: conv2 marched along z with its input planes in LDS, depth pooling "
                             "fused), 'implicit' (smk_conv3d_cl_forward: implicit GEMM, no patch matrix) or 'im2col' (explicit GEMM)")
        self.conv2_mode = conv2_mode
        self.conv1_mode = conv2_mode

    def _buffer(self, name, shape):
        """Activation buffers are kept between calls (a 512 x 512 x 64 volume's conv1 output is 4.3 GB: a fresh allocation per call costs
        more than the convolution)."""
        buf = self._bufs.get(name)
        if buf is None or tuple(buf.shape) != tuple(shape):
            self._bufs.pop(name, None)
            buf = self._bufs[name] = torch.empty(*shape, device=self._dev)
        return buf

    def _im2col(self, src, C, D, H, W, k, z0, nz, kpad):
        cols = torch.empty(nz * H * W, kpad, device=self._dev)
        _lib.check(self._L.smk_conv3d_im2col(src.data_ptr(), C, D, H, W, k, z0, nz, cols.data_ptr(), kpad, _lib.stream_ptr(self._dev)))
        return cols

    def conv1_activations(self, vol: torch.Tensor) -> torch.Tensor:
        """One volume [D, H, W] -> relu(bn1(conv1)) channels-last [D, H, W, 64]."""
        D, H, W = vol.shape
        a1 = self._buffer("a1", (D, H, W, 64))
        if self.conv1_mode == "march" and H % 8 == 0 and W % 16 == 0 and H * W * 256 < (1 << 31):
            _lib.check(self._L.smk_conv3d_s7_march_forward(self._lin1m._handle, vol.data_ptr(), D, H, W, a1.data_ptr(), _lib.SMK_ACT_RELU,
                                                           _lib.stream_ptr(self._dev)))
            return a1
        if self.conv1_mode != "im2col" and H <= 1023 and W <= 1023:
            nz = max(1, min(D, ((1 << 32) - 512) // (H * W * 4) - 6, 1000, ((1 << 31) - 512) // (H * W)))
            for z0 in range(0, D, nz):
                n = min(nz, D - z0)
                _lib.check(self._L.smk_conv3d_s7_forward(self._lin1i._handle, vol.data_ptr(), D, H, W, z0, n, a1[z0:z0 + n].data_ptr(), 64,
                                                         _lib.SMK_ACT_RELU, _lib.stream_ptr(self._dev)))
            return a1
        nz = max(1, min(D, self.slab_bytes // (H * W * 384 * 4)))
        for z0 in range(0, D, nz):
            n = min(nz, D - z0)
            cols = self._im2col(vol, 1, D, H, W, 7, z0, n, 384)
            self._lin1(cols, activation="relu", out=a1[z0:z0 + n].view(n * H * W, 64))
        return a1

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, 1, D, H, W] or [B, D, H, W] float32 -> features [B, 128, 32, 32] (H, W multiples of 32)."""
        if x.dim() == 5:
            if x.shape[1] != 1:
                raise ValueError("the 3-D encoder takes one channel")
            x = x[:, 0]
        if x.dim() != 4:
            raise ValueError("x must be [B, 1, D, H, W] or [B, D, H, W]")
        x = x.to(self._dev, torch.float32).contiguous()
        B, D, H, W = x.shape
        for n in (H, W):
            if n % 32 or not (n % 128 == 0 or 128 % n == 0):
                raise ValueError("HipEncoder3D: H and W must be 32, 64 or a multiple of 128 (the two adaptive pools then compose to a "
                                 "uniform block mean)")
        out = torch.empty(B, 128, 32, 32, device=self._dev)
        march = self.conv2_mode == "march" and H * W * 256 < (1 << 31)           # (H % 8, W % 16 hold: multiples of 32)
        implicit = self.conv2_mode == "implicit" or (self.conv2_mode == "march" and not march)
        # implicit GEMM: no patch matrix; a slab is bounded by the activated output it materialises and by 32-bit offsets into a1
        if implicit:
            nz = max(1, min(D, min(self.slab_bytes, self.a2_bytes) // (H * W * 128 * 4), ((1 << 32) - 512) // (H * W * 256) - 2))
        else:
            nz = max(1, min(D, self.slab_bytes // (H * W * 1728 * 4)))
        for b in range(B):
            a1 = self.conv1_activations(x[b])
            sums = torch.zeros(1024, 128, device=self._dev)
            if march:
                zsum = self._buffer("zsum", (H * W, 128))
                _lib.check(self._L.smk_conv3d_cl_zsum_forward(self._lin2._handle, a1.data_ptr(), D, H, W, zsum.data_ptr(), _lib.SMK_ACT_RELU,
                                                              _lib.stream_ptr(self._dev)))
                _lib.check(self._L.smk_pool3d_accumulate(zsum.data_ptr(), 128, H, W, 1, sums.data_ptr(), _lib.stream_ptr(self._dev)))
            for z0 in range(0, D, nz) if not march else ():
                n = min(nz, D - z0)
                if implicit:
                    a2 = self._buffer("a2", (nz * H * W, 128))[:n * H * W]
                    _lib.check(self._L.smk_conv3d_cl_forward(self._lin2._handle, a1.data_ptr(), D, H, W, z0, n, a2.data_ptr(), 128,
                                                             _lib.SMK_ACT_RELU, _lib.stream_ptr(self._dev)))
                else:
                    cols = self._im2col(a1, 64, D, H, W, 3, z0, n, 1728)
                    a2 = self._lin2(cols, activation="relu")                              # [n H W, 128] channels-last
                    del cols
                _lib.check(self._L.smk_pool3d_accumulate(a2.data_ptr(), 128, H, W, n, sums.data_ptr(), _lib.stream_ptr(self._dev)))
                del a2
            out[b] = (sums / float(D * (H // 32) * (W // 32))).t().reshape(128, 32, 32)
        return out

    def tokens(self, x: torch.Tensor) -> torch.Tensor:
        """Same features token-major [B, 1024, 128] (= features.flatten(2).transpose(1, 2), smokephys_net.py:95)."""
        return self(x).flatten(2).transpose(1, 2).contiguous()
