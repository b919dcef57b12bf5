"""SmokePhysNet on MI355X -- drop-in for src/models/smokephys_net.py:7-167 (same constructor, forward contract and
state_dict keys, so a reference best_model.pth loads unchanged).

Inference (eval mode): frames go through the fused HIP encoder (csrc/encoder.hip: conv7x7+BN+ReLU, conv3x3+BN+ReLU
on MFMA, both adaptive pools as one block mean) -- there is no CPU fallback for it.  The token-wise linear layers of the
transformer body (feature_proj, q/k/v/out projections, FFN, output_decoder) run on libsmokehip's split-bf16 MFMA kernel
(csrc/linear.hip) with bias / pos-embedding / chaos-term / GELU / residual fused into the GEMM epilogue, the softmax
attention on its split-bf16 flash kernel (csrc/transformer.hip, chaos term folded into Q), LayerNorm and the conv
reconstruction head (csrc/decoder.hip, BatchNorms folded) on their own kernels; only the 3-element physics head and the
token mean are PyTorch-ROCm ops.
Training (train mode): the encoder runs as autograd-tracked PyTorch ops with batch-statistics BatchNorm, exactly the
reference's op sequence (smokephys_net.py:87-91).
"""
import os
import warnings
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .attention import hip_layernorm_supported, hip_layernorm_train
from .chaos_attention import ChaosAttention
from .decoder import HipDecoder, decoder_weight_dict, hip_decoder_supported
from .encoder import HipEncoder, encoder_weight_dict
from .hip_body import HipBody
from .ffn import hip_dropout_add, hip_ffn_elementwise_supported, hip_gelu_dropout
from .linear import TrainableHipLinear, hip_linear_supported
from .conv import hip_conv1_train, hip_conv1_train_supported, hip_conv2_train, hip_conv2_train_supported
from .norm import hip_bn_relu_pool, hip_sync_bn_relu_pool
from .sync_bn import SyncBatchNorm2d
from .physics_regularizer import PhysicsRegularizer


def _conv_block(cin: int, cout: int, k: int):
    """Conv(k, same padding) + BatchNorm + ReLU (smokephys_net.py:25-30)."""
    return [nn.Conv2d(cin, cout, k, padding=k // 2), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]


def _upsample_block(cin: int, cout: int):
    """Stride-2 transposed conv (k4, p1) + BatchNorm + ReLU (smokephys_net.py:58-63)."""
    return [nn.ConvTranspose2d(cin, cout, 4, stride=2, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]


def _avg_pool_to(t: torch.Tensor, size) -> torch.Tensor:
    """F.adaptive_avg_pool2d(t, size).  When the input is a whole multiple of `size` (256 -> 128 -> 32: every training shape of
    the path) the windows are the fixed k x k blocks of F.avg_pool2d -- same means, and a backward that is a plain broadcast
    instead of adaptive pooling's atomic scatter (5 ms per step at batch 64)."""
    size = (size, size) if isinstance(size, int) else tuple(size)
    h, w = t.shape[-2:]
    if h % size[0] == 0 and w % size[1] == 0:
        k = (h // size[0], w // size[1])
        return t if k == (1, 1) else F.avg_pool2d(t, k)
    return F.adaptive_avg_pool2d(t, size)


def _hip_bn_ok(bn) -> bool:
    """The fused libsmokehip BatchNorm + ReLU + pool block implements exactly a training-mode affine nn.BatchNorm2d (or this package's
    SyncBatchNorm2d: the same passes with the statistics all-reduced in between) with running statistics and a fixed momentum.
    Anything else -- torch.nn.SyncBatchNorm, a frozen block (bn.eval() inside model.train()), momentum=None (cumulative average),
    track_running_stats=False, affine=False -- runs the PyTorch modules."""
    return (type(bn) in (nn.BatchNorm2d, SyncBatchNorm2d) and bn.training and bn.affine and bn.track_running_stats
            and bn.momentum is not None)


# SMK_TRAIN_CONV2_HIP: "1" (default) conv2's forward and both gradients on libsmokehip; "grads" the gradients only; "dgrad" the data gradient only;
# "0" the whole convolution on PyTorch-ROCm (diagnostic switches)
_HIP_CONV2_TRAIN = os.environ.get("SMK_TRAIN_CONV2_HIP", "1")
_HIP_CONV1_TRAIN = os.environ.get("SMK_TRAIN_CONV1_HIP", "1") != "0"      # diagnostic: 0 keeps the first convolution on PyTorch-ROCm


def _bn_relu_pool(z, bn, pool):
    return hip_sync_bn_relu_pool(z, bn, pool) if isinstance(bn, SyncBatchNorm2d) else hip_bn_relu_pool(z, bn, pool)


def _mlp(din: int, dhid: int, dout: int, linear=nn.Linear) -> nn.Sequential:
    return nn.Sequential(linear(din, dhid), nn.ReLU(inplace=True), linear(dhid, dout))


class SmokePhysNet(nn.Module):
    def __init__(self, input_dim: int = 128, hidden_dim: int = 512, num_layers: int = 6, num_heads: int = 8,
                 output_channels: int = 64, chaos_strength: float = 0.1, encoder_dtype: str = "bf16x3",
                 linear_dtype: str = "bf16x3"):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.encoder_dtype = encoder_dtype
        if linear_dtype not in ("bf16x3", "f32"):
            raise ValueError("linear_dtype: 'bf16x3' (libsmokehip split-bf16 MFMA kernel) or 'f32' (PyTorch-ROCm GEMMs)")
        self.linear_dtype = linear_dtype
        # Construction order and Sequential indices follow the reference exactly: that fixes both the state_dict keys
        # (input_encoder.{0,1,3,4}, reconstruction_head.{0,1,3,4,6}, ...) and the RNG stream of the default initialisation.
        self.input_encoder = nn.Sequential(*_conv_block(1, 64, 7), *_conv_block(64, 128, 3),
                                           nn.AdaptiveAvgPool2d((input_dim, input_dim)))
        self.pos_embedding = nn.Parameter(torch.randn(1, input_dim * input_dim, hidden_dim))
        self.feature_proj = TrainableHipLinear(128, hidden_dim)
        self.chaos_layers = nn.ModuleList(
            ChaosTransformerLayer(hidden_dim, num_heads, chaos_strength=chaos_strength) for _ in range(num_layers))
        self.output_decoder = _mlp(hidden_dim, 256, output_channels, linear=TrainableHipLinear)
        self.reconstruction_head = nn.Sequential(*_upsample_block(output_channels, 32), *_upsample_block(32, 16),
                                                 nn.Conv2d(16, 1, 3, padding=1), nn.Sigmoid())
        self.physics_head = _mlp(hidden_dim, 256, 3)
        self.physics_regularizer = PhysicsRegularizer()
        self._hip = None          # (HipEncoder, weight fingerprint)
        self._pos_cache = None    # (fingerprint, tensor)
        self._hip_body = HipBody()   # libsmokehip mirrors of the token-wise linear layers + per-layer scratch
        self._hip_dec = None         # (HipDecoder, weight fingerprint)
        # training: the token-wise linear layers run their forward and input-gradient GEMMs on libsmokehip as well (models/linear.py)
        for m in self.modules():
            if isinstance(m, (TrainableHipLinear, ChaosAttention, ChaosTransformerLayer)):
                m.hip_train = linear_dtype == "bf16x3"

    # copy.deepcopy / pickling of the module: the libsmokehip handles are per-instance device mirrors, rebuilt on first use
    def __getstate__(self):
        d = self.__dict__.copy()
        d.update(_hip=None, _pos_cache=None, _hip_body=None, _hip_dec=None)
        return d

    def __setstate__(self, state):
        super().__setstate__(state)
        if self.__dict__.get("_hip_body") is None:
            self._hip_body = HipBody()

    # ---- HIP encoder plumbing -------------------------------------------------------------------------------
    def _encoder_fingerprint(self):
        ws = encoder_weight_dict(self.input_encoder)
        return tuple((t.data_ptr(), t._version) for t in ws.values())

    def hip_encoder(self) -> HipEncoder:
        """Folded-weight HIP encoder for the current input_encoder tensors (rebuilt when they change)."""
        dev = self.pos_embedding.device
        fp = self._encoder_fingerprint()
        if self._hip is None or self._hip[1] != fp:
            if self._hip is not None:
                self._hip[0].close()
            self._hip = (HipEncoder(encoder_weight_dict(self.input_encoder), device=dev), fp)
        return self._hip[0]

    def _encoder_route(self, x: torch.Tensor) -> str:
        """Which implementation of input_encoder + pools serves this call:
        'train'   -- module in train mode: batch statistics (and their running update), as nn.BatchNorm2d does with or without grad;
        'modules' -- eval mode, but a gradient is wanted through the encoder (the fused kernel is forward-only), or a frame shape the
                     fused kernel is not built for on a ROCm device (a warning is issued once);
        'hip'     -- the fused libsmokehip encoder (eval, no gradient; raises off-GPU: no CPU fallback)."""
        if self.training:
            return "train"
        if not x.is_cuda:
            return "hip"                           # eval off-GPU: the product refuses (no CPU fallback)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.input_encoder.parameters())):
            if not self.__dict__.get("_warned_grad"):
                self.__dict__["_warned_grad"] = True
                warnings.warn("SmokePhysNet: eval forward with autograd enabled runs the differentiable PyTorch-ROCm route; wrap inference in "
                              "torch.no_grad() to use the fused libsmokehip kernels", stacklevel=3)
            return "modules"
        H, W = x.shape[-2:]
        ok = H == W and H in (64, 128, 256) and self.input_dim % 32 == 0 and (self.input_dim % H == 0 or H % self.input_dim == 0)
        if not ok:
            if not self.__dict__.get("_warned_shape"):
                self.__dict__["_warned_shape"] = True
                warnings.warn(f"SmokePhysNet: frames of {H}x{W} are outside the fused HIP encoder's shapes (square 64/128/256); "
                              "this call runs input_encoder on PyTorch-ROCm ops", stacklevel=3)
            return "modules"
        return "hip"

    def encode_frames(self, x: torch.Tensor, dtype: Optional[str] = None) -> torch.Tensor:
        """input_encoder + both pools (smokephys_net.py:87-91): [B,1,H,W] -> [B,128,32,32]."""
        route = self._encoder_route(x)
        if route == "modules":                     # eval-mode BatchNorm through the PyTorch-ROCm modules (differentiable; any frame size)
            return F.adaptive_avg_pool2d(self.input_encoder(x), (32, 32))
        if route == "train":
            conv1, bn1, _, conv2, bn2, _, pool = self.input_encoder
            H, W = x.shape[-2:]
            P = H // 32
            mid = pool.output_size[0]              # first pool target; no up-sampling stage and whole-number windows: one P x P mean
            if (self.linear_dtype == "bf16x3" and x.is_cuda and x.dtype == torch.float32 and H == W and P in (4, 8) and W == 32 * P
                    and _hip_bn_ok(bn1) and _hip_bn_ok(bn2)
                    and pool.output_size[0] == pool.output_size[1] and H % mid == 0 and mid % 32 == 0):
                # libsmokehip: BatchNorm (batch statistics) + ReLU as two passes over the conv output, and for the second block
                # the two average pools (one P x P block mean) in the same pass -- the 256 x 256 x 128 maps are never written
                z1 = hip_conv1_train(x, conv1) if (_HIP_CONV1_TRAIN and hip_conv1_train_supported(x, conv1)) else conv1(x)
                a1 = _bn_relu_pool(z1, bn1, 1)
                z2 = (hip_conv2_train(a1, conv2, hip_forward=_HIP_CONV2_TRAIN not in ("grads", "dgrad"), hip_wgrad=_HIP_CONV2_TRAIN != "dgrad")
                      if (_HIP_CONV2_TRAIN != "0" and hip_conv2_train_supported(a1, conv2)) else conv2(a1))
                return _bn_relu_pool(z2, bn2, P)
            encoded = x
            for m in self.input_encoder:
                encoded = _avg_pool_to(encoded, m.output_size) if isinstance(m, nn.AdaptiveAvgPool2d) else m(encoded)
            return _avg_pool_to(encoded, (32, 32))
        return self.hip_encoder()(x, input_dim=self.input_dim, dtype=dtype or self.encoder_dtype)

    def _pos_embed(self, pool_size: int) -> torch.Tensor:
        """smokephys_net.py:99-105; the bilinear resize is input-independent, so it is cached in eval."""
        expected = pool_size * pool_size
        if expected == self.pos_embedding.shape[1]:
            return self.pos_embedding
        fp = (self.pos_embedding.data_ptr(), self.pos_embedding._version, pool_size)
        if not self.training and self._pos_cache is not None and self._pos_cache[0] == fp:
            return self._pos_cache[1]
        pe = self.pos_embedding.reshape(1, self.input_dim, self.input_dim, self.hidden_dim).permute(0, 3, 1, 2)
        pe = F.interpolate(pe, size=(pool_size, pool_size), mode="bilinear", align_corners=False)
        pe = pe.permute(0, 2, 3, 1).reshape(1, expected, self.hidden_dim)
        if not self.training:
            self._pos_cache = (fp, pe.detach())
        return pe

    # ---- transformer body on libsmokehip (eval only; per-layer code: hip_body.py) ----------------------------------
    def _decoder_fingerprint(self):
        return tuple((t.data_ptr(), t._version) for t in decoder_weight_dict(self.reconstruction_head).values())

    def hip_decoder(self) -> HipDecoder:
        """BN-folded HIP reconstruction head for the current reconstruction_head tensors (rebuilt when they change)."""
        fp = self._decoder_fingerprint()
        if self._hip_dec is None or self._hip_dec[1] != fp:
            if self._hip_dec is not None:
                self._hip_dec[0].close()
            self._hip_dec = (HipDecoder(decoder_weight_dict(self.reconstruction_head), device=self.pos_embedding.device), fp)
        return self._hip_dec[0]

    def hip_weights_fingerprint(self):
        """Identity + version of every parameter and buffer of the module, plus the mirror epoch -- a function of the SOURCE
        tensors only (not of which lazily built libsmokehip mirrors exist yet), so it is the same before and after the first
        forward; GraphedSmokePhysNet re-captures when it changes."""
        import itertools
        return (self.__dict__.get("_mirror_epoch", 0),) + tuple(
            (t.data_ptr(), t._version) for t in itertools.chain(self.parameters(), self.buffers()))

    def invalidate_hip_mirrors(self) -> None:
        """Drop every re-laid-out device copy libsmokehip keeps of this module's weights (folded encoder / decoder weights, split
        linear weights, resized pos-embedding; training handles of the token-wise linears).  The mirrors are keyed on
        (data_ptr, _version) of their source tensors, which a write through `param.data` (EMA updates, `p.data.clamp_()`,
        `p.data.copy_()`) does NOT bump: call this after such a write, or write with `with torch.no_grad(): p.copy_(...)`."""
        if self._hip is not None:
            self._hip[0].close()
        if self._hip_dec is not None:
            self._hip_dec[0].close()
        self._hip = self._hip_dec = self._pos_cache = None
        body = self._hip_body
        for h, _ in body.linears.values():
            h.close()
        body.linears.clear(); body.sources.clear()
        for m in self.modules():
            if isinstance(m, ChaosAttention):
                for h in m.__dict__.pop("_hip_qkv", None) or ():
                    h.close()
                m.__dict__.pop("_hip_qkv_fp", None)
            if isinstance(m, TrainableHipLinear):
                for k in ("_hip_fwd", "_hip_bwd"):
                    h = m.__dict__.pop(k, None)
                    if h is not None:
                        h.close()
                m.__dict__.pop("_hip_fwd_fp", None); m.__dict__.pop("_hip_bwd_fp", None)
        self.__dict__["_mirror_epoch"] = self.__dict__.get("_mirror_epoch", 0) + 1

    def _body_hip(self, tokens: torch.Tensor, chaos_noise: Optional[torch.Tensor], pool_size: int):
        """feature_proj + pos-embed, the pre-LN chaos transformer layers and output_decoder (smokephys_net.py:95-114,
        136-168; chaos_attention.py:68-114) with every token-wise nn.Linear as one fused libsmokehip launch:
        bias, the pos-embedding / chaos-term addend, GELU / ReLU and the residual add ride in the GEMM epilogue;
        softmax attention on the flash kernel (smk_attention, chaos term folded into Q), the per-layer Lorenz /
        chaos-gate chain on smk_chaos_addend, LayerNorm on smk_layernorm.  Returns (features [B,L,D], decoded [B,L,C])."""
        B, L, _ = tokens.shape
        body = self._hip_body
        if chaos_noise is None:
            # every layer's three randn(B, 1) draws (chaos_attention.py:50-52) as ONE generator launch: drawn per layer they are 18 launches
            # plus 18 device copies per forward (randn into a view goes through a temporary) -- 0.11 ms of a 0.77 ms batch-1 forward.  Still
            # standard-normal and fresh per call; only the position in the generator's stream differs from per-layer draws.
            chaos_noise = torch.randn(len(self.chaos_layers), 3, B, 1, device=tokens.device, dtype=torch.float32)
        x = body.linear("feature_proj", self.feature_proj)(tokens, periodic_add=self._pos_embed(pool_size),
                                                           rows_per_group=B * L)
        names = [f"chaos_layers.{li}." for li in range(len(self.chaos_layers))]
        ready = body.chaos_addends(names, list(self.chaos_layers), chaos_noise, B, tokens.device)        # all layers' chaos terms: one launch
        for li, layer in enumerate(self.chaos_layers):
            body.layer(names[li], layer, x, chaos_noise[li], addend_ready=ready)
        dec = body.linear("output_decoder.0", self.output_decoder[0])(x, activation="relu")
        dec = body.linear("output_decoder.2", self.output_decoder[2])(dec)
        return x, dec

    def _hip_tail_ok(self, features: torch.Tensor) -> bool:
        h = self.physics_head
        return (self.linear_dtype == "bf16x3" and not self.training and not torch.is_grad_enabled() and features.is_cuda
                and features.dtype == torch.float32 and features.dim() == 3 and features.stride(2) == 1
                and features.stride(0) == features.shape[1] * features.stride(1) and features.shape[0] <= 65535
                and len(h) == 3 and type(h[0]) is nn.Linear and type(h[1]) is nn.ReLU and type(h[2]) is nn.Linear
                and h[0].bias is not None and h[2].bias is not None and h[0].in_features == features.shape[2]
                and h[0].in_features + h[0].out_features <= 16384 and h[0].in_features % 4 == 0 and h[0].weight.data_ptr() % 16 == 0
                and all(p.is_contiguous() and p.dtype == torch.float32 and p.device == features.device for p in h.parameters()))

    def _tail_hip(self, features: torch.Tensor):
        """smokephys_net.py:116-118 -- latent = features.mean(dim=1); physics = physics_head(latent) -- as smk_pooled_head (fp32)."""
        from .. import _lib
        B, L, D = features.shape
        l1, l2 = self.physics_head[0], self.physics_head[2]
        pooled = torch.empty(B, D, device=features.device, dtype=torch.float32)
        out = torch.empty(B, l2.out_features, device=features.device, dtype=torch.float32)
        ws = torch.empty(B * (32 * D + l1.out_features), device=features.device, dtype=torch.float32)
        _lib.check(_lib.load().smk_pooled_head(features.data_ptr(), B, L, D, features.stride(1), l1.weight.data_ptr(), l1.bias.data_ptr(),
                                               l1.out_features, l2.weight.data_ptr(), l2.bias.data_ptr(), l2.out_features, pooled.data_ptr(),
                                               out.data_ptr(), ws.data_ptr(), _lib.stream_ptr(features.device)))
        return pooled, out

    def _hip_body_ok(self, tokens: torch.Tensor) -> bool:
        if self.linear_dtype != "bf16x3" or self.training or torch.is_grad_enabled():
            return False
        if not tokens.is_cuda or tokens.dtype != torch.float32:
            return False
        lins = [self.feature_proj, self.output_decoder[0], self.output_decoder[2]]
        return (all(hip_linear_supported(m.in_features, m.out_features) for m in lins)
                and all(HipBody.layer_supported(layer, tokens.shape[1]) for layer in self.chaos_layers))

    def forward(self, x: torch.Tensor, return_features: bool = False, chaos_noise: Optional[torch.Tensor] = None,
                encoder_dtype: Optional[str] = None) -> dict:
        """x: [B,1,H,W].  chaos_noise (optional): [num_layers,3,B,1] standard-normal draws replacing the reference's
        in-forward torch.randn (chaos_attention.py:50-52) so results can be pinned."""
        dt = encoder_dtype or self.encoder_dtype
        if self._encoder_route(x) == "hip" and dt in ("bf16x3", "bf16"):
            # the bf16 MFMA kernels write the token-major layout feature_proj consumes (smokephys_net.py:95) directly
            flattened = self.hip_encoder().tokens(x, input_dim=self.input_dim, dtype=dt)
        else:
            flattened = self.encode_frames(x, encoder_dtype).flatten(2).transpose(1, 2)
        return self.forward_tokens(flattened, return_features, chaos_noise)

    def forward_tokens(self, flattened: torch.Tensor, return_features: bool = False, chaos_noise: Optional[torch.Tensor] = None) -> dict:
        """Everything behind the encoder (smokephys_net.py:95-122): tokens [B, 1024, input_channels] -> the forward's result dict.
        forward() ends here; forward_volumes() enters here with the 3-D encoder's tokens."""
        B = flattened.shape[0]
        pool_size = 32
        if self._hip_body_ok(flattened):
            features, output_features = self._body_hip(flattened.contiguous(), chaos_noise, pool_size)
        else:
            features = self.feature_proj(flattened)
            features = features + self._pos_embed(pool_size)
            for li, layer in enumerate(self.chaos_layers):
                features = layer(features, noise=None if chaos_noise is None else chaos_noise[li])
            output_features = self.output_decoder(features)
        if (not self.training and not torch.is_grad_enabled() and self.linear_dtype == "bf16x3" and output_features.is_cuda
                and output_features.dtype == torch.float32 and output_features.shape[2] == 64
                and hip_decoder_supported(self.reconstruction_head, pool_size)):
            reconstructed = self.hip_decoder()(output_features)           # 3 fused launches, BN folded
        else:
            output_reshaped = output_features.transpose(1, 2).reshape(B, -1, pool_size, pool_size)
            reconstructed = self.reconstruction_head(output_reshaped)
        if self._hip_tail_ok(features):
            pooled_features, physics_pred = self._tail_hip(features)             # token mean + the physics MLP: three small launches
        else:
            pooled_features = features.mean(dim=1)
            physics_pred = self.physics_head(pooled_features)
        results = {"reconstructed": reconstructed, "physics_features": physics_pred, "latent_features": pooled_features}
        if return_features:
            results["intermediate_features"] = features
        return results

    def forward_volumes(self, volumes: torch.Tensor, encoder3d, return_features: bool = False,
                        chaos_noise: Optional[torch.Tensor] = None) -> dict:
        """BASELINE configs[4]: volumes [B, 1, D, H, W] through the 3-D encoder (models/encoder3d.HipEncoder3D, SPEC_3D.md section 8: the
        depth axis is pooled to 1, so it emits the [B, 1024, 128] tokens the 2-D encoder emits) and then the unchanged network."""
        return self.forward_tokens(encoder3d.tokens(volumes), return_features, chaos_noise)


class ChaosTransformerLayer(nn.Module):
    """smokephys_net.py:136-168 (pre-LN block)."""

    def __init__(self, dim: int, num_heads: int, chaos_strength: float = 0.1, dropout: float = 0.1):
        super().__init__()
        self.chaos_attention = ChaosAttention(dim, num_heads, chaos_strength)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.ffn = nn.Sequential(TrainableHipLinear(dim, 4 * dim), nn.GELU(), nn.Dropout(dropout), TrainableHipLinear(4 * dim, dim),
                                 nn.Dropout(dropout))

    hip_train = False        # set by SmokePhysNet(linear_dtype="bf16x3"): LayerNorm forward + backward on libsmokehip in training

    def _norm(self, ln: nn.LayerNorm, x: torch.Tensor) -> torch.Tensor:
        if (self.hip_train and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and hip_layernorm_supported(x.shape[-1])
                and ln.weight is not None and ln.bias is not None):
            return hip_layernorm_train(x, ln)
        return ln(x)

    def _ffn_fused_ok(self, x: torch.Tensor) -> bool:
        lin0, act, drop0, lin1, drop1 = self.ffn
        return (self.hip_train and torch.is_grad_enabled() and hip_ffn_elementwise_supported(x) and type(act) is nn.GELU
                and getattr(act, "approximate", "none") == "none" and type(drop0) is nn.Dropout and type(drop1) is nn.Dropout
                and drop0.p < 1.0 and drop1.p < 1.0 and not drop0.inplace and not drop1.inplace)

    def forward(self, x: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = self.chaos_attention(self._norm(self.norm1, x), noise=noise, residual=x)      # x + attn(norm1(x)): the add in out_proj's epilogue
        if self._ffn_fused_ok(x):
            # GELU + dropout and dropout + residual add as one libsmokehip pass each (forward and backward): the reference's five modules
            # and the add move 12 full tensors of [B, L, 4D] / [B, L, D] per layer through HBM, these four calls 7, and no mask is stored
            lin0, _, drop0, lin1, drop1 = self.ffn
            h = hip_gelu_dropout(lin0(self._norm(self.norm2, x)), drop0.p, drop0.training)
            return hip_dropout_add(lin1(h), x, drop1.p, drop1.training)
        x = x + self.ffn(self._norm(self.norm2, x))
        return x
