"""The 3-D encoder (BASELINE configs[4]: "MFMA conv3d encoder", first slice) through the C ABI -- smk_conv3d_im2col -> smk_linear_forward
(split-bf16 MFMA) -> smk_pool3d_accumulate -- against oracle/encoder3d.py (fp64 numpy restatement of SPEC_3D.md section 8, itself checked
against torch's CPU conv3d / batch_norm / adaptive_avg_pool3d in tests/test_oracle_golden.py).  Bar: 1e-4 relative (max-norm), the
path's float tolerance; the patch matrices are index work and are checked exactly."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle.encoder3d import encoder3d_features

pytestmark = pytest.mark.gpu

from smokephysai_amd import _lib                                  # noqa: E402
from smokephysai_amd.models import HipEncoder3D                   # noqa: E402


def _weights(seed=0):
    rng = np.random.RandomState(seed)
    return {k: v.astype(np.float32) for k, v in dict(
        conv1_w=rng.randn(64, 1, 7, 7, 7) * 0.05, conv1_b=rng.randn(64) * 0.1, bn1_w=rng.rand(64) + 0.5, bn1_b=rng.randn(64) * 0.1,
        bn1_mean=rng.randn(64) * 0.2, bn1_var=rng.rand(64) + 0.3, conv2_w=rng.randn(128, 64, 3, 3, 3) * 0.03,
        conv2_b=rng.randn(128) * 0.1, bn2_w=rng.rand(128) + 0.5, bn2_b=rng.randn(128) * 0.1,
        bn2_mean=rng.randn(128) * 0.2, bn2_var=rng.rand(128) + 0.3).items()}


def test_patch_matrices_are_exact():
    """smk_conv3d_im2col: every entry is a copy of one source element or a padding zero -- exact, for the scalar (7^3) and the
    channels-last (3^3 x 64) form, for a slab in the middle of the volume and at both ends."""
    L = _lib.load()
    rng = np.random.RandomState(1)
    D, H, W = 6, 10, 12
    vol = rng.rand(D, H, W).astype(np.float32)
    a1 = rng.rand(D, H, W, 64).astype(np.float32)
    st = _lib.stream_ptr(torch.device("cuda", 0))
    for (src, C, k, kpad) in ((vol, 1, 7, 384), (a1, 64, 3, 1728)):
        t = torch.from_numpy(src).cuda()
        P = k // 2
        padded = np.pad(src.reshape(D, H, W, C), ((P, P), (P, P), (P, P), (0, 0)))
        for z0, nz in ((0, 2), (2, 3), (4, 2)):
            cols = torch.full((nz * H * W, kpad), -1.0, device="cuda")
            _lib.check(L.smk_conv3d_im2col(t.data_ptr(), C, D, H, W, k, z0, nz, cols.data_ptr(), kpad, st))
            ref = np.zeros((nz, H, W, kpad), np.float32)
            for kz in range(k):
                for ky in range(k):
                    for kx in range(k):
                        tap = (kz * k + ky) * k + kx
                        ref[..., tap * C:(tap + 1) * C] = padded[z0 + kz:z0 + kz + nz, ky:ky + H, kx:kx + W]
            np.testing.assert_array_equal(cols.cpu().numpy().reshape(nz, H, W, kpad), ref)


def test_deep_volume_walks_the_plane_rings_many_times():
    """D = 24: the z-marching kernels' plane rings (seven input planes for conv1, three a1 planes for conv2) wrap several times and the
    depth pooling sums 24 planes -- against the fp64 oracle like the shallow shapes (the rings had been oracle-checked at
    D <= 9 only).  One volume: the oracle needs ~20 s for it."""
    shape = (24, 32, 32)
    w = _weights(24)
    rng = np.random.RandomState(11)
    vol = (rng.rand(*shape) * 1.5).astype(np.float32)
    vol[5:9] = 0.0                                                  # a band of empty planes inside the march
    enc = HipEncoder3D({k: torch.from_numpy(v) for k, v in w.items()})
    assert enc.conv2_mode == "march"
    x = torch.from_numpy(vol).cuda()
    got = enc(x[None, None]).cpu().numpy()[0]
    a1 = enc.conv1_activations(x).cpu().numpy()
    ref, ref_a1 = encoder3d_features(vol, w)
    assert rel_err(got, ref) < 1e-4
    assert rel_err(np.moveaxis(a1, -1, 0), ref_a1) < 1e-4


@pytest.mark.parametrize("shape,slab", [((8, 32, 32), 1 << 22), ((6, 64, 64), 1 << 24), ((3, 128, 32), 2 << 30)])
def test_features_vs_oracle(shape, slab):
    """Two volumes -> [2, 128, 32, 32]: conv1 activations and the pooled features within 1e-4 of the fp64 oracle (measured ~1e-6: split-bf16
    operands, fp32 accumulation); small slab budgets force several slabs per volume, i.e. the accumulation across launches."""
    w = _weights(sum(shape))
    rng = np.random.RandomState(7)
    vols = (rng.rand(2, *shape) * 1.5).astype(np.float32)
    vols[1, :, : shape[1] // 2] = 0.0                               # an empty half: padding and zero activations are exercised
    enc = HipEncoder3D({k: torch.from_numpy(v) for k, v in w.items()}, slab_bytes=slab)
    x = torch.from_numpy(vols).cuda()
    got = enc(x[:, None]).cpu().numpy()
    assert got.shape == (2, 128, 32, 32)
    a1 = enc.conv1_activations(x[0]).cpu().numpy()                  # [D, H, W, 64]
    for b in range(2):
        ref, ref_a1 = encoder3d_features(vols[b], w)
        assert rel_err(got[b], ref) < 1e-4, b
        if b == 0:
            assert rel_err(np.moveaxis(a1, -1, 0), ref_a1) < 1e-4
    tok = enc.tokens(x).cpu().numpy()
    np.testing.assert_array_equal(tok, got.reshape(2, 128, 1024).transpose(0, 2, 1))
    # the explicit-GEMM form of conv2 (patch matrix + the plain layer kernel): the same products; only the slab sizes, hence the order in
    # which the pooled partial sums are added, differ
    ex = HipEncoder3D({k: torch.from_numpy(v) for k, v in w.items()}, slab_bytes=slab, conv2_mode="im2col")
    assert rel_err(ex(x).cpu().numpy(), got) < 1e-6
    # ... and the implicit-GEMM form on the layer kernel (the default is the z-marching kernel with the depth pooling fused)
    im = HipEncoder3D({k: torch.from_numpy(v) for k, v in w.items()}, slab_bytes=slab, conv2_mode="implicit")
    assert enc.conv2_mode == "march"
    assert rel_err(im(x).cpu().numpy(), got) < 1e-6


@pytest.mark.parametrize("D,H,W", [(1, 16, 48), (2, 16, 48), (5, 16, 48), (3, 8, 16), (4, 24, 16)])
def test_march_kernel_depth_edges_and_zsum(D, H, W):
    """smk_conv3d_cl_zsum_forward alone, against the fp64 oracle's conv2 on the same (random, signed) channels-last input: depths 1 and 2
    (planes z-1 / z+1 / z+2 of the ring outside the volume), walls in x and y on every tile (down to a volume that IS one 8 x 16 tile), a
    single tile column of three tiles, no activation and ReLU."""
    from oracle.encoder3d import conv3d
    from smokephysai_amd import _lib
    from smokephysai_amd.models.linear import HipLinear
    rng = np.random.RandomState(D + H)
    a1 = rng.randn(D, H, W, 64).astype(np.float32)
    w2 = (rng.randn(128, 64, 3, 3, 3) * 0.05).astype(np.float32)
    b2 = rng.randn(128).astype(np.float32)
    lin = HipLinear(torch.from_numpy(w2).permute(0, 2, 3, 4, 1).reshape(128, 1728).contiguous(), torch.from_numpy(b2), device="cuda")
    L = _lib.load()
    src = torch.from_numpy(a1).cuda()
    ref = conv3d(np.moveaxis(a1, -1, 0).astype(np.float64), w2.astype(np.float64), b2.astype(np.float64), 1)      # [128, D, H, W]
    for act, f in ((_lib.SMK_ACT_NONE, lambda t: t), (_lib.SMK_ACT_RELU, lambda t: np.maximum(t, 0.0))):
        zsum = torch.full((H, W, 128), float("nan"), device="cuda")
        _lib.check(L.smk_conv3d_cl_zsum_forward(lin._handle, src.data_ptr(), D, H, W, zsum.data_ptr(), act, _lib.stream_ptr(src.device)))
        want = np.moveaxis(f(ref).sum(axis=1), 0, -1)                 # [H, W, 128]
        assert rel_err(zsum.cpu().numpy(), want) < 1e-5, (D, act)
    with pytest.raises(RuntimeError):
        _lib.check(L.smk_conv3d_cl_zsum_forward(lin._handle, src.data_ptr(), D, 12, W, zsum.data_ptr(), 0, _lib.stream_ptr(src.device)))


@pytest.mark.parametrize("D,H,W", [(1, 16, 32), (4, 16, 32), (9, 16, 32), (3, 8, 16), (8, 24, 48)])
def test_conv1_march_kernel_depth_edges(D, H, W):
    """smk_conv3d_s7_march_forward alone against the fp64 oracle's conv1 (7 x 7 x 7, padding 3) on a signed random volume: depths below, at and
    above the ring's seven planes, every tile touching a wall in x or y, both activations."""
    from oracle.encoder3d import conv3d
    from smokephysai_amd.models.linear import HipLinear
    rng = np.random.RandomState(10 + D + W)
    vol = rng.randn(D, H, W).astype(np.float32)
    w1 = (rng.randn(64, 1, 7, 7, 7) * 0.05).astype(np.float32)
    b1 = rng.randn(64).astype(np.float32)
    w1m = torch.zeros(64, 7, 8, 8)
    w1m[:, :, :7, :7] = torch.from_numpy(w1[:, 0])
    lin = HipLinear(w1m.reshape(64, 448).contiguous(), torch.from_numpy(b1), device="cuda")
    L = _lib.load()
    src = torch.from_numpy(vol).cuda()
    ref = conv3d(vol[None].astype(np.float64), w1.astype(np.float64), b1.astype(np.float64), 3)      # [64, D, H, W]
    for act, f in ((_lib.SMK_ACT_NONE, lambda t: t), (_lib.SMK_ACT_RELU, lambda t: np.maximum(t, 0.0))):
        a1 = torch.full((D, H, W, 64), float("nan"), device="cuda")
        _lib.check(L.smk_conv3d_s7_march_forward(lin._handle, src.data_ptr(), D, H, W, a1.data_ptr(), act, _lib.stream_ptr(src.device)))
        assert rel_err(a1.cpu().numpy(), np.moveaxis(f(ref), 0, -1)) < 1e-5, (D, act)


def test_loud_failures():
    enc = HipEncoder3D({k: torch.from_numpy(v) for k, v in _weights().items()})
    with pytest.raises(ValueError):
        enc(torch.zeros(1, 1, 4, 96, 32, device="cuda"))            # 96: the two adaptive pools do not compose to a uniform block mean
    with pytest.raises(ValueError):
        enc(torch.zeros(1, 2, 4, 32, 32, device="cuda"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HipEncoder3D({k: torch.from_numpy(v) for k, v in _weights().items()}, device="cpu")


def test_full_network_on_volumes():
    """configs[4] end to end: volumes [B, 1, D, H, W] -> HipEncoder3D tokens -> the unchanged SmokePhysNet body and heads
    (SmokePhysNet.forward_volumes).  The 3-D encoder emits the tensor the 2-D encoder emits, so the result must equal forward_tokens on
    those tokens, and the libsmokehip body must stay within 1e-4 of the PyTorch-ROCm fp32 body on them."""
    from smokephysai_amd.models import SmokePhysNet
    torch.manual_seed(0)
    model = SmokePhysNet().cuda().eval()
    enc = HipEncoder3D({k: torch.from_numpy(v) for k, v in _weights(3).items()})
    g = torch.Generator(device="cuda").manual_seed(1)
    vol = torch.rand(2, 1, 4, 64, 32, device="cuda", generator=g)
    noise = torch.randn(len(model.chaos_layers), 3, 2, 1, device="cuda", generator=g)
    with torch.no_grad():
        out = model.forward_volumes(vol, enc, chaos_noise=noise)
        tok = enc.tokens(vol)
        same = model.forward_tokens(tok, chaos_noise=noise)
        model.linear_dtype = "f32"
        ref = model.forward_tokens(tok, chaos_noise=noise)
        model.linear_dtype = "bf16x3"
    assert out["reconstructed"].shape == (2, 1, 128, 128) and out["physics_features"].shape == (2, 3)
    for k in ref:
        assert torch.equal(out[k], same[k]), k
        assert rel_err(out[k].cpu().numpy(), ref[k].cpu().numpy()) < 1e-4, k


def test_config4_full_size_volume_three_forms_agree():
    """One 512 x 512 x 64 volume (configs[4]'s size: 16.8 M voxels, conv1 output 4.3 GB, i.e. byte offsets past 2^32 inside the volume): the
    z-marching kernels (per-plane buffer resources, 2,048 tile columns, the full 64-plane ring walk), the implicit-GEMM convolutions (shifted
    addressing inside the layer kernel, 32-bit offsets per slab) and the explicit ones (patch matrix from smk_conv3d_im2col -- checked exactly
    above -- times the plain layer kernel).  Three different data paths through the same arithmetic: the features must agree to
    summation-order accuracy, and every token must be finite and non-trivial."""
    w = {k: torch.from_numpy(v) for k, v in _weights(11).items()}
    g = torch.Generator(device="cuda").manual_seed(4)
    vol = torch.rand(1, 64, 512, 512, device="cuda", generator=g)
    vol[0, :, 300:, :] = 0.0                                        # an empty region: zero activations and the volume's walls both occur
    enc = HipEncoder3D(w)
    assert enc.conv2_mode == "march"
    mar = enc(vol).clone()
    del enc
    imp = HipEncoder3D(w, conv2_mode="implicit")(vol).clone()
    exp = HipEncoder3D(w, conv2_mode="im2col", slab_bytes=1 << 30)(vol)
    assert mar.shape == (1, 128, 32, 32) and torch.isfinite(mar).all()
    assert rel_err(mar.cpu().numpy(), exp.cpu().numpy()) < 1e-6
    assert rel_err(imp.cpu().numpy(), exp.cpu().numpy()) < 1e-6
    assert float(mar.std()) > 0 and float((mar[0, :, :12] - mar[0, :, 20:]).abs().max()) > 0
