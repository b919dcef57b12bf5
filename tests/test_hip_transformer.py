"""GPU parity of the non-GEMM transformer-body kernels of libsmokehip (csrc/transformer.hip) against the PyTorch
restatement of the reference's ops (chaos_attention.py, smokephys_net.py:136-168), fp32, tolerance 1e-4 max-norm
(measured ~1e-6)."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

from smokephysai_amd.models.chaos_attention import ChaosAttention      # noqa: E402


@pytest.mark.parametrize("B,dim", [(1, 512), (5, 512), (64, 512), (3, 64), (2, 2048)])
def test_chaos_addend_kernel_matches_torch_ops(B, dim):
    """smk_chaos_addend = 0.1 * sigmoid(gate(C)) * C on the five Lorenz states (chaos_attention.py:39-66,85-100)."""
    torch.manual_seed(B * 1000 + dim)
    att = ChaosAttention(dim, num_heads=8).cuda().eval()
    noise = torch.randn(3, B, 1, device="cuda")
    with torch.no_grad():
        ref = att.chaos_addend(B, noise.device, torch.float32, noise)
        out = att.chaos_addend_hip(B, noise.device, noise)
    assert out.shape == ref.shape == (B, 5, dim)
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 1e-5


def test_chaos_addend_draws_noise_like_the_reference():
    """Unpinned: three randn(B,1) generator calls per layer, in the reference's order (chaos_attention.py:50-52)."""
    att = ChaosAttention(512).cuda().eval()
    with torch.no_grad():
        torch.manual_seed(123)
        a = att.chaos_addend_hip(4, torch.device("cuda"))
        torch.manual_seed(123)
        noise = torch.stack([torch.randn(4, 1, device="cuda") for _ in range(3)])
        b = att.chaos_addend(4, noise.device, torch.float32, noise)
    assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5


def _attention_ref(q, k, v, H, scale):
    B, L, D = q.shape
    d = D // H
    qh, kh, vh = (t.double().view(B, L, H, d).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, L, D)


@pytest.mark.parametrize("B,L,H,spread,tol", [(1, 128, 8, 1.0, 2e-5), (2, 1024, 8, 1.0, 2e-5), (3, 256, 2, 4.0, 1e-4),
                                               (1, 1024, 8, 12.0, 5e-4)])
def test_attention_matches_fp64_softmax_attention(B, L, H, spread, tol):
    """smk_attention against softmax(QK^T / 8) V in fp64 (chaos_attention.py:102-112); `spread` widens the score range so
    that the online-softmax rescale path is exercised hard (running max grows across key tiles).  The logits are split-bf16
    products: their absolute error grows with |score| (~1e-5 |s|), so the wide-range stress cases get wider bounds; at
    the model's score range (|s| of a few units) the result is within 2e-5.  spread 12 = logits up to ~+-150."""
    from smokephysai_amd.models.attention import hip_attention
    g = torch.Generator(device="cuda").manual_seed(B * 131 + L + H)
    D = H * 64
    q = torch.randn(B, L, D, device="cuda", generator=g) * spread
    k = torch.randn(B, L, D, device="cuda", generator=g)
    k[:, :, :] *= torch.linspace(0.2, 2.0, L, device="cuda")[None, :, None]      # later keys score higher: max keeps moving
    v = torch.randn(B, L, D, device="cuda", generator=g)
    out = hip_attention(q, k, v, H, 0.125)
    ref = _attention_ref(q, k, v, H, 0.125)
    sdpa = torch.nn.functional.scaled_dot_product_attention(*(t.view(B, L, H, 64).transpose(1, 2) for t in (q, k, v)), scale=0.125)
    e_hip = rel_err(out.cpu().numpy(), ref.cpu().numpy())
    e_sdpa = rel_err(sdpa.transpose(1, 2).reshape(B, L, D).cpu().numpy(), ref.cpu().numpy())
    assert e_hip < tol, (e_hip, e_sdpa)


@pytest.mark.parametrize("B,L,H", [(1, 1024, 8), (2, 1024, 8), (1, 512, 8), (1, 2048, 2)])
def test_attention_split_keys_for_small_grids(B, L, H, monkeypatch):
    """Grids that would leave most CUs idle (one frame of the model's shape: 64 workgroups) deal the keys of a query block to several
    workgroups and merge their (output, max, sum) states in a second launch: against the fp64 reference, against the unsplit kernel
    (SMK_ATTN_SPLIT=1 is read once per process, so the unsplit result comes from the workspace-less entry point), and run to run."""
    from smokephysai_amd import _lib
    from smokephysai_amd.models.attention import hip_attention
    Lh = _lib.load()
    ws = int(Lh.smk_attention_workspace_bytes(B, L, H, 64))
    assert ws > 0, "this shape is expected to split on a 256-CU device"
    assert int(Lh.smk_attention_workspace_bytes(64, 1024, 8, 64)) == 0          # the full batch fills the chip: no split
    g = torch.Generator(device="cuda").manual_seed(B * 7 + L + H)
    D = H * 64
    q = torch.randn(B, L, D, device="cuda", generator=g) * 2.0
    k = torch.randn(B, L, D, device="cuda", generator=g)
    k[:, :, :] *= torch.linspace(0.2, 2.0, L, device="cuda")[None, :, None]      # later keys score higher: the splits' maxima differ
    v = torch.randn(B, L, D, device="cuda", generator=g)
    out = hip_attention(q, k, v, H, 0.125)
    ref = _attention_ref(q, k, v, H, 0.125)
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 5e-5
    plain = torch.empty_like(out)
    _lib.check(Lh.smk_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), plain.data_ptr(), B, L, H, 64, D, D, D, D, 0.125, 0,
                                _lib.stream_ptr(q.device)))
    # against the unsplit kernel: the probabilities are formed relative to different running maxima before their bf16 split, so the two
    # agree to the arithmetic's accuracy (a few 1e-6), not bit for bit
    assert rel_err(out.cpu().numpy(), plain.cpu().numpy()) < 1e-5
    assert torch.equal(out, hip_attention(q, k, v, H, 0.125))                     # fixed merge order: deterministic


@pytest.mark.parametrize("B,L,H", [(1, 1024, 8), (4, 1024, 8), (24, 1024, 8), (2, 256, 4)])
def test_attention_on_presplit_k_v_equals_the_fp32_route_bit_for_bit(B, L, H):
    """smk_attention_kv with kv_format SMK_FMT_SPLIT4_INPLACE: k and v arrive as the {hi, lo} pairs the kernel would have formed itself, so
    the output equals the fp32 route's bit for bit -- on the key-split small-grid form (one frame), the two-wave-group form (batch 4), the
    single-buffer form (chip-filling grids) and strided slices of one fused q | k | v tensor."""
    from smokephysai_amd.models.attention import hip_attention
    from smokephysai_amd.models.linear import split4_inplace
    g = torch.Generator(device="cuda").manual_seed(B + L + H)
    D = H * 64
    qkv = torch.randn(B, L, 3 * D, device="cuda", generator=g)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    ref = hip_attention(q, k, v, H, 0.125)
    enc = qkv.clone()
    enc[..., D:] = split4_inplace(qkv[..., D:].contiguous())
    out = hip_attention(enc[..., :D], enc[..., D:2 * D], enc[..., 2 * D:], H, 0.125, kv_split=True)
    assert torch.equal(out, ref)
    assert rel_err(out.cpu().numpy(), _attention_ref(q, k, v, H, 0.125).cpu().numpy()) < 2e-5


def test_attention_reads_strided_qkv_and_rejects_other_head_dims():
    from smokephysai_amd.models.attention import hip_attention, hip_attention_supported
    g = torch.Generator(device="cuda").manual_seed(5)
    qkv = torch.randn(2, 256, 3 * 512, device="cuda", generator=g)
    q, k, v = qkv[..., :512], qkv[..., 512:1024], qkv[..., 1024:]
    out = hip_attention(q, k, v, 8, 0.125)
    assert rel_err(out.cpu().numpy(), _attention_ref(q, k, v, 8, 0.125).cpu().numpy()) < 2e-5
    assert not hip_attention_supported(1024, 32) and not hip_attention_supported(100, 64)
    with pytest.raises(Exception, match="head_dim 64"):
        hip_attention(qkv[..., :256].contiguous(), qkv[..., :256].contiguous(), qkv[..., :256].contiguous(), 8, 0.125)


@pytest.mark.parametrize("rows,D", [(1024, 512), (4099, 512), (37, 64), (256, 2048), (5, 132)])
def test_layernorm_matches_torch(rows, D):
    from smokephysai_amd.models.attention import hip_layernorm
    g = torch.Generator(device="cuda").manual_seed(rows + D)
    ln = torch.nn.LayerNorm(D).cuda()
    with torch.no_grad():
        ln.weight.copy_(torch.randn(D, device="cuda", generator=g))
        ln.bias.copy_(torch.randn(D, device="cuda", generator=g))
        x = torch.randn(rows, D, device="cuda", generator=g) * 3 + 1.5
        ref = torch.nn.functional.layer_norm(x.double(), (D,), ln.weight.double(), ln.bias.double(), ln.eps)
        out = hip_layernorm(x, ln)
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 2e-6


def test_hip_transformer_layer_matches_reference_golden(golden):
    """One ChaosTransformerLayer through libsmokehip (LayerNorm, chaos addend, fused q|k|v, flash attention, out_proj +
    residual, FFN with GELU + residual) against the REFERENCE's captured output for the same weights, input and noise
    draws (tests/golden/transformer_layer.npz; dim 128, 2 heads of 64, L = 128) -- and the attention sub-block alone."""
    from smokephysai_amd.models import ChaosTransformerLayer
    from smokephysai_amd.models.hip_body import HipBody
    g = golden("transformer_layer.npz")
    layer = ChaosTransformerLayer(128, 2, chaos_strength=0.1)
    layer.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w::")})
    layer = layer.cuda().eval()
    x = torch.from_numpy(g["x"]).cuda()
    noise = torch.from_numpy(g["noise"]).cuda()
    body = HipBody()
    assert HipBody.layer_supported(layer, x.shape[1])
    with torch.no_grad():
        out = body.layer("t.", layer, x.clone(), noise)
        assert rel_err(out.cpu().numpy(), g["layer_out"]) < 1e-4
        assert body.kv_presplit                               # (default: k | v written split by the q | k | v layer's epilogue)
        body.kv_presplit = False                              # fp32 k | v, split inside the attention kernel: the same bits
        assert torch.equal(body.layer("t.", layer, x.clone(), noise), out)
        body.kv_presplit = True
        body.split_activations = True                         # same layer with split-bf16 activations between the kernels
        out_s = body.layer("t.", layer, x.clone(), noise)
        assert rel_err(out_s.cpu().numpy(), g["layer_out"]) < 1e-4
        body.split_activations = False
        # attention sub-block: LN1 -> q|k|v (+ chaos term) -> attention -> out_proj (no residual)
        from smokephysai_amd.models.attention import hip_attention
        h = body.layernorm(x, layer.norm1)
        att = layer.chaos_attention
        add15 = torch.zeros(2, 5, 3 * 128, device="cuda")
        att.chaos_addend_hip(2, x.device, noise, out=add15)
        qkv = body.qkv("t.chaos_attention.qkv", att)(h, periodic_add=add15, rows_per_group=128)
        o = hip_attention(qkv[..., :128], qkv[..., 128:256], qkv[..., 256:], 2, 0.125)
        attn = body.linear("t.chaos_attention.out_proj", att.out_proj)(o)
        assert rel_err(attn.cpu().numpy(), g["attention_out"]) < 1e-4


def test_layernorm_and_attention_split_bf16_outputs():
    from smokephysai_amd.models.attention import hip_attention, hip_layernorm
    from smokephysai_amd.models.linear import from_split
    g = torch.Generator(device="cuda").manual_seed(17)
    ln = torch.nn.LayerNorm(512).cuda()
    x = torch.randn(3, 256, 512, device="cuda", generator=g) * 2 + 0.3
    with torch.no_grad():
        y = hip_layernorm(x, ln)
        ys = hip_layernorm(x, ln, out_split=True)
    assert ys.shape == (3, 256, 64, 2, 8) and rel_err(from_split(ys).cpu().numpy(), y.cpu().numpy()) < 2.0 ** -16
    q, k, v = (torch.randn(2, 256, 512, device="cuda", generator=g) for _ in range(3))
    o = hip_attention(q, k, v, 8, 0.125)
    os_ = hip_attention(q, k, v, 8, 0.125, out_split=True)
    assert os_.shape == (2, 256, 64, 2, 8) and rel_err(from_split(os_).cpu().numpy(), o.cpu().numpy()) < 2.0 ** -16


@pytest.mark.parametrize("B,S", [(1, 32), (3, 32), (2, 16), (12, 32)])      # (B <= 8: the latency-cut small-batch kernels; 12: the tiled ones)
def test_decoder_head_matches_torch(B, S):
    """smk_decoder_forward (BN-folded direct fp32 kernels) against the PyTorch reconstruction_head in eval mode
    (smokephys_net.py:57-66,117-118), including the tokens -> [B,64,S,S] re-view."""
    from smokephysai_amd.models import SmokePhysNet
    from smokephysai_amd.models.decoder import HipDecoder, decoder_weight_dict, hip_decoder_supported
    torch.manual_seed(B * 10 + S)
    head = SmokePhysNet().reconstruction_head.cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(S)
    with torch.no_grad():
        for m in head:
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.num_features, device="cuda", generator=g) * 0.2)
                m.running_var.copy_(torch.rand(m.num_features, device="cuda", generator=g) + 0.5)
                m.weight.copy_(torch.randn(m.num_features, device="cuda", generator=g) * 0.3 + 1.0)
                m.bias.copy_(torch.randn(m.num_features, device="cuda", generator=g) * 0.1)
        assert hip_decoder_supported(head, S)
        tokens = torch.randn(B, S * S, 64, device="cuda", generator=g)
        ref = head(tokens.transpose(1, 2).reshape(B, 64, S, S).double().float())
        ref64 = head.double()(tokens.double().transpose(1, 2).reshape(B, 64, S, S))
        head.float()
        out = HipDecoder(decoder_weight_dict(head))(tokens)
    assert out.shape == (B, 1, 4 * S, 4 * S)
    assert rel_err(out.cpu().numpy(), ref64.cpu().numpy()) < 5e-6
    assert rel_err(ref.cpu().numpy(), ref64.cpu().numpy()) < 1e-4


def test_large_inputs_are_chunked_transparently(monkeypatch):
    """One launch addresses its inputs with 32-bit offsets; the wrappers split larger problems into row / batch chunks.
    With the thresholds lowered, chunked results must equal the single-launch results bit for bit."""
    import math
    from smokephysai_amd.models import attention as A, linear as Lm
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(6, 256, 512, device="cuda", generator=g)
    w = torch.randn(1536, 512, device="cuda", generator=g) / math.sqrt(512)
    b = torch.randn(1536, device="cuda", generator=g)
    padd = torch.randn(6, 5, 1536, device="cuda", generator=g)
    res = torch.randn(6, 256, 1536, device="cuda", generator=g)
    lin = Lm.HipLinear(w, b)
    y_p, y_r = lin(x, periodic_add=padd), lin(x, residual=res, activation=None)
    qkv = y_p
    o = A.hip_attention(qkv[..., :512], qkv[..., 512:1024], qkv[..., 1024:], 8, 0.125)
    monkeypatch.setattr(Lm, "MAX_X_ELEMS", (512 + 256) * 512 + 1)          # -> chunks of 512 rows (2 groups of 256)
    monkeypatch.setattr(A, "MAX_QKV_ELEMS", 2 * 256 * 1536 + 1)            # -> chunks of 2 batch elements
    assert torch.equal(lin(x, periodic_add=padd), y_p)
    assert torch.equal(lin(x, residual=res), y_r)
    assert torch.equal(A.hip_attention(qkv[..., :512], qkv[..., 512:1024], qkv[..., 1024:], 8, 0.125), o)


# ---------------------------------------------------------------- training: attention forward + backward on libsmokehip
@pytest.mark.parametrize("B,L,H,spread", [(1, 128, 2, 1.0), (2, 1024, 8, 1.0), (2, 256, 4, 3.0), (1, 384, 1, 1.0)])
def test_attention_backward_matches_fp64_autograd(B, L, H, spread):
    """smk_attention_forward_lse + smk_attention_backward (dq, dk, dv) against fp64 softmax attention under autograd;
    strided q/k/v (slices of one fused tensor); deterministic."""
    from smokephysai_amd.models.attention import hip_attention_train
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + L + H)
    D = 64 * H
    qkv = torch.randn(B, L, 3 * D, device="cuda", generator=g) * spread
    qkv.requires_grad_(True)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    dout = torch.randn(B, L, D, device="cuda", generator=g)
    scale = 0.125
    out = hip_attention_train(q, k, v, H, scale)
    out.backward(dout)
    got = qkv.grad.clone()

    ref_in = qkv.detach().double().requires_grad_(True)
    q64, k64, v64 = [ref_in[..., i * D:(i + 1) * D].view(B, L, H, 64).transpose(1, 2) for i in range(3)]
    p = torch.softmax(q64 @ k64.transpose(-1, -2) * scale, dim=-1)
    ref = (p @ v64).transpose(1, 2).reshape(B, L, D)
    ref.backward(dout.double())
    tol = 2e-5 if spread == 1.0 else 1e-4

    def err(a, b):
        return rel_err(a.detach().cpu().numpy(), b.detach().cpu().numpy())
    assert err(out, ref) < tol
    for i, name in enumerate("qkv"):
        e = err(got[..., i * D:(i + 1) * D], ref_in.grad[..., i * D:(i + 1) * D])
        assert e < tol, (name, e)
    qkv.grad = None
    hip_attention_train(q, k, v, H, scale).backward(dout)
    assert torch.equal(qkv.grad, got)


def test_chaos_attention_module_trains_on_the_hip_attention():
    """ChaosAttention (dim 128, 2 heads of 64) in train mode: output and every parameter / input gradient with linears + attention on
    libsmokehip against the same module in fp64 on PyTorch ops."""
    import copy
    torch.manual_seed(11)
    attn = ChaosAttention(128, 2).cuda().train()
    attn.hip_train = True
    for m in attn.modules():
        if hasattr(m, "hip_train"):
            m.hip_train = True
    ref_mod = copy.deepcopy(attn).double()
    for m in ref_mod.modules():
        if hasattr(m, "hip_train"):
            m.hip_train = False
    x = torch.randn(2, 256, 128, device="cuda", requires_grad=True)
    noise = torch.randn(3, 2, 1, device="cuda")
    dy = torch.randn(2, 256, 128, device="cuda")
    y = attn(x, noise=noise)
    assert "HipLinearFn" in type(y.grad_fn).__name__                    # out_proj on the HIP node, fed by _HipAttentionFn
    y.backward(dy)
    x64 = x.detach().double().requires_grad_(True)
    y64 = ref_mod(x64, noise=noise.double())
    y64.backward(dy.double())

    def err(a, b):
        return rel_err(a.detach().cpu().numpy(), b.detach().cpu().numpy())
    assert err(y, y64) < 1e-4 and err(x.grad, x64.grad) < 1e-4
    scale = max(float(p.grad.abs().max()) for p in ref_mod.parameters())
    for (n, p), (_, p64) in zip(attn.named_parameters(), ref_mod.named_parameters()):
        if float(p64.grad.abs().max()) > 1e-9 * scale:
            assert err(p.grad, p64.grad) < 1e-4, n


@pytest.mark.parametrize("rows,D", [(4096, 512), (1030, 512), (37, 64), (300, 2048), (5, 132)])
def test_layernorm_backward_matches_fp64_autograd(rows, D):
    from smokephysai_amd.models.attention import hip_layernorm_train
    torch.manual_seed(rows + D)
    ln = torch.nn.LayerNorm(D).cuda()
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.uniform_(-0.5, 0.5)
    x = (torch.randn(rows, D, device="cuda") * 2.0 + 1.0).requires_grad_(True)
    dy = torch.randn(rows, D, device="cuda")
    y = hip_layernorm_train(x, ln)
    y.backward(dy)
    ln64 = torch.nn.LayerNorm(D).cuda().double()
    ln64.load_state_dict({k: v.double() for k, v in ln.state_dict().items()})
    x64 = x.detach().double().requires_grad_(True)
    y64 = ln64(x64)
    y64.backward(dy.double())

    def err(a, b):
        return rel_err(a.detach().cpu().numpy(), b.detach().cpu().numpy())
    assert err(y, y64) < 1e-5 and err(x.grad, x64.grad) < 1e-5
    assert err(ln.weight.grad, ln64.weight.grad) < 1e-5 and err(ln.bias.grad, ln64.bias.grad) < 1e-5
    g1 = (x.grad.clone(), ln.weight.grad.clone())
    x.grad = None
    ln.zero_grad()
    hip_layernorm_train(x, ln).backward(dy)
    assert torch.equal(x.grad, g1[0]) and torch.equal(ln.weight.grad, g1[1])


def test_ffn_elementwise_kernels_vs_fp64_autograd():
    """smk_ffn_elementwise: dropout(gelu(h)) and residual + dropout(y) with their backward passes.  p = 0: forward and gradients against
    fp64 autograd of the exact-erf GELU; p = 0.1: the keep mask is a function of (seed, index) -- the same mask in forward and backward,
    keep rate 0.9 +- 0.5 %, kept elements scaled by 1 / (1 - p), and a different seed gives a different mask."""
    from smokephysai_amd.models import ffn
    g = torch.Generator(device="cuda").manual_seed(5)
    h = (torch.randn(257, 512, device="cuda", generator=g) * 2.5).requires_grad_(True)
    go = torch.randn(257, 512, device="cuda", generator=g)
    out = ffn.hip_gelu_dropout(h, 0.0)
    out.backward(go)
    h64 = h.detach().double().requires_grad_(True)
    ref = torch.nn.functional.gelu(h64)
    ref.backward(go.double())
    assert float((out.double() - ref).abs().max()) < 2e-6 and float((h.grad.double() - h64.grad).abs().max()) < 5e-6 * float(go.abs().max())
    # dropout: same mask both ways
    h2 = h.detach().clone().requires_grad_(True)
    seed = 1234567
    a = ffn._GeluDropoutFn.apply(h2, 0.1, seed)
    a.backward(torch.ones_like(a))
    keep = a != 0
    dense = torch.nn.functional.gelu(h2.detach())
    nz = dense.abs() > 1e-6
    rate = float(keep[nz].float().mean())
    assert abs(rate - 0.9) < 5e-3, rate
    assert torch.allclose(a[keep], dense[keep] / 0.9, rtol=2e-5, atol=1e-6)
    assert bool(((h2.grad != 0) == keep)[nz & (h2.detach().abs() > 1e-3)].all())               # backward used the forward's mask
    b = ffn._GeluDropoutFn.apply(h2.detach(), 0.1, seed + 1)
    assert not torch.equal(b != 0, keep) and torch.equal(ffn._GeluDropoutFn.apply(h2.detach(), 0.1, seed), a.detach())
    # residual + dropout(y)
    y = torch.randn(64, 1024, 512, device="cuda", generator=g).requires_grad_(True)
    res = torch.randn(64, 1024, 512, device="cuda", generator=g).requires_grad_(True)
    o0 = ffn.hip_dropout_add(y, res, 0.0)
    assert torch.equal(o0, y.detach() + res.detach())
    o = ffn._DropoutAddFn.apply(y, res, 0.1, 99)
    gout = torch.randn_like(o)
    o.backward(gout)
    m = ffn._DropoutAddFn.apply(torch.ones_like(res), torch.zeros_like(res), 0.1, 99) != 0   # the same (seed, index) mask, read off ones + 0
    assert abs(float(m.float().mean()) - 0.9) < 2e-3
    scale = 1.0 / (1.0 - round(0.1 * 65536) / 65536)                                       # p is applied in units of 2^-16
    assert torch.allclose(o.detach(), res.detach() + y.detach() * m * scale, rtol=1e-5, atol=2e-6)
    assert torch.equal(res.grad, gout) and torch.allclose(y.grad, gout * m / 0.9, rtol=2e-5, atol=0)    # p is applied in units of 2^-16
    # off-GPU / odd sizes are refused, not silently computed elsewhere
    assert not ffn.hip_ffn_elementwise_supported(torch.zeros(3)) and not ffn.hip_ffn_elementwise_supported(torch.zeros(6, device="cuda"))


def test_transformer_layer_training_ffn_route_matches_the_module_chain():
    """ChaosTransformerLayer in train mode with the fused GELU / dropout / residual kernels (dropout p = 0 so that both routes are
    deterministic) against the same layer forced through the reference's module chain: outputs and every gradient agree."""
    from smokephysai_amd.models import ChaosTransformerLayer
    torch.manual_seed(0)
    layer = ChaosTransformerLayer(512, 8).cuda().train()
    for m in layer.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "hip_train"):
            m.hip_train = True
    x = torch.randn(2, 1024, 512, device="cuda")
    noise = torch.randn(3, 2, 1, device="cuda")
    assert layer._ffn_fused_ok(x)
    outs, grads = [], []
    for fused in (True, False):
        layer.zero_grad()
        layer._ffn_fused_ok = (lambda t: True) if fused else (lambda t: False)
        xin = x.clone().requires_grad_(True)
        y = layer(xin, noise=noise)
        y.square().mean().backward()
        outs.append(y.detach())
        grads.append([xin.grad.clone()] + [p.grad.clone() for p in layer.parameters()])
    del layer._ffn_fused_ok
    assert float((outs[0] - outs[1]).abs().max()) < 2e-5 * float(outs[1].abs().max())
    gscale = max(float(b.abs().max()) for b in grads[1])              # (k_proj.bias has an exactly-zero gradient: softmax ignores a key bias)
    for a, b in zip(*grads):
        assert float((a - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1e-3 * gscale)


def test_fused_qkv_attention_training_node_matches_the_three_projection_route(monkeypatch):
    """ChaosAttention in train mode: q | k | v projection + chaos addend + attention as one autograd node (one [3D, D] launch, dq | dk | dv in
    one buffer, one dX GEMM, one weight-gradient call) against the route with three TrainableHipLinear projections: output and every
    gradient (x, the three weights and biases, chaos_proj / chaos_gate through the addend) agree; a parameter update is seen."""
    from smokephysai_amd.models.chaos_attention import ChaosAttention
    torch.manual_seed(1)
    att = ChaosAttention(512, 8).cuda().train()
    att.hip_train = True
    for m in att.modules():
        if hasattr(m, "hip_train"):
            m.hip_train = True
    x = torch.randn(2, 1024, 512, device="cuda")
    noise = torch.randn(3, 2, 1, device="cuda")
    go = torch.randn(2, 1024, 512, device="cuda")
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SMK_TRAIN_QKV_FUSED", fused)
        att.zero_grad()
        xin = x.clone().requires_grad_(True)
        y = att(xin, noise=noise)
        y.backward(go)
        res[fused] = (y.detach(), [xin.grad.clone()] + [p.grad.clone() for p in att.parameters()])
    assert "_hip_qkv" in att.__dict__
    names = ["x"] + [n for n, _ in att.named_parameters()]
    assert float((res["1"][0] - res["0"][0]).abs().max()) < 2e-5 * float(res["0"][0].abs().max())
    gscale = max(float(b.abs().max()) for b in res["0"][1])
    for n, a, b in zip(names, res["1"][1], res["0"][1]):
        assert float((a - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1e-3 * gscale), n
    # an in-place parameter update re-splits the concatenated mirror
    monkeypatch.setenv("SMK_TRAIN_QKV_FUSED", "1")
    with torch.no_grad():
        att.k_proj.weight.mul_(1.5)
    y2 = att(x, noise=noise)
    monkeypatch.setenv("SMK_TRAIN_QKV_FUSED", "0")
    y3 = att(x, noise=noise)
    assert float((y2 - y3).abs().max()) < 2e-5 * float(y3.abs().max()) and not torch.allclose(y2, res["1"][0], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("B,L,H", [(2, 128, 8), (3, 256, 2), (1, 384, 11)])
def test_attention_delta_kernel(B, L, H):
    """smk_attention_delta: rowsum(dout * out) per head (the softmax-backward row term), dense and row-strided inputs."""
    from smokephysai_amd.models.attention import hip_attention_delta
    g = torch.Generator(device="cuda").manual_seed(B * L + H)
    D = 64 * H
    dout = torch.randn(B, L, D, device="cuda", generator=g)
    wide = torch.randn(B, L, D + 64, device="cuda", generator=g)
    out = wide[..., :D]                                              # row pitch > H * 64
    ref = (dout.double() * out.double()).view(B, L, H, 64).sum(-1)
    got = hip_attention_delta(dout, out, H)
    assert got.shape == (B, L, H) and float((got.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())


def test_lorenz_states_kernel_matches_the_torch_chain():
    """smk_lorenz_states: the five Euler states of generate_chaos_field for given noise, against the module's own torch op chain
    (same operation order, fp32; no FMA contraction on either side)."""
    from smokephysai_amd.models.chaos_attention import ChaosAttention
    att = ChaosAttention(128, 2).cuda()
    noise = torch.randn(3, 37, 1, device="cuda")
    ref = att.chaos_states(37, "cuda", noise)
    got = att.chaos_states_hip(37, "cuda", noise)
    assert got.shape == (37, 5, 3) and float((got - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    # drawn noise: the same three generator calls in the same order
    torch.manual_seed(11); a = att.chaos_states(5, "cuda")
    torch.manual_seed(11); b = att.chaos_states_hip(5, "cuda")
    assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max())


def test_batched_chaos_addend_equals_the_per_layer_launches():
    """smk_chaos_addend_batched (one launch for all layers of a model) writes bit for bit what six smk_chaos_addend launches write."""
    from smokephysai_amd.models import SmokePhysNet
    torch.manual_seed(5)
    model = SmokePhysNet().cuda().eval()
    B = 3
    noise = torch.randn(len(model.chaos_layers), 3, B, 1, device="cuda")
    body = model._hip_body
    names = [f"chaos_layers.{i}." for i in range(len(model.chaos_layers))]
    with torch.no_grad():
        assert body.chaos_addends(names, list(model.chaos_layers), noise, B, noise.device)
        for i, layer in enumerate(model.chaos_layers):
            att = layer.chaos_attention
            got = body.addend_bufs[(names[i], B)][:, :, :att.dim].clone()
            ref = att.chaos_addend_hip(B, noise.device, noise[i])
            assert torch.equal(got, ref), i
            assert float(body.addend_bufs[(names[i], B)][:, :, att.dim:].abs().max()) == 0.0       # the k | v columns stay zero


@pytest.mark.parametrize("B,L", [(1, 1024), (5, 1024), (2, 100)])
def test_pooled_head_kernel_matches_torch(B, L):
    """smk_pooled_head: features.mean(dim=1) and Linear(ReLU(Linear(.))) (smokephys_net.py:116-118) against the same modules in fp64."""
    from smokephysai_amd import _lib
    torch.manual_seed(B * 10 + L)
    D, H1, H2 = 512, 256, 3
    x = torch.randn(B, L, D, device="cuda")
    l1, l2 = torch.nn.Linear(D, H1).cuda(), torch.nn.Linear(H1, H2).cuda()
    pooled = torch.empty(B, D, device="cuda")
    out = torch.empty(B, H2, device="cuda")
    ws = torch.empty(B * (32 * D + H1), device="cuda")
    _lib.check(_lib.load().smk_pooled_head(x.data_ptr(), B, L, D, D, l1.weight.data_ptr(), l1.bias.data_ptr(), H1, l2.weight.data_ptr(),
                                           l2.bias.data_ptr(), H2, pooled.data_ptr(), out.data_ptr(), ws.data_ptr(), _lib.stream_ptr(x.device)))
    with torch.no_grad():
        p64 = x.double().mean(dim=1)
        o64 = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(p64, l1.weight.double(), l1.bias.double())), l2.weight.double(),
                                         l2.bias.double())
    assert rel_err(pooled.cpu().numpy(), p64.cpu().numpy()) < 1e-6
    assert rel_err(out.cpu().numpy(), o64.cpu().numpy()) < 1e-5
