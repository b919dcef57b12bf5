"""GPU parity of libsmokehip's split-bf16 linear kernel (smk_linear_*) against the fp32/fp64 GEMM the reference runs
through nn.Linear (smokephys_net.py:38,50-54,153-158; chaos_attention.py:25-28).  Tolerance: 1e-4 relative (max-norm),
the bar SURVEY 8(c) sets for the aten-backed layers; measured errors are ~1e-6."""
import math

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

from smokephysai_amd.models.linear import HipLinear, hip_linear_supported      # noqa: E402

TOL = 1e-4


def _ref(x, w, b, act=None, residual=None, padd=None):
    y = x.double() @ w.double().t() + (0 if b is None else b.double())
    if padd is not None:
        G, P, N = padd.shape
        L = x.shape[-2]
        idx = torch.arange(L, device=x.device) % P
        y = y + padd.double()[:, idx]
    if act == "gelu":
        y = 0.5 * y * (1 + torch.erf(y / math.sqrt(2.0)))
    if residual is not None:
        y = residual.double() + y
    return y


@pytest.mark.parametrize("M,K,N", [(1024, 128, 512), (4096, 512, 512), (8192, 512, 2048), (8192, 2048, 512),
                                   (2048, 512, 256), (2048, 256, 64), (1000, 512, 512), (37, 64, 32), (65536, 512, 512)])
def test_linear_matches_fp64_gemm(M, K, N):
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)
    b = torch.randn(N, device="cuda", generator=g)
    lin = HipLinear(w, b)
    y = lin(x)
    ref = _ref(x, w, b)
    assert y.shape == (M, N)
    e_hip = rel_err(y.cpu().numpy(), ref.cpu().numpy())
    e_f32 = rel_err(torch.nn.functional.linear(x, w, b).cpu().numpy(), ref.cpu().numpy())
    assert e_hip < TOL, (e_hip, e_f32)
    assert e_hip < 2e-5, (e_hip, e_f32)          # split-bf16 drops lo*lo (~2^-17 per product): a few 1e-6 in practice


def test_linear_epilogues_gelu_residual_periodic_add():
    g = torch.Generator(device="cuda").manual_seed(7)
    B, L, D = 3, 1024, 512
    x = torch.randn(B, L, D, device="cuda", generator=g)
    w1 = torch.randn(4 * D, D, device="cuda", generator=g) / math.sqrt(D)
    b1 = torch.randn(4 * D, device="cuda", generator=g)
    w2 = torch.randn(D, 4 * D, device="cuda", generator=g) / math.sqrt(4 * D)
    b2 = torch.randn(D, device="cuda", generator=g)
    h = HipLinear(w1, b1)(x, activation="gelu")
    assert rel_err(h.cpu().numpy(), _ref(x, w1, b1, act="gelu").cpu().numpy()) < TOL
    y = HipLinear(w2, b2)(h, residual=x)
    assert rel_err(y.cpu().numpy(), _ref(h, w2, b2, residual=x).cpu().numpy()) < TOL
    # in-place residual (y aliases the residual): the pre-LN block's x = x + sublayer(x)
    xc = x.clone()
    HipLinear(w2, b2)(h, residual=xc, out=xc)
    assert torch.equal(xc, y)
    # row-periodic addend (chaos term folded into Q: 5 distinct rows tiled along the sequence)
    wq = torch.randn(D, D, device="cuda", generator=g) / math.sqrt(D)
    padd = torch.randn(B, 5, D, device="cuda", generator=g)
    q = HipLinear(wq, None)(x, periodic_add=padd)
    assert rel_err(q.cpu().numpy(), _ref(x, wq, None, padd=padd).cpu().numpy()) < TOL


def test_linear_strided_rows_and_unsupported_shapes():
    g = torch.Generator(device="cuda").manual_seed(11)
    big = torch.randn(512, 1536, device="cuda", generator=g)
    x = big[:, 512:1024]                                          # row pitch 1536, 16-byte aligned rows
    w = torch.randn(512, 512, device="cuda", generator=g) / 22.6
    y = HipLinear(w)(x)
    assert rel_err(y.cpu().numpy(), _ref(x, w, None).cpu().numpy()) < TOL
    assert not hip_linear_supported(3, 512) and not hip_linear_supported(512, 3)
    with pytest.raises(Exception, match="HIP path is built for"):
        HipLinear(torch.randn(3, 512, device="cuda"))
    with pytest.raises(Exception):
        HipLinear(torch.randn(64, 64))                            # CPU tensor: no CPU path


def test_linear_split_bf16_input_and_output_formats():
    """SMK_FMT_SPLIT_BF16 on either side of a layer: a pre-split x gives the same result as fp32 x (the kernel splits fp32
    the same way; 1e-6: only the K summation order may differ), a split y decodes (hi + lo) to the fp32 y within the
    format's resolution (2^-17 relative)."""
    from smokephysai_amd.models.linear import to_split, from_split
    g = torch.Generator(device="cuda").manual_seed(3)
    for (B, L, K, N, act) in [(2, 1024, 512, 2048, "gelu"), (1, 256, 2048, 512, None), (3, 128, 128, 384, None), (1, 40, 64, 32, "relu")]:
        x = torch.randn(B, L, K, device="cuda", generator=g)
        w = torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)
        b = torch.randn(N, device="cuda", generator=g)
        lin = HipLinear(w, b)
        y = lin(x, activation=act)
        xs = to_split(x)
        assert xs.shape == (B, L, K // 8, 2, 8) and xs.dtype == torch.bfloat16
        assert (from_split(xs) - x).abs().max().item() <= x.abs().max().item() * 2.0 ** -16
        y_from_split = lin(xs, activation=act, x_split=True)
        # same operand values; the summation order over K can differ (the fp32-input form may split K across wave groups)
        assert rel_err(y_from_split.cpu().numpy(), y.cpu().numpy()) < 1e-6
        ys = lin(x, activation=act, out_split=True)
        assert ys.shape == (B, L, N // 8, 2, 8)
        assert rel_err(from_split(ys).cpu().numpy(), y.cpu().numpy()) < 2.0 ** -16
        both = lin(xs, activation=act, x_split=True, out_split=True)
        assert rel_err(from_split(both).cpu().numpy(), from_split(ys).cpu().numpy()) < 2.0 ** -15      # two decodes of the format
        res = torch.randn(B, L, N, device="cuda", generator=g)
        assert rel_err(lin(xs, residual=res, x_split=True).cpu().numpy(), lin(x, residual=res).cpu().numpy()) < 1e-6


# ---------------------------------------------------------------- training: forward + input gradient on the HIP kernel
@pytest.mark.parametrize("M,K,N", [(4096, 512, 512), (2048, 512, 2048), (2048, 2048, 512), (1024, 128, 512), (512, 256, 64)])
def test_trainable_linear_forward_and_gradients_match_fp64(M, K, N):
    """nn.Linear's autograd contract (y, dX = dY W, dW = dY^T X, db = sum dY) with y and dX on smk_linear_forward; after an
    in-place parameter update (an optimizer step) both device mirrors are re-split (smk_linear_update)."""
    from smokephysai_amd.models.linear import TrainableHipLinear
    torch.manual_seed(M + K + N)
    lin = TrainableHipLinear(K, N).cuda()
    lin.hip_train = True
    for rnd in range(2):
        x = torch.randn(2, M // 2, K, device="cuda", requires_grad=True)
        dy = torch.randn(2, M // 2, N, device="cuda")
        y = lin(x)
        assert y.grad_fn is not None and "HipLinearFn" in type(y.grad_fn).__name__
        y.backward(dy)
        xd = x.detach().double().requires_grad_(True)
        wd = lin.weight.detach().double().requires_grad_(True)
        bd = lin.bias.detach().double().requires_grad_(True)
        yd = torch.nn.functional.linear(xd, wd, bd)
        yd.backward(dy.double())
        def err(a, b):
            return rel_err(a.detach().cpu().numpy(), b.detach().cpu().numpy())
        assert err(y, yd) < TOL and err(x.grad, xd.grad) < TOL
        assert err(lin.weight.grad, wd.grad) < TOL and err(lin.bias.grad, bd.grad) < TOL
        assert err(y, yd) < 2e-5 and err(x.grad, xd.grad) < 2e-5        # measured ~1e-6
        with torch.no_grad():                                                     # "optimizer step": bumps the parameter versions
            lin.weight.add_(0.05 * torch.randn_like(lin.weight))
            lin.bias.mul_(0.5)
        lin.zero_grad()


def test_trainable_linear_falls_back_to_f_linear_when_not_applicable():
    from smokephysai_amd.models.linear import TrainableHipLinear
    lin = TrainableHipLinear(3, 512).cuda()            # chaos_proj-like shape: not a kernel shape
    lin.hip_train = True
    x = torch.randn(8, 3, device="cuda", requires_grad=True)
    assert "HipLinearFn" not in type(lin(x).grad_fn).__name__
    lin2 = TrainableHipLinear(512, 512).cuda()
    lin2.hip_train = True
    with torch.no_grad():
        y = lin2(torch.randn(64, 512, device="cuda"))
    assert y.grad_fn is None
    lin2.hip_train = False
    assert "HipLinearFn" not in type(lin2(x.new_zeros(4, 512).requires_grad_()).grad_fn).__name__


@pytest.mark.parametrize("rows,out_f,in_f", [(65536, 512, 512), (8192, 2048, 512), (8192, 512, 2048), (1000, 64, 256),
                                             (70001, 256, 128), (300, 36, 32)])
def test_weight_gradient_matches_fp64(rows, out_f, in_f):
    """smk_linear_wgrad: dW = dY^T X over the token rows (K-segmented launch + ordered partial sum), ragged row counts, row-strided
    inputs; deterministic from run to run."""
    from smokephysai_amd.models.linear import hip_linear_wgrad
    g = torch.Generator(device="cuda").manual_seed(rows + out_f + in_f)
    dy_full = torch.randn(rows, out_f + 8, device="cuda", generator=g)
    x_full = torch.randn(rows, in_f + 4, device="cuda", generator=g)
    dy, x = dy_full[:, :out_f], x_full[:, :in_f]                     # row pitch > feature count
    dw = hip_linear_wgrad(dy, x)
    ref = dy.double().t() @ x.double()
    e_hip = rel_err(dw.cpu().numpy(), ref.cpu().numpy())
    assert dw.shape == (out_f, in_f) and e_hip < TOL and e_hip < 2e-5, e_hip
    assert torch.equal(dw, hip_linear_wgrad(dy, x))
    # the bias gradient rides on the call's transposed copy of dY
    dw2, db = hip_linear_wgrad(dy, x, want_db=True)
    assert torch.equal(dw2, dw) and db.shape == (out_f,)
    ref_b = dy.double().sum(0)
    assert float((db.double() - ref_b).abs().max()) <= 2e-6 * float(dy.abs().sum(0).max())


def test_weight_gradient_row_chunking(monkeypatch):
    import smokephysai_amd.models.linear as hl
    g = torch.Generator(device="cuda").manual_seed(3)
    dy = torch.randn(5000, 128, device="cuda", generator=g)
    x = torch.randn(5000, 64, device="cuda", generator=g)
    whole = hl.hip_linear_wgrad(dy, x)
    monkeypatch.setattr(hl, "MAX_WGRAD_ROWS", 2048)
    parts = hl.hip_linear_wgrad(dy, x)
    ref = dy.double().t() @ x.double()
    assert rel_err(parts.cpu().numpy(), ref.cpu().numpy()) < 2e-5 and rel_err(whole.cpu().numpy(), ref.cpu().numpy()) < 2e-5
    _, db = hl.hip_linear_wgrad(dy, x, want_db=True)                # chunked: per-chunk column sums added
    assert float((db.double() - dy.double().sum(0)).abs().max()) < 1e-3


@pytest.mark.parametrize("rows,K,N,act,padd", [(1024, 512, 1536, None, True), (1024, 512, 2048, "gelu", False), (96, 128, 64, "relu", False),
                                                (1000, 512, 512, None, False),
                                                # several tiles per workgroup (the statistics follow the chunk stream across tiles): batch 4 and
                                                # batch 64 of the body's two fused layers, a one-chunk K, a ragged row count on the 128-row tiles
                                                (4096, 512, 1536, None, True), (4096, 512, 2048, "gelu", False), (65536, 512, 1536, None, True),
                                                (65536, 512, 2048, "gelu", False), (40000, 64, 256, None, False), (33001, 512, 512, "relu", False)])
def test_layernorm_fused_into_the_linear_layer(rows, K, N, act, padd):
    """smk_linear_forward_ln (HipLinearLN): act(LayerNorm(x) W^T + b + periodic_add) against the same chain in fp64 -- rows whose mean is
    far from zero (the epilogue's mean * wsum correction carries weight), a ragged last tile, the periodic addend of the q | k | v layer."""
    from smokephysai_amd.models.linear import HipLinearLN
    g = torch.Generator(device="cuda").manual_seed(rows + K + N)
    x = torch.randn(rows, K, device="cuda", generator=g) * 1.7 + torch.randn(rows, 1, device="cuda", generator=g) * 1.5      # per-row offsets
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    gamma = torch.rand(K, device="cuda", generator=g) + 0.5
    beta = torch.randn(K, device="cuda", generator=g) * 0.3
    lin = HipLinearLN(w, b, gamma, beta, 1e-5)
    assert lin.max_rows >= rows
    rpg = (1024 if rows % 1024 == 0 and rows > 1024 else 32) if padd else 0      # (1,024: the body's tokens per frame -> 128-row tiles)
    pa = torch.randn(rows // rpg, 5, N, device="cuda", generator=g) if padd else None
    y = lin.forward_ln(x, activation=act, periodic_add=pa, rows_per_group=rpg)
    h = torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5)
    ref = h @ w.double().t() + b.double()
    if padd:
        idx = torch.arange(rows, device="cuda")
        ref = ref + pa.double()[idx // rpg, (idx % rpg) % 5]
    if act == "gelu":
        ref = torch.nn.functional.gelu(ref)
    elif act == "relu":
        ref = torch.relu(ref)
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 2e-5
    lin.max_rows = 64                                                   # (the library's own limit is the 32-bit offset range: not allocatable here)
    with pytest.raises(ValueError):
        lin.forward_ln(torch.zeros(96, K, device="cuda"))


@pytest.mark.parametrize("rows", [1024, 2048, 4096, 16384])
def test_fused_layer_writes_its_k_v_columns_as_inplace_split_bf16(rows):
    """smk_linear_forward_ln_split (HipLinearLN.forward_ln(split_from=D)): the q columns are the fp32 values of the unsplit call bit for bit,
    the k | v columns hold SMK_FMT_SPLIT4_INPLACE -- exactly the {hi, lo} pairs the attention kernel would have formed from the fp32 values
    (so the two routes multiply identical numbers).  Row counts that take each of the layer's kernels (32- and 64-row tiles on k_linear_x3,
    128-row tiles on k_linear_b16), with the periodic addend of the q | k | v layer."""
    from smokephysai_amd.models.linear import HipLinearLN, split4_inplace, unsplit4_inplace
    K, D = 512, 512
    g = torch.Generator(device="cuda").manual_seed(rows)
    x = torch.randn(rows, K, device="cuda", generator=g) * 1.3 + torch.randn(rows, 1, device="cuda", generator=g)
    lin = HipLinearLN(torch.randn(3 * D, K, device="cuda", generator=g) / K ** 0.5, torch.randn(3 * D, device="cuda", generator=g),
                      torch.rand(K, device="cuda", generator=g) + 0.5, torch.randn(K, device="cuda", generator=g) * 0.3, 1e-5)
    pa = torch.zeros(rows // 1024, 5, 3 * D, device="cuda")
    pa[:, :, :D] = torch.randn(rows // 1024, 5, D, device="cuda", generator=g)
    plain = lin.forward_ln(x, periodic_add=pa, rows_per_group=1024)
    mixed = lin.forward_ln(x, periodic_add=pa, rows_per_group=1024, split_from=D)
    assert torch.equal(mixed[:, :D], plain[:, :D])
    want = split4_inplace(plain[:, D:].contiguous())
    assert torch.equal(mixed[:, D:].contiguous().view(torch.int32), want.view(torch.int32))
    dec = unsplit4_inplace(mixed[:, D:].contiguous())
    assert float((dec - plain[:, D:]).abs().max()) <= 2.0 ** -16 * float(plain[:, D:].abs().max())
    with pytest.raises(Exception, match="split_from_col"):
        lin.forward_ln(x, split_from=520)


@pytest.mark.parametrize("rows", [1024, 4096])
def test_fused_layernorm_rows_whose_mean_dwarfs_their_spread(rows):
    """Rows like 50 + 0.1 randn (|mean| = 500 sigma): a one-pass E[x^2] - mean^2 loses its digits there.  The kernel gathers shifted sums
    about a per-thread pivot and merges them pairwise, and its epilogue's rstd (x W'^T - mean wsum) subtracts two large terms -- the result
    must still sit within 1e-4 of LayerNorm + Linear in fp64."""
    from smokephysai_amd.models.linear import HipLinearLN
    K, N = 512, 512
    g = torch.Generator(device="cuda").manual_seed(rows)
    x = 50.0 + 0.1 * torch.randn(rows, K, device="cuda", generator=g)
    x[::7] = -300.0 + 0.01 * torch.randn(x[::7].shape, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    gamma = torch.rand(K, device="cuda", generator=g) + 0.5
    beta = torch.randn(K, device="cuda", generator=g) * 0.3
    y = HipLinearLN(w, b, gamma, beta, 1e-5).forward_ln(x)
    ref = torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5) @ w.double().t() + b.double()
    # the statistics themselves: recover rstd from the output scale -- and the end result
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 1e-4
