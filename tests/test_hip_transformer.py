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
