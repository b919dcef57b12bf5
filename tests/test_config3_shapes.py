"""Parity at BASELINE configs[2]'s REAL shapes: 256 x 256 grids, batch 64, Jacobi-100, 200 time steps; the
persistent encoder kernels over all 64 frames of 256 x 256; and the exact step bench.py times (step_into -> enc.tokens).

The oracle is the scalar C restatement (Jacobi-100 has no reference counterpart: the reference hard-codes 20 sweeps,
navier_stokes.py:139; the restatement is pinned bit-exact to the reference at J=20 by tests/test_oracle_golden.py)."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from conftest import rel_err

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from smokephysai_amd.models.encoder import HipEncoder         # noqa: E402
from smokephysai_amd.physics import SmokeSimulator            # noqa: E402

B, N, J, STEPS = 64, 256, 100, 200
KEYS = ("u", "v", "p", "density")


def _bench_sources(seed=0):
    import bench
    return bench.draw_sources(B, N, seed)


def _oracle_grid(srcs, b, steps, add_fractal):
    """Grid b of the batch alone on the oracle: same sources in the same order, `steps` time steps; returns (sim, last frame)."""
    o = oracle.OracleSmokeSimulator((N, N), jacobi_iters=J, cache_fractal=True)
    for (g, x, y, r, inten) in srcs:
        if g == b:
            o.ns_solver.add_smoke_source(x, y, r, inten)
    frame = None
    for _ in range(steps):
        frame = o.simulate_step(add_fractal=add_fractal)
    return o, frame


def test_200_steps_jacobi100_batch64_bit_exact_vs_oracle():
    """configs[2] as stated: 64 grids of 256^2, 200 steps, Jacobi-100.  Three grids (first, one in the middle, last -- different
    workgroup bands / batch offsets) are compared BIT-EXACT with the oracle after all 200 steps: u, v, p, density."""
    srcs = _bench_sources(0)
    sim = SmokeSimulator((N, N), batch_size=B, jacobi_iters=J)
    init = None
    sim.ns_solver.add_smoke_sources(srcs)
    init = {b: sim.ns_solver.density[b].cpu().numpy().copy() for b in (0, 29, 63)}
    frame = torch.empty(B, N, N, device="cuda")
    for _ in range(STEPS):
        sim.ns_solver.step_into(frame, 1, add_fractal=False)
    torch.cuda.synchronize()
    assert torch.isfinite(frame).all()
    for b in (0, 29, 63):
        o = oracle.OracleNS((N, N), jacobi_iters=J)
        o.density = init[b].copy()            # sources enter through the device expf (<= 2 ulp from libm): start from the SAME density
        for _ in range(STEPS):
            last = o.step()
        for k in KEYS:
            np.testing.assert_array_equal(getattr(sim.ns_solver, k)[b].cpu().numpy(), getattr(o, k), err_msg=f"grid {b} {k}")
        np.testing.assert_array_equal(frame[b].cpu().numpy(), last, err_msg=f"grid {b} frame")
    # and the whole trajectory from the oracle's own sources (libm expf) stays inside the float bar after 200 steps
    o, _ = _oracle_grid(srcs, 7, STEPS, add_fractal=False)
    for k in KEYS:
        assert rel_err(getattr(sim.ns_solver, k)[7].cpu().numpy(), getattr(o.ns_solver, k)) < 1e-4, k


def test_every_grid_of_the_batch_20_steps_jacobi100_bit_exact_vs_oracle():
    """ALL 64 grids (the persistent launch maps bands to grids through an XCD permutation, stencil.hip k_jacobi_band: a wrong map would
    leave some grid unchecked by a three-grid sample): 20 steps at Jacobi-100 with the fractal frame emit, then u, v, p, density and the
    last frame of every grid against the per-grid oracle -- compared as whole arrays, so a single differing word fails."""
    from concurrent.futures import ThreadPoolExecutor
    steps = 20
    srcs = _bench_sources(5)
    sim = SmokeSimulator((N, N), batch_size=B, jacobi_iters=J)
    sim.ns_solver.add_smoke_sources(srcs)
    init = sim.ns_solver.density.cpu().numpy().copy()
    assert sim.ns_solver.jacobi_plan()["projection"].get("persistent") is True
    frames = sim.simulate_sequence(steps, add_fractal=True)
    state = {k: getattr(sim.ns_solver, k).cpu().numpy() for k in KEYS}
    last = frames[:, -1].cpu().numpy()

    def one(b):
        o = oracle.OracleSmokeSimulator((N, N), jacobi_iters=J, cache_fractal=True)
        o.ns_solver.density = init[b].copy()      # (sources enter through the device expf: start from the SAME density)
        fr = None
        for _ in range(steps):
            fr = o.simulate_step(add_fractal=True)
        bad = [k for k in KEYS if not np.array_equal(state[k][b], getattr(o.ns_solver, k))]
        ferr = rel_err(last[b], fr)               # the fractal constant's sin / cos differ by ulps between libm and the device
        return b, bad, ferr

    with ThreadPoolExecutor(8) as ex:             # (the C oracle runs outside the GIL)
        res = list(ex.map(one, range(B)))
    assert [(b, bad) for b, bad, _ in res if bad] == []
    assert max(f for _, _, f in res) < 1e-6
    assert len({state["p"][b].tobytes() for b in range(B)}) > B // 2      # the grids really differ (no accidental broadcast)


@pytest.fixture(scope="module")
def frames64():
    """64 emitted frames of 256^2 after 12 steps (with the fractal multiplier), as the bench produces them."""
    sim = SmokeSimulator((N, N), batch_size=B, jacobi_iters=20)
    sim.ns_solver.add_smoke_sources(_bench_sources(3))
    frame = torch.empty(B, N, N, device="cuda")
    for _ in range(12):
        sim.ns_solver.step_into(frame, 1, add_fractal=True, fractal_intensity=0.05)
    torch.cuda.synchronize()
    return frame


@pytest.fixture(scope="module")
def bench_weights():
    import bench
    return bench.encoder_weights(0)


@pytest.mark.parametrize("dtype,tol_f32,tol_oracle", [("bf16x3", 1e-5, 1e-4), ("i8x3", 1e-4, 1e-4)])
def test_encoder_batch64_256_all_frames(frames64, bench_weights, dtype, tol_f32, tol_oracle):
    """The persistent MFMA encoder kernels at their real launch shape (64 frames x 256^2 = 8,192 tiles walked by 512 workgroups):
    every frame against the fp32-MFMA kernel, and the first / middle / last frame against the fp64-accumulating oracle."""
    enc = HipEncoder(bench_weights, device="cuda")
    tok = enc.tokens(frames64, input_dim=128, dtype=dtype)                  # [B,1024,128] token-major (what the bench times)
    f32 = enc(frames64, input_dim=128, dtype="f32")                          # [B,128,32,32]
    torch.cuda.synchronize()
    got = tok.transpose(1, 2).reshape(B, 128, 32, 32).cpu().numpy()
    ref32 = f32.cpu().numpy()
    for b in range(B):
        assert rel_err(got[b], ref32[b]) < tol_f32, f"frame {b} vs fp32 kernel"
    w = {k: v.cpu().numpy() for k, v in bench_weights.items()}
    fr = frames64.cpu().numpy()
    for b in (0, 31, 63):
        ref = oracle.encoder_features(fr[b:b + 1], w, input_dim=128)[0]
        assert rel_err(got[b], ref) < tol_oracle, f"frame {b} vs oracle"
        assert rel_err(ref32[b], ref) < 2e-5, f"fp32 kernel frame {b} vs oracle"
    # NCHW entry point of the same kernel: same numbers
    nchw = enc(frames64, input_dim=128, dtype=dtype)
    assert torch.equal(nchw, tok.transpose(1, 2).reshape(B, 128, 32, 32))


def test_bench_step_pipeline_outputs(bench_weights):
    """Exactly bench.py's timed step -- step_into(frame, 1, fractal) then enc.tokens(frame) on the same stream, every step, no
    synchronisation in between -- for 6 steps at configs[2]'s shape: the emitted frames of three grids are bit-identical to the
    oracle's, and the tokens of the LAST step (computed from a frame buffer that was overwritten every step) match the oracle's
    features of that frame."""
    srcs = _bench_sources(0)
    sim = SmokeSimulator((N, N), batch_size=B, jacobi_iters=J)
    sim.ns_solver.add_smoke_sources(srcs)
    init = {b: sim.ns_solver.density[b].cpu().numpy().copy() for b in (0, 40, 63)}
    enc = HipEncoder(bench_weights, device="cuda")
    frame = torch.empty(B, N, N, device="cuda")
    toks = []
    for _ in range(6):
        sim.ns_solver.step_into(frame, 1, add_fractal=True, fractal_intensity=0.05)
        toks.append(enc.tokens(frame, input_dim=128, dtype="bf16x3"))
    torch.cuda.synchronize()
    w = {k: v.cpu().numpy() for k, v in bench_weights.items()}
    F = oracle.fractal_field(N, N)
    for b in (0, 40, 63):
        o = oracle.OracleNS((N, N), jacobi_iters=J)
        o.density = init[b].copy()
        for _ in range(6):
            d = o.step()
        emitted = oracle.apply_fractal_perturbation(d, 0.05, fractal=F)
        np.testing.assert_array_equal(sim.ns_solver.density[b].cpu().numpy(), o.density, err_msg=f"grid {b} state")
        assert rel_err(frame[b].cpu().numpy(), emitted) < 1e-6, f"grid {b} emitted frame"      # perlin sinf/cosf: <= 2 ulp
        ref = oracle.encoder_features(frame[b:b + 1].cpu().numpy(), w, input_dim=128)[0]
        got = toks[-1][b].T.reshape(128, 32, 32).cpu().numpy()
        assert rel_err(got, ref) < 1e-4, f"grid {b} tokens"
    assert not torch.equal(toks[0], toks[-1])
