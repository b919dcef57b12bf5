"""Parity of the 3-D stepper (csrc/stencil3d.hip, BASELINE configs[4]) through the C ABI (smk_sim3d_*) against oracle/ns_nd.py, the
executable form of SPEC_3D.md -- the same code whose 2-D instance tests/test_oracle_golden.py holds bit-exact to the reference's fixtures
(the reference itself is 2-D only: navier_stokes.py:10,21).  Bit-exact everywhere (fp32 index and stencil work in the oracle's operation
order); the full-size case (8 grids of 512 x 512 x 64) uses a size-independent property: for one time step a compact source evolves exactly
like the same source in a small grid that contains its neighbourhood (the pressure's support grows one cell per Jacobi sweep)."""
import numpy as np
import pytest
import torch

from oracle.ns_nd import OracleNSnd

pytestmark = pytest.mark.gpu

from smokephysai_amd.physics import NavierStokesSimulator3D                   # noqa: E402
from smokephysai_amd.physics import navier_stokes3d as ns3                    # noqa: E402

KEYS = ("u", "v", "w", "p", "density")


def _random_state(shape, seed, vel_scale):
    """A dense random state: velocities large enough that back-traces leave the cell (and, at vel_scale 300, cross several cells)."""
    rng = np.random.default_rng(seed)
    o = OracleNSnd(shape)
    for k in ("u", "v", "w"):
        setattr(o, k, (rng.standard_normal(getattr(o, k).shape) * vel_scale).astype(np.float32))
    o.p = (rng.standard_normal(o.p.shape) * 0.1).astype(np.float32)
    o.density = rng.random(o.density.shape).astype(np.float32) * 2.0
    return o


def _to_device(sim, o, b=None):
    for k in KEYS:
        cur = getattr(sim, k)
        if b is None:
            setattr(sim, k, torch.from_numpy(getattr(o, k)))
        else:
            cur[b].copy_(torch.from_numpy(getattr(o, k)))


def _assert_equal(sim, o, tag, b=None):
    for k in KEYS:
        got = getattr(sim, k) if b is None else getattr(sim, k)[b]
        np.testing.assert_array_equal(got.cpu().numpy(), getattr(o, k), err_msg=f"{tag}: {k}")


@pytest.mark.parametrize("shape,vel", [((16, 24, 32), 3.0), ((13, 22, 19), 300.0), ((32, 32, 32), 30.0)])
def test_every_stage_bit_exact_vs_oracle(shape, vel):
    """buoyancy + 4 diffusions, projection (divergence, 20 sweeps, gradient), the four sequentially dependent advections (+ decay):
    after EACH stage every field equals the oracle's, bit for bit -- W % 4 == 0 (16-byte Jacobi form) and an odd shape (scalar form)."""
    o = _random_state(shape, seed=sum(shape), vel_scale=vel)
    sim = NavierStokesSimulator3D(shape)
    _to_device(sim, o)
    o.buoyancy(); o.diffuse_all()
    sim.run_stage(ns3.STAGE3D_BUOY_DIFFUSE)
    _assert_equal(sim, o, "buoy+diffuse")
    o.pressure_projection()
    sim.run_stage(ns3.STAGE3D_PROJECT)
    _assert_equal(sim, o, "project")
    for stage, name in ((ns3.STAGE3D_ADVECT_U, "u"), (ns3.STAGE3D_ADVECT_V, "v"), (ns3.STAGE3D_ADVECT_W, "w")):
        setattr(o, name, o.advection_step(getattr(o, name), o.velocity()))
        sim.run_stage(stage)
        _assert_equal(sim, o, f"advect {name}")
    o.density = o.advection_step(o.density, o.velocity()) * np.float32(0.995)
    sim.run_stage(ns3.STAGE3D_ADVECT_D)
    _assert_equal(sim, o, "advect density")
    assert np.abs(o.u).max() > 0 and np.isfinite(o.density).all()


@pytest.mark.parametrize("shape,J,steps", [((32, 32, 32), 20, 12), ((16, 64, 64), 20, 10), ((9, 20, 28), 7, 6)])
def test_batched_trajectories_bit_exact_vs_per_grid_oracle(shape, J, steps):
    """B = 3 different grids stepped together (sources of 1-3 balls each) == three single-grid oracle runs: u, v, w, p, density after
    `steps` steps and every emitted frame, bit for bit; sources: mask exact, values within the device expf's 2 ulp."""
    D, H, W = shape
    B = 3
    rng = np.random.default_rng(7 * D + H)
    srcs = []
    for b in range(B):
        for _ in range(1 + b):
            srcs.append((b, int(rng.integers(4, W - 4)), int(rng.integers(4, H - 4)), int(rng.integers(2, D - 2)),
                         int(rng.integers(2, 6)), float(rng.uniform(0.5, 2.0))))
    sim = NavierStokesSimulator3D(shape, batch_size=B, jacobi_iters=J)
    sim.add_smoke_sources(srcs)
    init = sim.density.cpu().numpy().copy()
    frames = torch.empty(B, steps, D, H, W, device="cuda")
    sim.step_into(frames, steps)
    torch.cuda.synchronize()
    fr = frames.cpu().numpy()
    for b in range(B):
        o = OracleNSnd(shape, jacobi_iters=J)
        for (g, x, y, z, r, inten) in srcs:
            if g == b:
                o.add_smoke_source(x, y, z, radius=r, intensity=inten)
        assert np.array_equal(o.density != 0, init[b] != 0)
        assert np.abs(o.density - init[b]).max() <= 1e-6 * max(1.0, np.abs(init[b]).max())
        o.density = init[b].copy()                    # (sources enter through the device expf: continue from the SAME density)
        for t in range(steps):
            out = o.step()
            np.testing.assert_array_equal(fr[b, t], out, err_msg=f"grid {b} frame {t}")
        _assert_equal(sim, o, f"grid {b}", b)
        # SPEC_3D.md section 5: last row and last column of every advected field are 0; the Jacobi shell of p is 0
        for k in ("u", "v", "w", "density"):
            f = getattr(sim, k)[b]
            assert not f[:, -1].any() and not f[:, :, -1].any(), k
        p = sim.p[b]
        assert not p[0].any() and not p[-1].any() and not p[:, 0].any() and not p[:, -1].any() and not p[:, :, 0].any() and not p[:, :, -1].any()
    assert np.abs(sim.w.cpu().numpy()).max() > 0


@pytest.mark.parametrize("shape,vel,J", [((13, 22, 19), 0.5, 6), ((13, 22, 19), 300.0, 6), ((4, 16, 64), 150.0, 4), ((6, 40, 130), 40.0, 8),
                                         ((6, 40, 130), 400.0, 8), ((5, 35, 64), 2.0, 5)])
def test_whole_steps_from_dense_random_states_bit_exact_vs_oracle(shape, vel, J):
    """smk_sim3d_step -- the z-marching launches: buoyancy + diffusion + divergence, the blocked Jacobi, gradient subtraction + the four
    advections -- from DENSE random states of two grids, two steps, against the oracle's step().  vel 0.5-40: every cell moves, every
    back-trace stays in its 2 x 2 x 2 LDS neighbourhood (the fast form, with non-zero weights everywhere); vel 150-400: dt * velocity
    reaches several cells, so whole batches of units leave through far_value3 (general form on global memory, gradient re-applied per
    tap) while others of the same launch stay on the fast form.  Shapes: odd everything; exactly one tile; three x-tiles with a 2-wide last
    one and rows that do not fill the last wave; H not a multiple of the rows per wave."""
    D, H, W = shape
    B = 2
    sim = NavierStokesSimulator3D(shape, batch_size=B, jacobi_iters=J)
    os_ = [_random_state(shape, 100 * D + W + b, vel) for b in range(B)]
    for b in range(B):
        os_[b].jacobi_iters = J
        _to_device(sim, os_[b], b)
    steps = 2
    frames = torch.empty(B, steps, D, H, W, device="cuda")
    sim.step_into(frames, steps)
    torch.cuda.synchronize()
    fr = frames.cpu().numpy()
    for b in range(B):
        o = os_[b]
        for t in range(steps):
            out = o.step()
            np.testing.assert_array_equal(fr[b, t], out, err_msg=f"grid {b} frame {t}")
        _assert_equal(sim, o, f"grid {b}", b)
        assert np.isfinite(o.density).all() and np.abs(o.u).max() > 0


def test_reset_of_a_grid_subset_and_loud_failures():
    sim = NavierStokesSimulator3D((8, 16, 16), batch_size=3)
    sim.add_smoke_source(8, 8, 4, radius=3, intensity=1.0)
    sim.step_into(None, 2)
    ref = {k: getattr(sim, k).clone() for k in KEYS}
    sim.setup_grid(grids=[1])
    for k in KEYS:
        f = getattr(sim, k)
        assert not f[1].any() and torch.equal(f[0], ref[k][0]) and torch.equal(f[2], ref[k][2]), k
    with pytest.raises(ValueError):
        sim.step_into(torch.empty(3, 8, 16, 16, device="cuda", dtype=torch.float64))
    with pytest.raises(ValueError):
        sim.step_into(torch.empty(3, 2, 8, 16, 16, device="cuda"), 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        NavierStokesSimulator3D((8, 16, 16), device="cpu")
    with pytest.raises(ValueError):
        NavierStokesSimulator3D((16, 16))


def test_config4_full_size_step_vs_small_grid_oracle():
    """BASELINE configs[4] as stated: 8 grids of 512 x 512 x 64 (D = 64), Jacobi-20, one step.  In one step nothing travels farther than
    radius + 1 (diffusion) + 20 (the pressure's support grows one cell per sweep) + 2 cells, so a compact source evolves exactly as in a
    small grid that contains that neighbourhood.  Every grid gets two sources that cannot interact within the step:
      * one inside the corner window [0,128) x [0,128) (walls y = 0 / x = 0 are the real walls): the window must equal the oracle's step
        on a 64 x 128 x 128 grid BIT FOR BIT (same index values, so the same roundings);
      * one far away, at a different place in every grid: its 96 x 96 window against the oracle's 64 x 96 x 96 grid translated there --
        the back-trace `index - dt * velocity` rounds differently at index 463 than at 48, so this one is held to 1e-5 of the field's
        maximum (the path's float bar is 1e-4), not to equality;
    and every cell outside the two windows must still be exactly 0."""
    D, H, W, B, R, RN = 64, 512, 512, 8, 96, 128
    sim = NavierStokesSimulator3D((D, H, W), batch_size=B, jacobi_iters=20)
    far = [(300, 180), (463, 48), (48, 463), (463, 463), (256, 256), (100, 411), (333, 77), (200, 300)]        # (x, y)
    near = [(40 + 6 * b, 90 - 7 * b) for b in range(B)]
    srcs = []
    for b in range(B):
        srcs.append((b, near[b][0], near[b][1], 12 + 5 * b, 5, 1.0 + 0.1 * b))
        srcs.append((b, far[b][0], far[b][1], 20 + 3 * b, 5, 1.5 - 0.1 * b))
    sim.add_smoke_sources(srcs)
    d_near = [sim.density[b, :, :RN, :RN].cpu().numpy().copy() for b in range(B)]
    d_far = [sim.density[b, :, cy - 48:cy + 48, cx - 48:cx + 48].cpu().numpy().copy() for b, (cx, cy) in enumerate(far)]
    frame = torch.empty(B, D, H, W, device="cuda")
    sim.step_into(frame, 1)
    torch.cuda.synchronize()
    assert torch.isfinite(frame).all()

    def window(b, y0, x0, r):
        return {"u": sim.u[b, :, y0:y0 + r + 1, x0:x0 + r], "v": sim.v[b, :, y0:y0 + r, x0:x0 + r + 1], "w": sim.w[b, :, y0:y0 + r, x0:x0 + r],
                "p": sim.p[b, :, y0:y0 + r, x0:x0 + r], "density": sim.density[b, :, y0:y0 + r, x0:x0 + r]}

    for b in range(B):
        o = OracleNSnd((D, RN, RN), jacobi_iters=20)
        o.density = d_near[b]
        out = o.step()
        wn = window(b, 0, 0, RN)
        for k in KEYS:
            got = wn[k].cpu().numpy()
            ref = getattr(o, k)
            # the oracle's own high walls (row / column 128) force zeros the big grid does not have there; the source's reach ends far below
            np.testing.assert_array_equal(got[:, :RN - 1, :RN - 1], ref[:, :RN - 1, :RN - 1], err_msg=f"grid {b} near {k}")
        np.testing.assert_array_equal(frame[b, :, :RN - 1, :RN - 1].cpu().numpy(), out[:, :RN - 1, :RN - 1])
        cx, cy = far[b]
        y0, x0 = cy - 48, cx - 48
        o = OracleNSnd((D, R, R), jacobi_iters=20)
        o.density = d_far[b]
        o.step()
        wf = window(b, y0, x0, R)
        for k in KEYS:
            got, ref = wf[k].cpu().numpy(), getattr(o, k)
            assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max(), (b, k, np.abs(got - ref).max(), np.abs(ref).max())
        for k in KEYS:                                 # nothing outside the two windows
            f = getattr(sim, k)[b]
            m = torch.ones_like(f, dtype=torch.bool)
            m[:, :wn[k].shape[1], :wn[k].shape[2]] = False
            m[:, y0:y0 + wf[k].shape[1], x0:x0 + wf[k].shape[2]] = False
            assert not f[m].any(), (b, k)
    assert float(sim.w.abs().max()) > 0 and float(sim.p.abs().max()) > 0
