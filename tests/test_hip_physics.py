"""GPU parity of the HIP stencil path (through the C ABI) against the CPU oracle and the reference's golden vectors.
Velocity / pressure / density grids and gather indices are compared BIT-EXACT (BASELINE.json: "bit-exact for index
ops"; the stencil kernels round once per reference op, so float fields are bit-exact too)."""
import numpy as np
import pytest
import torch

import oracle
from conftest import rel_err

pytestmark = pytest.mark.gpu

from smokephysai_amd import _lib                                       # noqa: E402
from smokephysai_amd.physics import FractalGenerator, NavierStokesSimulator, SmokeSimulator   # noqa: E402

KEYS = ("u", "v", "p", "density")


def _set(ns, g, tag):
    for k in KEYS:
        setattr(ns, k, torch.from_numpy(g[f"{tag}_{k}"]))


def _eq(ns, g, tag):
    torch.cuda.synchronize()
    for k in KEYS:
        np.testing.assert_array_equal(getattr(ns, k).cpu().numpy(), g[f"{tag}_{k}"], err_msg=f"{tag}_{k}")


@pytest.mark.parametrize("tag,shape", [("s1", (64, 64)), ("s2", (64, 64)), ("r1", (40, 56))])
def test_stages_bit_exact_vs_golden(golden, tag, shape):
    g = golden("physics_stages_64.npz")
    ns = NavierStokesSimulator(shape)
    _set(ns, g, f"{tag}_in")
    ns.run_stage(_lib.STAGE_BUOY_DIFFUSE); _eq(ns, g, f"{tag}_diff")
    np.testing.assert_array_equal(ns.divergence().cpu().numpy(), g[f"{tag}_div"])
    ns.run_stage(_lib.STAGE_PROJECT); _eq(ns, g, f"{tag}_proj")
    ns.run_stage(_lib.STAGE_ADVECT_U); _eq(ns, g, f"{tag}_advu")
    ns.run_stage(_lib.STAGE_ADVECT_V); _eq(ns, g, f"{tag}_advv")
    ns.run_stage(_lib.STAGE_ADVECT_D); _eq(ns, g, f"{tag}_out")
    ns2 = NavierStokesSimulator(shape)
    _set(ns2, g, f"{tag}_in")
    d = ns2.step()
    _eq(ns2, g, f"{tag}_out")
    np.testing.assert_array_equal(d.cpu().numpy(), g[f"{tag}_out_density"])


def test_pure_functions_vs_golden(golden):
    g = golden("physics_stages_64.npz")
    ns = NavierStokesSimulator((64, 64))
    u = torch.from_numpy(g["s2_buoy_u"]).cuda()
    np.testing.assert_array_equal(ns.diffusion_step(u, ns.viscosity).cpu().numpy(), g["s2_diff_u"])
    d = torch.from_numpy(g["s2_buoy_density"]).cuda()
    np.testing.assert_array_equal(ns.diffusion_step(d, ns.viscosity * 0.1).cpu().numpy(), g["s2_diff_density"])
    pu, pv, pd = (torch.from_numpy(g[f"s2_proj_{k}"]).cuda() for k in ("u", "v", "density"))
    au = ns.advection_step(pu, pu, pv)
    np.testing.assert_array_equal(au.cpu().numpy(), g["s2_advu_u"])
    av = ns.advection_step(pv, au, pv)
    np.testing.assert_array_equal(av.cpu().numpy(), g["s2_advv_v"])
    np.testing.assert_array_equal(ns.advection_step(pd, au, av).cpu().numpy(), g["s2_advd_density"])


def test_backtrace_indices_bit_exact(golden):
    g = golden("backtrace_64.npz")
    for st in (1, 2, 50):
        ns = NavierStokesSimulator((64, 64))
        _set(ns, g, f"st{st}_pre")
        for which, nm, stage in ((0, "u", _lib.STAGE_ADVECT_U), (1, "v", _lib.STAGE_ADVECT_V), (2, "d", _lib.STAGE_ADVECT_D)):
            x0, y0 = ns.backtrace_indices(which)
            np.testing.assert_array_equal(x0.cpu().numpy(), g[f"st{st}_{nm}_x0"])
            np.testing.assert_array_equal(y0.cpu().numpy(), g[f"st{st}_{nm}_y0"])
            ns.run_stage(stage)
    ns = NavierStokesSimulator((48, 48))
    ns.u, ns.v, ns.density = (torch.from_numpy(g[k]) for k in ("big_u", "big_v", "big_density"))
    for which, nm in ((0, "u"), (1, "v"), (2, "d")):
        x0, y0 = ns.backtrace_indices(which)
        np.testing.assert_array_equal(x0.cpu().numpy(), g[f"big_{nm}_x0"])
        np.testing.assert_array_equal(y0.cpu().numpy(), g[f"big_{nm}_y0"])
        fld = {"u": ns.u, "v": ns.v, "d": ns.density}[nm]
        np.testing.assert_array_equal(ns.advection_step(fld, ns.u, ns.v).cpu().numpy(), g[f"big_{nm}_out"])


@pytest.mark.parametrize("N,steps", [(64, 50), (128, 100), (256, 200)])
def test_trajectory_bit_exact_vs_golden(golden, N, steps):
    """Config-sized runs from the reference's own initial density: end state bit-identical to the reference."""
    g = golden(f"physics_traj_{N}_2src_{steps}.npz")
    sim = SmokeSimulator((N, N))
    sim.ns_solver.density = torch.from_numpy(g["src_density"])
    frames = sim.simulate_sequence(steps, add_fractal=True)
    torch.cuda.synchronize()
    for k in KEYS:
        np.testing.assert_array_equal(getattr(sim.ns_solver, k).cpu().numpy(), g[f"final_{k}"], err_msg=k)
    assert rel_err(frames[0, -1].cpu().numpy(), g["final_frame_fractal"]) < 1e-6     # sin/cos differ by ulps


def test_config0_one_source_64_50_steps_bit_exact_vs_reference(golden):
    """BASELINE configs[0] -- 64 x 64, one source (32, 32, r 8, 1.0), 50 solver steps -- on the HIP path against the reference's own
    fixture: end state bit-identical, the per-step density sums (fp64 sum of each returned frame) to summation-order accuracy, the same 8
    samples (benchmark.py --num_samples 8) as ONE batch of 8 grids."""
    g = golden("physics_traj_64_1src_50.npz")
    sim = SmokeSimulator((64, 64), batch_size=8)
    sim.ns_solver.density = torch.from_numpy(np.broadcast_to(g["src_density"], (8, 64, 64)).copy())
    frames = sim.simulate_sequence(50, add_fractal=False)
    for b in range(8):
        for k in KEYS:
            np.testing.assert_array_equal(getattr(sim.ns_solver, k)[b].cpu().numpy(), g[f"final_{k}"], err_msg=f"grid {b} {k}")
    sums = frames[3].double().sum(dim=(1, 2)).cpu().numpy()
    np.testing.assert_allclose(sums, g["density_sums"], rtol=1e-12)
    assert abs(sums[-1] - 34.338299) < 1e-4                     # SURVEY 8c sanity value


def test_add_source_and_run_vs_reference(golden):
    g = golden("physics_traj_64_2src_50.npz")
    sim = SmokeSimulator((64, 64))
    sim.add_incense_source([(32, 32), (16, 21)], [1.0, 1.7])
    src = sim.ns_solver.density.cpu().numpy()
    assert np.array_equal(src != 0, g["src_density"] != 0)              # mask: index work, exact
    assert rel_err(src, g["src_density"]) < 1e-6                         # device expf vs SLEEF
    for _ in range(50):
        sim.simulate_step(add_fractal=False)
    for k in KEYS:
        assert rel_err(getattr(sim.ns_solver, k).cpu().numpy(), g[f"final_{k}"]) < 1e-5, k   # bar: 1e-4


def test_batched_equals_oracle_per_grid():
    """B=5 independent grids with different sources, ragged W (not a multiple of 64): each equals the oracle run alone."""
    B, H, W = 5, 48, 48
    rng = np.random.RandomState(3)
    sim = SmokeSimulator((H, W), batch_size=B, jacobi_iters=7)
    orcs = []
    dens = np.zeros((B, H, W), np.float32)
    for b in range(B):
        o = oracle.OracleSmokeSimulator((H, W), jacobi_iters=7)
        for _ in range(rng.randint(1, 4)):
            o.ns_solver.add_smoke_source(rng.randint(10, W - 10), rng.randint(10, H - 10), 8, rng.uniform(0.5, 2.0))
        dens[b] = o.ns_solver.density
        orcs.append(o)
    sim.ns_solver.density = torch.from_numpy(dens)
    frames = sim.simulate_sequence(12, add_fractal=False).cpu().numpy()
    for b, o in enumerate(orcs):
        for t in range(12):
            np.testing.assert_array_equal(frames[b, t], o.simulate_step(add_fractal=False), err_msg=f"grid {b} step {t}")
        for k in KEYS:
            np.testing.assert_array_equal(getattr(sim.ns_solver, k)[b].cpu().numpy(), getattr(o.ns_solver, k))
    # reset of a subset (setup_grid is also the reset, data_loader.py:46)
    sim.ns_solver.setup_grid(grids=[1, 3])
    assert float(sim.ns_solver.density[1].abs().sum()) == 0 and float(sim.ns_solver.u[3].abs().sum()) == 0
    assert float(sim.ns_solver.density[0].abs().sum()) > 0


@pytest.mark.parametrize("H,W,J", [(512, 512, 37), (128, 256, 45), (320, 64, 100), (192, 128, 21), (256, 256, 7), (64, 64, 100), (33, 50, 20), (70, 131, 9)])
def test_band_plans_at_other_shapes_and_sweep_counts_equal_the_oracle(H, W, J):
    """The register-resident Jacobi picks its band plan (rows per wave, unequal band ranges, halo, launch count) from the grid shape and
    splits J sweeps into launches: 3 time steps at shapes / sweep counts beyond the fixtures stay bit-identical to the oracle.  The two odd
    shapes (W not a multiple of 4 / of 64, partial advection tiles) take the one-cell-per-thread diffusion, the per-sweep Jacobi and
    ragged tiles of the fused advection."""
    B = 2
    rng = np.random.RandomState(H + W + J)
    sim = SmokeSimulator((H, W), batch_size=B, jacobi_iters=J)
    orcs = []
    dens = np.zeros((B, H, W), np.float32)
    for b in range(B):
        o = oracle.OracleSmokeSimulator((H, W), jacobi_iters=J)
        for _ in range(3):
            o.ns_solver.add_smoke_source(rng.randint(10, W - 10), rng.randint(10, H - 10), 8, rng.uniform(0.5, 2.0))
        dens[b] = o.ns_solver.density
        orcs.append(o)
    sim.ns_solver.density = torch.from_numpy(dens)
    frames = sim.simulate_sequence(3, add_fractal=False).cpu().numpy()
    for b, o in enumerate(orcs):
        for t in range(3):
            np.testing.assert_array_equal(frames[b, t], o.simulate_step(add_fractal=False), err_msg=f"grid {b} step {t}")
        for k in KEYS:
            np.testing.assert_array_equal(getattr(sim.ns_solver, k)[b].cpu().numpy(), getattr(o.ns_solver, k), err_msg=k)


@pytest.mark.parametrize("H,W,vel,J", [(45, 70, 0.5, 6), (45, 70, 300.0, 6), (32, 64, 150.0, 4), (96, 192, 40.0, 8), (96, 192, 400.0, 8),
                                       (67, 129, 2.0, 5), (40, 128, 90.0, 3), (64, 64, 120.0, 5), (31, 63, 60.0, 3), (33, 65, 60.0, 3), (8, 200, 0.5, 2), (130, 66, 130.0, 2)])
def test_whole_steps_from_dense_random_states_bit_exact_vs_oracle(H, W, vel, J):
    """smk_sim_step from DENSE random states (every cell moves) of two grids, two steps, against the oracle's step().  vel 0.5-40: every
    back-trace of the one-launch advection (k_advect_rows) stays in its 2 x 2 LDS neighbourhood, with non-zero weights everywhere; vel
    90-400: dt * velocity reaches several cells, so whole batches of units leave through far_value2 (the general form on global memory)
    while others of the same launch stay on the fast form, and the clamps at the field's edges bind.  Shapes: odd everything; exactly one
    tile; three x-tiles; a 1-wide last tile with rows that do not fill the last wave; W = two full tiles (the field's extra column of v is
    then the last tile's own unit); a square grid with the fractal multiplier on the emitted frame; one column short of a tile, one column over it (a 1-wide
    second tile whose only lane is also v's extra column), fewer rows than one wave owns, five tile rows with a 2-row last one."""
    B = 2
    sim = SmokeSimulator((H, W), batch_size=B, jacobi_iters=J)
    orcs = []
    for b in range(B):
        rng = np.random.RandomState(1000 * H + W + b)
        o = oracle.OracleSmokeSimulator((H, W), jacobi_iters=J)
        o.ns_solver.u = (rng.standard_normal((H + 1, W)) * vel).astype(np.float32)
        o.ns_solver.v = (rng.standard_normal((H, W + 1)) * vel).astype(np.float32)
        o.ns_solver.p = (rng.standard_normal((H, W)) * 0.1).astype(np.float32)
        o.ns_solver.density = rng.uniform(0.0, 1.8, (H, W)).astype(np.float32)
        orcs.append(o)
    for k in KEYS:
        setattr(sim.ns_solver, k, torch.from_numpy(np.stack([getattr(o.ns_solver, k) for o in orcs])))
    steps, fractal = 2, H == W                                         # (the fractal field is defined for square grids only)
    frames = sim.simulate_sequence(steps, add_fractal=fractal).cpu().numpy()
    for b, o in enumerate(orcs):
        for t in range(steps):
            want = o.simulate_step(add_fractal=fractal)
            if fractal:      # the multiplier's perlin term is a sin / cos sum: <= 2e-6 from the oracle's (test_fractal_constants), not bit-equal
                np.testing.assert_allclose(frames[b, t], want, rtol=1e-6, atol=0, err_msg=f"grid {b} step {t}")
            else:
                np.testing.assert_array_equal(frames[b, t], want, err_msg=f"grid {b} step {t}")
        for k in KEYS:
            np.testing.assert_array_equal(getattr(sim.ns_solver, k)[b].cpu().numpy(), getattr(o.ns_solver, k), err_msg=f"grid {b} {k}")
        assert np.isfinite(o.ns_solver.density).all() and np.abs(o.ns_solver.u).max() > 0


@pytest.mark.parametrize("N", [64, 128, 256])
def test_fractal_constants(golden, N):
    g = golden(f"fractal_{N}.npz")
    fg = FractalGenerator()
    man = fg.generate_mandelbrot_field((N, N)).cpu().numpy()
    np.testing.assert_array_equal(np.round(man * 100).astype(np.uint8), g["mandel_counts"])     # integer work: exact
    np.testing.assert_array_equal(man, g["mandel_counts"].astype(np.float32) / np.float32(100))
    assert np.abs(fg.generate_perlin_noise((N, N)).cpu().numpy() - g["perlin"]).max() < 2e-6
    ones = torch.ones(N, N, device="cuda")
    assert rel_err(fg.apply_fractal_perturbation(ones, 0.05).cpu().numpy(), g["ones_perturbed"]) < 1e-6
    with pytest.raises(IndexError):
        fg.apply_fractal_perturbation(torch.ones(32, 48, device="cuda"))


def test_full_size_properties():
    """BASELINE config 3 size (256^2 x 64, Jacobi-100): size-independent properties instead of a stored answer."""
    B, N = 64, 256
    sim = SmokeSimulator((N, N), batch_size=B, jacobi_iters=100)
    np.random.seed(0)
    srcs = []
    for b in range(B):
        pos, inten = oracle.draw_sources((N, N))
        srcs += [(b, x, y, 8, i) for (x, y), i in zip(pos, inten)]
    sim.ns_solver.add_smoke_sources(srcs)
    # duplicate grid 0's sources into grid 63 -> the two grids must stay bit-identical (independence of grids)
    sim.ns_solver.setup_grid(grids=[63])
    sim.ns_solver.add_smoke_sources([(63, x, y, r, i) for (g, x, y, r, i) in srcs if g == 0])
    ns = sim.ns_solver
    init5 = ns.density[5].cpu().numpy().copy()
    frames = sim.simulate_sequence(5, add_fractal=True)
    assert torch.equal(ns.density[0], ns.density[63]) and torch.equal(ns.p[0], ns.p[63])
    assert torch.isfinite(frames).all()
    # quirk: last row and last column of every advected field are exactly zero
    for f in (ns.u, ns.v, ns.density):
        assert float(f[:, -1, :].abs().max()) == 0.0 and float(f[:, :, -1].abs().max()) == 0.0
    # Jacobi ring is exactly zero
    assert float(ns.p[:, 0].abs().max()) == 0 and float(ns.p[:, :, 0].abs().max()) == 0
    # grid 5 alone on the oracle (5 steps, J=100) from the same initial density: bit-identical
    o = oracle.OracleSmokeSimulator((N, N), jacobi_iters=100)
    o.ns_solver.density = init5
    for _ in range(5):
        o.ns_solver.step()
    np.testing.assert_array_equal(ns.density[5].cpu().numpy(), o.ns_solver.density)
    np.testing.assert_array_equal(ns.p[5].cpu().numpy(), o.ns_solver.p)
    np.testing.assert_array_equal(ns.u[5].cpu().numpy(), o.ns_solver.u)


def test_abi_error_paths_return_status_and_message():
    """Bad arguments come back as negative smk_status + smk_last_error text (no crash, no silent fallback)."""
    import ctypes as C
    L = _lib.load()
    ns = NavierStokesSimulator((32, 32), batch_size=2)
    st = _lib.stream_ptr(ns._dev)
    # unknown stage
    assert L.smk_sim_run_stage(ns._handle, 99, st) == -1 and b"stage" in L.smk_last_error()
    # source on a grid that does not exist
    bad = (_lib.SmkSource * 1)(_lib.SmkSource(5, 3, 3, 2, 1.0))
    assert L.smk_sim_add_sources(ns._handle, bad, 1, st) == -1 and b"grid" in L.smk_last_error()
    # pitch smaller than the row
    t = torch.zeros(2, 33, 32, device="cuda")
    desc = _lib.SmkSimDesc(2, 32, 32, 20, 0.01, 0.001, ns._dev.index, 16, 33, t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr())
    h = C.c_void_p()
    assert L.smk_sim_create(C.byref(desc), C.byref(h)) == -1 and b"pitch" in L.smk_last_error()
    # fractal emit on a non-square grid: the reference raises there too
    rect = NavierStokesSimulator((32, 48))
    frames = torch.empty(1, 32, 48, device="cuda")
    with pytest.raises(_lib.SmokeHipError, match="square"):
        rect.step_into(frames, 1, add_fractal=True)
    rect.step_into(frames, 1, add_fractal=False)             # fine without the fractal
    with pytest.raises(ValueError):
        ns.step_into(torch.empty(2, 32, 32, device="cuda", dtype=torch.float64))
    with pytest.raises(ValueError):
        ns.advection_step(torch.zeros(5, 5, device="cuda"), ns.u, ns.v)


def test_sources_apply_in_list_order_per_grid():
    """fp32 += is order-sensitive: two overlapping sources on one grid must equal the oracle's sequential adds."""
    o = oracle.OracleNS((64, 64))
    o.add_smoke_source(30, 30, 8, 1.3)
    o.add_smoke_source(33, 31, 8, 0.7)
    o.add_smoke_source(30, 30, 8, 0.9)
    ns = NavierStokesSimulator((64, 64), batch_size=3)
    ns.add_smoke_sources([(1, 30, 30, 8, 1.3), (0, 10, 10, 8, 1.0), (1, 33, 31, 8, 0.7), (1, 30, 30, 8, 0.9)])
    got = ns.density[1].cpu().numpy()
    assert rel_err(got, o.density) < 1e-6
    assert float(ns.density[2].abs().sum()) == 0.0 and float(ns.density[0].abs().sum()) > 0.0


@pytest.mark.parametrize("name", ["cell", "u", "v"])
def test_public_interpolation_helpers_vs_reference(golden, name):
    """NavierStokesSimulator.bilinear_interpolate / interpolate_velocity_u / _v (navier_stokes.py:97-131) as
    stand-alone methods, on the reference's own outputs for seeded fields and coordinates on integers, on the exact upper edge
    (returns 0: the quirk), negative and far outside the field: bit-exact."""
    g = golden("interp_64.npz")
    f, y, x = (torch.from_numpy(g[f"{name}_{k}"]).cuda() for k in ("field", "y", "x"))
    ns = NavierStokesSimulator((48, 64))
    np.testing.assert_array_equal(ns.bilinear_interpolate(f, y, x).cpu().numpy(), g[f"{name}_bilinear"])
    np.testing.assert_array_equal(ns.interpolate_velocity_u(f, y, x).cpu().numpy(), g[f"{name}_interp_u"])
    np.testing.assert_array_equal(ns.interpolate_velocity_v(f, y, x).cpu().numpy(), g[f"{name}_interp_v"])
    # upper-edge quirk, stated directly: f[.., w-1] is NOT returned at x == w-1
    R, C = f.shape
    yy = torch.full((4,), 3.0, device="cuda"); xx = torch.tensor([C - 1.0, C - 2.0, C - 1.5, 0.0], device="cuda")
    got = ns.bilinear_interpolate(f, yy, xx).cpu().numpy()
    assert got[0] == 0.0 and got[1] == g[f"{name}_field"][3, C - 2] and got[3] == g[f"{name}_field"][3, 0]
    # batched fields: shared 2-D coordinates, and one coordinate list per field
    fb = torch.stack([f, 2 * f, -f])
    sh = ns.bilinear_interpolate(fb, y, x)
    assert sh.shape == (3,) + tuple(y.shape) and torch.equal(sh[1], ns.bilinear_interpolate(2 * f, y, x))
    yb = torch.stack([y, y + 0.25, y - 1.0]); xb = torch.stack([x, x - 0.5, x + 2.0])
    per = ns.interpolate_velocity_u(fb, yb, xb)
    for b in range(3):
        assert torch.equal(per[b], ns.interpolate_velocity_u(fb[b], yb[b], xb[b]))
    # the composition the reference's advection_step spells out (navier_stokes.py:74-95) equals the fused advect kernel
    h, w = 48, 64
    rng = np.random.RandomState(5)
    u = torch.from_numpy(rng.randn(h + 1, w).astype(np.float32) * 30).cuda()
    v = torch.from_numpy(rng.randn(h, w + 1).astype(np.float32) * 30).cuda()
    d = torch.from_numpy(rng.rand(h, w).astype(np.float32)).cuda()
    Y, X = torch.meshgrid(torch.arange(h, device="cuda", dtype=torch.float32), torch.arange(w, device="cuda", dtype=torch.float32), indexing="ij")
    ui, vi = ns.interpolate_velocity_u(u, Y, X), ns.interpolate_velocity_v(v, Y, X)
    px = torch.clamp(X - ns.dt * ui, 0, w - 1); py = torch.clamp(Y - ns.dt * vi, 0, h - 1)
    assert torch.equal(ns.bilinear_interpolate(d, py, px), ns.advection_step(d, u, v))


@pytest.mark.parametrize("H,W,J,B,steps", [(256, 256, 100, 64, 3), (256, 256, 20, 32, 4), (192, 128, 40, 64, 3), (320, 64, 100, 64, 2),
                                            (128, 128, 20, 128, 3), (128, 256, 8, 64, 3), (512, 256, 24, 32, 2)])
def test_chip_filling_batches_equal_the_oracle(H, W, J, B, steps):
    """Batches that fill the chip (the band plan depends on the batch through the workgroup count): 1 / 2 / 4 cells per lane, one to
    several bands per grid, several launches per projection and a handful of sweeps -- the first, a middle and the last grid stay
    bit-identical to the oracle."""
    rng = np.random.RandomState(H * 7 + W + J)
    sim = SmokeSimulator((H, W), batch_size=B, jacobi_iters=J)
    picks = sorted({0, B // 2 + 1, B - 1})
    srcs, per = [], {b: [] for b in picks}
    for b in range(B):
        for _ in range(rng.randint(1, 4)):
            s = (b, int(rng.randint(10, W - 10)), int(rng.randint(10, H - 10)), 8, float(rng.uniform(0.5, 2.0)))
            srcs.append(s)
            if b in per:
                per[b].append(s)
    sim.ns_solver.add_smoke_sources(srcs)
    init = {b: sim.ns_solver.density[b].cpu().numpy().copy() for b in picks}
    for _ in range(steps):
        sim.ns_solver.step()
    torch.cuda.synchronize()
    for b in picks:
        o = oracle.OracleNS((H, W), jacobi_iters=J)
        o.density = init[b].copy()
        for _ in range(steps):
            o.step()
        for k in KEYS:
            np.testing.assert_array_equal(getattr(sim.ns_solver, k)[b].cpu().numpy(), getattr(o, k), err_msg=f"grid {b} {k}")
