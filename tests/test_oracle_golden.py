"""Pins the CPU oracle (oracle/smoke_oracle.c) against golden vectors captured from the reference
(tests/golden/generate_golden.py).  CPU only.  Stepper stages are compared BIT-EXACT: the oracle restates
the reference's fp32 operation order, and torch's CPU elementwise kernels round once per op.
Only transcendental-function call sites (exp in add_smoke_source, sin/cos in perlin) get a tolerance,
because torch uses SLEEF and the oracle uses libm (<= 2 ulp apart)."""
import numpy as np
import pytest

import oracle
from conftest import rel_err

STAGES = ["in", "buoy", "diff", "proj", "advu", "advv", "advd", "out"]


def _load_state(ns, g, tag):
    ns.u, ns.v = g[f"{tag}_u"].copy(), g[f"{tag}_v"].copy()
    ns.p, ns.density = g[f"{tag}_p"].copy(), g[f"{tag}_density"].copy()


def _assert_state(ns, g, tag):
    for k in ("u", "v", "p", "density"):
        np.testing.assert_array_equal(getattr(ns, k), g[f"{tag}_{k}"], err_msg=f"{tag}_{k}")


@pytest.mark.parametrize("tag,shape", [("s1", (64, 64)), ("s2", (64, 64)), ("r1", (40, 56))])
def test_stepper_stages_bit_exact(golden, tag, shape):
    g = golden("physics_stages_64.npz")
    ns = oracle.OracleNS(shape)
    _load_state(ns, g, f"{tag}_in")
    ns.buoyancy()
    _assert_state(ns, g, f"{tag}_buoy")
    ns.u = ns.diffusion_step(ns.u, ns.viscosity)
    ns.v = ns.diffusion_step(ns.v, ns.viscosity)
    ns.density = ns.diffusion_step(ns.density, ns.viscosity * 0.1)
    _assert_state(ns, g, f"{tag}_diff")
    np.testing.assert_array_equal(ns.divergence(), g[f"{tag}_div"])
    ns.pressure_projection()
    _assert_state(ns, g, f"{tag}_proj")
    ns.u = ns.advection_step(ns.u, ns.u, ns.v)
    _assert_state(ns, g, f"{tag}_advu")
    ns.v = ns.advection_step(ns.v, ns.u, ns.v)
    _assert_state(ns, g, f"{tag}_advv")
    ns.density = ns.advection_step(ns.density, ns.u, ns.v)
    _assert_state(ns, g, f"{tag}_advd")
    # fused step() from the same input
    ns2 = oracle.OracleNS(shape)
    _load_state(ns2, g, f"{tag}_in")
    ns2.step()
    _assert_state(ns2, g, f"{tag}_out")


def test_upper_edge_quirk(golden):
    """bilinear returns 0 at the upper clamp edge -> last row/col of every advected field is 0."""
    g = golden("physics_stages_64.npz")
    for k in ("u", "v", "density"):
        a = g[f"s1_out_{k}"]
        assert not a[-1, :].any() and not a[:, -1].any()
    f = np.arange(12, dtype=np.float32).reshape(3, 4) + 1
    out = oracle.bilinear_interpolate(f, np.array([1.0, 2.0, 0.5], np.float32), np.array([3.0, 1.0, 1.5], np.float32))
    assert out[0] == 0.0 and out[1] == 0.0 and out[2] == np.float32(0.25 * (2 + 3 + 6 + 7))


def test_add_source(golden):
    g = golden("physics_stages_64.npz")
    ns = oracle.OracleNS((64, 64))
    ns.add_smoke_source(32, 32, radius=8, intensity=1.0)
    ref = g["source_density"]
    assert np.array_equal(ns.density != 0, ref != 0)            # mask is index work: exact
    assert int((ref != 0).sum()) <= 201
    assert rel_err(ns.density, ref) < 5e-7                      # expf vs SLEEF


def test_backtrace_indices_bit_exact(golden):
    g = golden("backtrace_64.npz")
    for st in (1, 2, 50):
        ns = oracle.OracleNS((64, 64))
        _load_state(ns, g, f"st{st}_pre")
        out, x0, y0 = ns.advection_step(ns.u, ns.u, ns.v, want_indices=True)
        np.testing.assert_array_equal(x0, g[f"st{st}_u_x0"]); np.testing.assert_array_equal(y0, g[f"st{st}_u_y0"])
        ns.u = out
        out, x0, y0 = ns.advection_step(ns.v, ns.u, ns.v, want_indices=True)
        np.testing.assert_array_equal(x0, g[f"st{st}_v_x0"]); np.testing.assert_array_equal(y0, g[f"st{st}_v_y0"])
        ns.v = out
        out, x0, y0 = ns.advection_step(ns.density, ns.u, ns.v, want_indices=True)
        np.testing.assert_array_equal(x0, g[f"st{st}_d_x0"]); np.testing.assert_array_equal(y0, g[f"st{st}_d_y0"])
    # strong velocities: indices far from the identity map
    ns = oracle.OracleNS((48, 48))
    ns.u, ns.v, ns.density = g["big_u"].copy(), g["big_v"].copy(), g["big_density"].copy()
    for nm, fld in (("u", ns.u), ("v", ns.v), ("d", ns.density)):
        out, x0, y0 = ns.advection_step(fld, ns.u, ns.v, want_indices=True)
        np.testing.assert_array_equal(x0, g[f"big_{nm}_x0"]); np.testing.assert_array_equal(y0, g[f"big_{nm}_y0"])
        np.testing.assert_array_equal(out, g[f"big_{nm}_out"])
        ident = np.broadcast_to(np.arange(fld.shape[1]), fld.shape)
        assert (x0 != ident).mean() > 0.5


def test_trajectory_from_golden_source_bit_exact(golden):
    """50/100 steps from the reference's own initial density: bit-exact end state."""
    g = golden("physics_traj_64_1src_50.npz")
    ns = oracle.OracleNS((64, 64))
    ns.density = g["src_density"].copy()
    sums = []
    for _ in range(50):
        sums.append(float(ns.step().astype(np.float64).sum()))
    for k in ("u", "v", "p", "density"):
        np.testing.assert_array_equal(getattr(ns, k), g[f"final_{k}"])
    np.testing.assert_allclose(sums, g["density_sums"], rtol=1e-12)   # fp64 sum order only
    assert abs(sums[-1] - 34.338299) < 1e-4                     # SURVEY 8c sanity value


@pytest.mark.parametrize("N,steps", [(64, 50), (128, 100), (256, 200)])
def test_two_source_trajectories(golden, N, steps):
    g = golden(f"physics_traj_{N}_2src_{steps}.npz")
    # (a) from the reference's initial density: bit-exact
    ns = oracle.OracleNS((N, N))
    ns.density = g["src_density"].copy()
    for _ in range(steps):
        ns.step()
    for k in ("u", "v", "p", "density"):
        np.testing.assert_array_equal(getattr(ns, k), g[f"final_{k}"])
    frame = oracle.apply_fractal_perturbation(ns.density, 0.05)
    assert rel_err(frame, g["final_frame_fractal"]) < 1e-6
    # (b) from the oracle's own add_source (libm expf): within 1e-6 of the reference
    sim = oracle.OracleSmokeSimulator((N, N))
    sim.add_incense_source([(N // 2, N // 2), (N // 4, N // 3)], [1.0, 1.7])
    for _ in range(steps):
        sim.ns_solver.step()
    for k in ("u", "v", "p", "density"):
        assert rel_err(getattr(sim.ns_solver, k), g[f"final_{k}"]) < 2e-6, k


@pytest.mark.parametrize("N", [64, 128, 256])
def test_fractal(golden, N):
    g = golden(f"fractal_{N}.npz")
    np.testing.assert_array_equal(oracle.linspace(0.0, 10.0, N), g["lin_perlin"])
    np.testing.assert_array_equal(oracle.linspace(-2.5, 1.5, N), g["lin_mx"])
    np.testing.assert_array_equal(oracle.linspace(-1.5, 1.5, N), g["lin_my"])
    np.testing.assert_array_equal(oracle.mandelbrot_counts(N, N), g["mandel_counts"])   # integer work: exact
    assert np.abs(oracle.perlin(N, N) - g["perlin"]).max() < 1e-6
    assert np.abs(oracle.fractal_field(N, N) - g["fractal_field"]).max() < 1e-6
    ones = np.ones((N, N), np.float32)
    assert rel_err(oracle.apply_fractal_perturbation(ones, 0.05), g["ones_perturbed"]) < 1e-6


def test_linspace_probe(golden):
    g = golden("linspace_probe.npz")
    rng = dict(p=(0.0, 10.0), mx=(-2.5, 1.5), my=(-1.5, 1.5))
    for k, ref in g.items():
        nm, n = k.split("_")
        np.testing.assert_array_equal(oracle.linspace(*rng[nm], int(n)), ref, err_msg=k)


@pytest.mark.parametrize("N", [64, 128])
def test_dataset_frame_stream(golden, N):
    """np.random.seed(0) -> identical source lists and frame sequence."""
    g = golden(f"dataset_seed0_{N}.npz")
    np.random.seed(0)
    nsamp = 2 if N == 64 else 1
    sim = oracle.OracleSmokeSimulator((N, N), cache_fractal=True)
    for i in range(nsamp):
        sim.ns_solver.setup_grid()
        pos, inten = oracle.draw_sources((N, N))
        np.testing.assert_array_equal(np.array(pos, dtype=np.int64), g[f"s{i}_positions"])
        np.testing.assert_array_equal(np.array(inten), g[f"s{i}_intensities"])
        sim.add_incense_source(pos, inten)
        seq, chaos = [], []
        for t in range(20):
            seq.append(sim.simulate_step())
            if t >= 10:
                f = sim.get_chaos_features()
                if f:
                    chaos.append(f)
        seq = np.stack(seq)
        if N == 64:
            assert rel_err(seq, g[f"s{i}_sequence"]) < 2e-6
        else:
            assert rel_err(seq[[0, 5, 10, 19]], g[f"s{i}_sequence_sel"]) < 2e-6
        np.testing.assert_allclose(seq.astype(np.float64).sum(axis=(1, 2)), g[f"s{i}_frame_sums"], rtol=2e-6)
        avg = [np.mean([c[k] for c in chaos]) for k in ("lyapunov_exponent", "fractal_dimension", "entropy")]
        np.testing.assert_allclose(avg, g[f"s{i}_chaos"], rtol=1e-4, atol=1e-6)
    if N == 128:
        assert g["s0_positions"].tolist() == [[67, 84]]          # SURVEY 8c probe values
        assert abs(g["s0_intensities"][0] - 1.4041450641) < 1e-9


def test_chaos_stats_integer_exact(golden):
    g = golden("chaos_stats_64.npz")
    sim = oracle.OracleSmokeSimulator((64, 64))
    sim.history = [f for f in g["frames"]]
    np.testing.assert_array_equal(sim.box_counts(), g["box_counts"])
    np.testing.assert_array_equal(sim.hist_counts(), g["hist_counts"])
    f = sim.get_chaos_features()
    np.testing.assert_allclose([f["lyapunov_exponent"], f["fractal_dimension"], f["entropy"]], g["feats"],
                               rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("N", [64, 128])
def test_encoder_features(golden, N):
    """CNN features within 1e-4 rel (BASELINE.json tolerance) -- the oracle accumulates in fp64, the
    reference in fp32 (oneDNN), so they agree to ~1e-6."""
    w = golden("encoder_weights.npz")
    g = golden(f"encoder_io_{N}.npz")
    frames = g["frames"]
    if N == 64:
        feats, c1 = oracle.encoder_features(frames[:1], w, want_conv1=True)
        assert rel_err(c1, g["conv1_act"]) < 1e-5
        assert rel_err(feats, g["features"][:1]) < 1e-5
        feats = oracle.encoder_features(frames[-1:], w)
        assert rel_err(feats, g["features"][-1:]) < 1e-5
    else:
        feats = oracle.encoder_features(frames[-1:], w)         # the dense frame
        assert rel_err(feats, g["features"][-1:]) < 1e-5
        fast = oracle.encoder_features_fast(frames[-1:], w)     # the timing-grade port bench.py uses as CPU baseline
        assert rel_err(fast, g["features"][-1:]) < 1e-5


def test_transformer_layer_oracle_matches_reference(golden):
    """oracle.chaos_attention / chaos_transformer_layer (numpy, the reference's own formulation with the separate chaos-score
    term) against the reference ChaosTransformerLayer's captured outputs (dim 128, 2 heads of 64, L = 128)."""
    g = golden("transformer_layer.npz")
    w = {k[3:]: v for k, v in g.items() if k.startswith("w::")}
    attn = oracle.chaos_attention(oracle.layernorm(g["x"], w["norm1.weight"], w["norm1.bias"]), w, g["noise"], 2,
                                  prefix="chaos_attention.")
    assert rel_err(attn, g["attention_out"]) < 2e-6
    out = oracle.chaos_transformer_layer(g["x"], w, g["noise"], 2)
    assert rel_err(out, g["layer_out"]) < 2e-6
    # the Q-fold identity the HIP path relies on: (Q + strength * gate * C_h) K^T == scores + strength * gate * chaos_scores
    states = oracle.lorenz_states(g["noise"])
    assert states.shape == (2, 5, 3) and states.dtype == np.float32


@pytest.mark.parametrize("name", ["cell", "u", "v"])
def test_interpolation_helpers_vs_reference(golden, name):
    """The reference's public bilinear_interpolate / interpolate_velocity_u / _v (navier_stokes.py:97-131) called directly on
    seeded fields with coordinates on integers, on the exact upper edge (the zero quirk), negative and far outside: bit-exact."""
    g = golden("interp_64.npz")
    f, y, x = g[f"{name}_field"], g[f"{name}_y"], g[f"{name}_x"]
    np.testing.assert_array_equal(oracle.bilinear_interpolate(f, y, x), g[f"{name}_bilinear"])
    np.testing.assert_array_equal(oracle.interpolate_velocity_u(f, y, x), g[f"{name}_interp_u"])
    np.testing.assert_array_equal(oracle.interpolate_velocity_v(f, y, x), g[f"{name}_interp_v"])
    R, C = f.shape
    edge = (y == R - 1) | (x == C - 1)                       # the quirk: exactly 0 on the upper clamp edge
    assert edge.sum() >= 30 and np.all(g[f"{name}_bilinear"][edge & (y <= R - 1) & (x <= C - 1)] == 0)


# ---- the N-D restatement (oracle/ns_nd.py = SPEC_3D.md's executable form): its 2-D instance against the reference's own fixtures --------
def _nd(shape, **kw):
    from oracle.ns_nd import OracleNSnd
    return OracleNSnd(shape, **kw)


@pytest.mark.parametrize("tag,shape", [("s1", (64, 64)), ("s2", (64, 64)), ("r1", (40, 56))])
def test_nd_oracle_2d_instance_stages_bit_exact_vs_reference(golden, tag, shape):
    """Every stage of a step (buoyancy, 3 diffusions, divergence, projection, 3 advections, the fused step) of the generic N-D code with
    a 2-tuple grid == the reference's captured states, bit for bit: the 3-D oracle is this code with one more axis."""
    g = golden("physics_stages_64.npz")
    ns = _nd(shape)
    _load_state(ns, g, f"{tag}_in")
    ns.buoyancy()
    _assert_state(ns, g, f"{tag}_buoy")
    ns.diffuse_all()
    _assert_state(ns, g, f"{tag}_diff")
    np.testing.assert_array_equal(ns.divergence(), g[f"{tag}_div"])
    ns.pressure_projection()
    _assert_state(ns, g, f"{tag}_proj")
    ns.u = ns.advection_step(ns.u, [ns.u, ns.v])
    _assert_state(ns, g, f"{tag}_advu")
    ns.v = ns.advection_step(ns.v, [ns.u, ns.v])
    _assert_state(ns, g, f"{tag}_advv")
    ns.density = ns.advection_step(ns.density, [ns.u, ns.v])
    _assert_state(ns, g, f"{tag}_advd")
    ns2 = _nd(shape)
    _load_state(ns2, g, f"{tag}_in")
    ns2.step()
    _assert_state(ns2, g, f"{tag}_out")


@pytest.mark.parametrize("name,N,steps", [("physics_traj_64_1src_50.npz", 64, 50), ("physics_traj_64_2src_50.npz", 64, 50),
                                          ("physics_traj_128_2src_100.npz", 128, 100)])
def test_nd_oracle_2d_instance_trajectories_bit_exact_vs_reference(golden, name, N, steps):
    g = golden(name)
    ns = _nd((N, N))
    ns.density = g["src_density"].copy()
    for _ in range(steps):
        ns.step()
    for k in ("u", "v", "p", "density"):
        np.testing.assert_array_equal(getattr(ns, k), g[f"final_{k}"], err_msg=k)


def test_nd_oracle_2d_instance_backtrace_indices_and_sources(golden):
    g = golden("backtrace_64.npz")
    ns = _nd((48, 48))
    ns.u, ns.v, ns.density = g["big_u"].copy(), g["big_v"].copy(), g["big_density"].copy()
    for nm, fld in (("u", ns.u), ("v", ns.v), ("d", ns.density)):
        out, (y0, x0) = ns.advection_step(fld, [ns.u, ns.v], want_indices=True)
        np.testing.assert_array_equal(x0, g[f"big_{nm}_x0"]); np.testing.assert_array_equal(y0, g[f"big_{nm}_y0"])
        np.testing.assert_array_equal(out, g[f"big_{nm}_out"])
    s = golden("physics_stages_64.npz")
    ns = _nd((64, 64))
    ns.add_smoke_source(32, 32, radius=8, intensity=1.0)
    assert np.array_equal(ns.density != 0, s["source_density"] != 0) and rel_err(ns.density, s["source_density"]) < 5e-7
    # and the C oracle == the N-D code on a Jacobi-100 run (the C restatement is what the 2-D GPU tests use at J = 100)
    a, b = oracle.OracleNS((64, 64), jacobi_iters=100), _nd((64, 64), jacobi_iters=100)
    a.density = s["source_density"].copy(); b.density = s["source_density"].copy()
    for _ in range(5):
        a.step(); b.step()
    for k in ("u", "v", "p", "density"):
        np.testing.assert_array_equal(getattr(a, k), getattr(b, k), err_msg=k)


def test_nd_oracle_3d_instance_properties():
    """The 3-D instance: shapes, the zero-at-the-upper-edge quirk on every axis, the Jacobi ring, warm-started p, finite values, and
    the u / v / w coupling (buoyancy drives v; the projection spreads it to u and w)."""
    ns = _nd((12, 20, 16))
    assert ns.u.shape == (12, 21, 16) and ns.v.shape == (12, 20, 17) and ns.w.shape == (13, 20, 16)
    ns.add_smoke_source(8, 10, 6, radius=4, intensity=1.5)                 # (x, y, z)
    assert ns.density[6, 10, 8] == np.float32(1.5) and ns.density[6, 10, 13] == 0
    p_prev = None
    for _ in range(4):
        fr = ns.step()
        assert np.isfinite(fr).all()
        for f in (ns.u, ns.v, ns.w, ns.density):
            # the 2-D consequence of the upper-edge quirk carries over to y and x: u (v) is not staggered along the axis it acts on, so its
            # shifted sample lands exactly on the clamp edge in the last column (row), the velocity there is 0, the back-trace stays on the
            # edge and the gather returns 0.  Along z it does not: w IS staggered along its own axis (SPEC_3D.md section 5)
            assert not f[:, -1].any() and not f[:, :, -1].any()
        assert not ns.p[0].any() and not ns.p[-1].any() and not ns.p[:, 0].any() and not ns.p[:, :, -1].any()
        p_prev = ns.p.copy()
    assert np.abs(ns.u).max() > 0 and np.abs(ns.v).max() > 0 and np.abs(ns.w).max() > 0
    assert np.abs(p_prev).max() > 0 and ns.density[-1].any()
    # the interpolation routine itself: a coordinate exactly on the upper edge of ANY axis gives 0 (weights from the clamped indices)
    f = (np.arange(4 * 5 * 6, dtype=np.float32) + 1).reshape(4, 5, 6)
    one = lambda z, y, x: float(ns.interpolate(f, [np.array([c], np.float32) for c in (z, y, x)])[0])
    assert one(3.0, 1.5, 2.5) == 0.0 and one(1.5, 4.0, 2.5) == 0.0 and one(1.5, 2.5, 5.0) == 0.0
    assert one(1.5, 2.5, 3.5) == float(np.float32(f[1:3, 2:4, 3:5].mean()))


def test_encoder3d_oracle_matches_torch_cpu_operators():
    """oracle/encoder3d.py (numpy, fp64: SPEC_3D.md section 8) against the same operator chain evaluated by torch's CPU kernels in fp64 --
    F.conv3d / F.batch_norm / F.adaptive_avg_pool3d, the third-party arithmetic the reference's own (2-D) encoder runs on."""
    import torch
    import torch.nn.functional as F
    from oracle.encoder3d import encoder3d_features
    rng = np.random.RandomState(3)
    w = dict(conv1_w=rng.randn(64, 1, 7, 7, 7) * 0.05, conv1_b=rng.randn(64) * 0.1, bn1_w=rng.rand(64) + 0.5, bn1_b=rng.randn(64) * 0.1,
             bn1_mean=rng.randn(64) * 0.2, bn1_var=rng.rand(64) + 0.3, conv2_w=rng.randn(128, 64, 3, 3, 3) * 0.03,
             conv2_b=rng.randn(128) * 0.1, bn2_w=rng.rand(128) + 0.5, bn2_b=rng.randn(128) * 0.1,
             bn2_mean=rng.randn(128) * 0.2, bn2_var=rng.rand(128) + 0.3)
    vol = rng.rand(5, 32, 64)
    feats, a1 = encoder3d_features(vol, w)
    t = {k: torch.from_numpy(np.asarray(v, np.float64)) for k, v in w.items()}
    x = torch.from_numpy(vol)[None, None]
    y = F.relu(F.batch_norm(F.conv3d(x, t["conv1_w"], t["conv1_b"], padding=3), t["bn1_mean"], t["bn1_var"], t["bn1_w"], t["bn1_b"], False, 0.1, 1e-5))
    assert rel_err(a1, y[0].numpy()) < 1e-12
    y = F.relu(F.batch_norm(F.conv3d(y, t["conv2_w"], t["conv2_b"], padding=1), t["bn2_mean"], t["bn2_var"], t["bn2_w"], t["bn2_b"], False, 0.1, 1e-5))
    y = F.adaptive_avg_pool3d(F.adaptive_avg_pool3d(y, (1, 128, 128)), (1, 32, 32))
    assert feats.shape == (128, 32, 32) and rel_err(feats, y[0, :, 0].numpy()) < 1e-12
