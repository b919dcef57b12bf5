#!/usr/bin/env python3
"""Golden-vector generator (container-only tool).

Imports the *reference* implementation from /root/reference on CPU and records
inputs + expected outputs of the hot path as small .npz fixtures next to this
script.  The reference never travels to the GPU box; only these data files do.

Run (here, never on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden.py

What is recorded:
  physics_stages_64.npz   per-stage states inside step 1 and 2 (64x64, 1 source)
  physics_traj_*.npz      final states of 50/100/200-step runs (1- and 2-source scenes)
  backtrace_64.npz        int64 x0/y0 back-trace indices of the 3 advects, steps 1/2/50
  fractal_{64,128,256}.npz  linspace coords, perlin fp32, mandelbrot escape counts (uint8)
  dataset_seed0_*.npz     np.random.seed(0) SyntheticSmokeDataset samples
  encoder_*.npz           input_encoder weights (+ BN running stats), frames, features
  model_small.npz         small-config SmokePhysNet weights + explicit chaos noise + outputs
  train_batch.npz         one seeded batch -> the four train.py loss scalars + grad norm
  interp_64.npz           the reference's own bilinear_interpolate / interpolate_velocity_u / _v on seeded fields and coordinates
                          (interior, exact upper edge = the zero quirk, negative, far outside)
  ref_cache_64.pkl        the pickle cache the REFERENCE's SyntheticSmokeDataset writes (data_loader.py:25-35), 2 samples at 64^2
  transformer_layer.npz   one ChaosTransformerLayer (dim 128, 2 heads of 64, L=128): weights, input, the three noise draws,
                          the ChaosAttention output and the layer output
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn.functional as F

from src.physics.navier_stokes import NavierStokesSimulator
from src.physics.smoke_simulator import SmokeSimulator
from src.physics.fractal_generator import FractalGenerator
from src.models.smokephys_net import SmokePhysNet, ChaosTransformerLayer
from src.models.physics_regularizer import PhysicsRegularizer
from src.utils.data_loader import SyntheticSmokeDataset

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


def state(ns):
    return dict(u=ns.u.numpy().copy(), v=ns.v.numpy().copy(),
                p=ns.p.numpy().copy(), density=ns.density.numpy().copy())


# ---------------------------------------------------------------- physics stages
def staged_step(ns, rec, tag):
    """Run ns.step() stage by stage, calling the reference's own methods in
    the reference's own order (navier_stokes.py:151-173), recording after each."""
    def snap(stage):
        for k, a in state(ns).items():
            rec[f"{tag}_{stage}_{k}"] = a
    snap("in")
    buoyancy = ns.density * 0.1
    ns.v[:, :-1] += ns.dt * buoyancy
    snap("buoy")
    ns.u = ns.diffusion_step(ns.u, ns.viscosity)
    ns.v = ns.diffusion_step(ns.v, ns.viscosity)
    ns.density = ns.diffusion_step(ns.density, ns.viscosity * 0.1)
    snap("diff")
    div = (ns.u[1:, :] - ns.u[:-1, :] + ns.v[:, 1:] - ns.v[:, :-1]) / ns.dt
    rec[f"{tag}_div"] = div.numpy().copy()
    ns.pressure_projection()
    snap("proj")
    ns.u = ns.advection_step(ns.u, ns.u, ns.v)
    snap("advu")
    ns.v = ns.advection_step(ns.v, ns.u, ns.v)
    snap("advv")
    ns.density = ns.advection_step(ns.density, ns.u, ns.v)
    snap("advd")
    ns.density *= 0.995
    snap("out")


def gen_stages():
    rec = {}
    ns = NavierStokesSimulator((64, 64), device="cpu")
    ns.add_smoke_source(32, 32, radius=8, intensity=1.0)
    rec["source_density"] = ns.density.numpy().copy()
    staged_step(ns, rec, "s1")
    staged_step(ns, rec, "s2")
    # cross-check: staged == monolithic step()
    ns2 = NavierStokesSimulator((64, 64), device="cpu")
    ns2.add_smoke_source(32, 32, radius=8, intensity=1.0)
    ns2.step(); ns2.step()
    for k, a in state(ns2).items():
        assert np.array_equal(a, rec[f"s2_out_{k}"]), k
    # a non-square, odd-sized grid with a hand-made velocity field (exercises
    # clamps, the zero-at-upper-edge quirk and larger displacements)
    g = torch.Generator().manual_seed(1234)
    ns3 = NavierStokesSimulator((40, 56), device="cpu")
    ns3.u = (torch.rand(ns3.u.shape, generator=g) - 0.5) * 300.0
    ns3.v = (torch.rand(ns3.v.shape, generator=g) - 0.5) * 300.0
    ns3.p = (torch.rand(ns3.p.shape, generator=g) - 0.5)
    ns3.density = torch.rand(ns3.density.shape, generator=g)
    staged_step(ns3, rec, "r1")
    save("physics_stages_64.npz", **rec)


def backtrace_indices(ns, field, u, v):
    """x0/y0 (int64) of the final bilinear gather of advection_step
    (navier_stokes.py:74-95,115-123), via the reference's own helpers."""
    h, w = field.shape
    y = torch.arange(h, dtype=torch.float32)
    x = torch.arange(w, dtype=torch.float32)
    Y, X = torch.meshgrid(y, x, indexing="ij")
    ui = ns.interpolate_velocity_u(u, Y, X)
    vi = ns.interpolate_velocity_v(v, Y, X)
    px = torch.clamp(X - ns.dt * ui, 0, w - 1)
    py = torch.clamp(Y - ns.dt * vi, 0, h - 1)
    x0 = torch.clamp(torch.floor(px).long(), 0, w - 1)
    y0 = torch.clamp(torch.floor(py).long(), 0, h - 1)
    return x0.numpy(), y0.numpy(), px.numpy(), py.numpy()


def gen_backtrace():
    rec = {}
    ns = NavierStokesSimulator((64, 64), device="cpu")
    ns.add_smoke_source(32, 32, radius=8, intensity=1.0)
    for step in range(1, 51):
        want = step in (1, 2, 50)
        # replicate step() up to the advects
        buoyancy = ns.density * 0.1
        ns.v[:, :-1] += ns.dt * buoyancy
        ns.u = ns.diffusion_step(ns.u, ns.viscosity)
        ns.v = ns.diffusion_step(ns.v, ns.viscosity)
        ns.density = ns.diffusion_step(ns.density, ns.viscosity * 0.1)
        ns.pressure_projection()
        if want:
            for k, a in state(ns).items():
                rec[f"st{step}_pre_{k}"] = a
            x0, y0, px, py = backtrace_indices(ns, ns.u, ns.u, ns.v)
            rec[f"st{step}_u_x0"], rec[f"st{step}_u_y0"] = x0, y0
        ns.u = ns.advection_step(ns.u, ns.u, ns.v)
        if want:
            x0, y0, px, py = backtrace_indices(ns, ns.v, ns.u, ns.v)
            rec[f"st{step}_v_x0"], rec[f"st{step}_v_y0"] = x0, y0
        ns.v = ns.advection_step(ns.v, ns.u, ns.v)
        if want:
            x0, y0, px, py = backtrace_indices(ns, ns.density, ns.u, ns.v)
            rec[f"st{step}_d_x0"], rec[f"st{step}_d_y0"] = x0, y0
        ns.density = ns.advection_step(ns.density, ns.u, ns.v)
        ns.density *= 0.995
    for k, a in state(ns).items():
        rec[f"final50_{k}"] = a
    # strong-velocity case so that indices differ from the identity map
    g = torch.Generator().manual_seed(99)
    ns3 = NavierStokesSimulator((48, 48), device="cpu")
    ns3.u = (torch.rand(ns3.u.shape, generator=g) - 0.5) * 800.0
    ns3.v = (torch.rand(ns3.v.shape, generator=g) - 0.5) * 800.0
    ns3.density = torch.rand(ns3.density.shape, generator=g)
    rec["big_u"], rec["big_v"], rec["big_density"] = ns3.u.numpy().copy(), ns3.v.numpy().copy(), ns3.density.numpy().copy()
    for nm, fld in (("u", ns3.u), ("v", ns3.v), ("d", ns3.density)):
        x0, y0, px, py = backtrace_indices(ns3, fld, ns3.u, ns3.v)
        rec[f"big_{nm}_x0"], rec[f"big_{nm}_y0"] = x0, y0
        rec[f"big_{nm}_out"] = ns3.advection_step(fld, ns3.u, ns3.v).numpy()
    save("backtrace_64.npz", **rec)


# ---------------------------------------------------------------- trajectories
def gen_traj():
    # single source, 64^2, 50 steps (config C1) with per-step density sums
    ns = NavierStokesSimulator((64, 64), device="cpu")
    ns.add_smoke_source(32, 32, radius=8, intensity=1.0)
    rec = {"src_density": ns.density.numpy().copy()}
    sums = []
    for _ in range(50):
        d = ns.step()
        sums.append(float(d.double().sum()))
    rec.update({f"final_{k}": a for k, a in state(ns).items()})
    rec["density_sums"] = np.array(sums)
    save("physics_traj_64_1src_50.npz", **rec)

    for N, steps in ((64, 50), (128, 100), (256, 200)):
        sim = SmokeSimulator((N, N), device="cpu")
        sim.add_incense_source([(N // 2, N // 2), (N // 4, N // 3)], [1.0, 1.7])
        rec = {"src_density": sim.ns_solver.density.numpy().copy()}
        sums, fsums = [], []
        frame = None
        for t in range(steps):
            # fractal only on first/last frames: it is a shape-only constant and slow
            frame = sim.simulate_step(add_fractal=(t == steps - 1))
            sums.append(float(sim.ns_solver.density.double().sum()))
        rec.update({f"final_{k}": a for k, a in state(sim.ns_solver).items()})
        rec["final_frame_fractal"] = frame.numpy().copy()
        rec["density_sums"] = np.array(sums)
        rec["meta"] = np.array([N, steps])
        save(f"physics_traj_{N}_2src_{steps}.npz", **rec)


# ---------------------------------------------------------------- fractal
def gen_fractal():
    fg = FractalGenerator(device="cpu")
    for N in (64, 128, 256):
        perlin = fg.generate_perlin_noise((N, N))
        mand = fg.generate_mandelbrot_field((N, N))
        counts = torch.round(mand * 100).to(torch.uint8)
        assert torch.equal(counts.float() / 100, mand)
        ones = torch.ones(N, N)
        mult = fg.apply_fractal_perturbation(ones, intensity=0.05)  # 1 + 0.05*F
        save(f"fractal_{N}.npz",
             lin_perlin=torch.linspace(0, 10.0, N).numpy(),
             lin_mx=torch.linspace(-2.5, 1.5, N).numpy(),
             lin_my=torch.linspace(-1.5, 1.5, N).numpy(),
             perlin=perlin.numpy(), mandel_counts=counts.numpy(),
             fractal_field=(0.7 * perlin + 0.3 * mand).numpy(),
             ones_perturbed=mult.numpy())
    # odd sizes pin the linspace restatement
    lin = {}
    for n in (2, 3, 7, 8, 9, 15, 16, 17, 31, 33, 37, 100, 255, 257, 512):
        lin[f"p_{n}"] = torch.linspace(0, 10.0, n).numpy()
        lin[f"mx_{n}"] = torch.linspace(-2.5, 1.5, n).numpy()
        lin[f"my_{n}"] = torch.linspace(-1.5, 1.5, n).numpy()
    save("linspace_probe.npz", **lin)


# ---------------------------------------------------------------- dataset
def gen_dataset():
    for N, nsamp in ((64, 2), (128, 1)):
        np.random.seed(0)
        torch.manual_seed(0)
        ds = SyntheticSmokeDataset(num_samples=nsamp, grid_size=(N, N), device="cpu")
        rec = {}
        for i, s in enumerate(ds.data):
            seq = s["sequence"].numpy()
            pos = np.array(s["source_config"]["positions"], dtype=np.int64)
            inten = np.array(s["source_config"]["intensities"], dtype=np.float64)
            rec[f"s{i}_positions"] = pos
            rec[f"s{i}_intensities"] = inten
            if N == 64:
                rec[f"s{i}_sequence"] = seq
            else:
                rec[f"s{i}_sequence_sel"] = seq[[0, 5, 10, 19]]
            rec[f"s{i}_frame_sums"] = seq.astype(np.float64).sum(axis=(1, 2))
            cf = s["chaos_features"]
            rec[f"s{i}_chaos"] = np.array([cf["lyapunov_exponent"], cf["fractal_dimension"], cf["entropy"]], dtype=np.float64)
        # __getitem__ draw order after generation (data_loader.py:108)
        np.random.seed(123)
        item = ds[0]
        rec["item0_seed123_input"] = item["input"].numpy()
        rec["item0_seed123_target"] = item["target"].numpy()
        rec["item0_seed123_chaos"] = item["chaos_features"].numpy()
        save(f"dataset_seed0_{N}.npz", **rec)


def gen_chaos_stats():
    """get_chaos_features() pieces on a 64^2 25-frame history."""
    sim = SmokeSimulator((64, 64), device="cpu")
    sim.add_incense_source([(32, 32), (20, 40)], [1.0, 1.5])
    frames = []
    for t in range(25):
        frames.append(sim.simulate_step().numpy().copy())
    feats = sim.get_chaos_features()
    cur = sim.history[-1]
    binary = (cur > cur.mean()).float()
    counts = []
    for scale in (2, 4, 8, 16, 32):
        bh, bw = 64 // scale, 64 // scale
        c = 0
        for i in range(bh):
            for j in range(bw):
                if binary[i*scale:(i+1)*scale, j*scale:(j+1)*scale].sum() > 0:
                    c += 1
        counts.append(c)
    hist = torch.histogram(cur.flatten(), bins=256, range=(0, 1)).hist
    states = torch.stack(sim.history[-20:])
    dists = np.array([torch.norm(states[i+1] - states[i]).item() for i in range(19)])
    save("chaos_stats_64.npz", frames=np.stack(frames),
         feats=np.array([feats["lyapunov_exponent"], feats["fractal_dimension"], feats["entropy"]], dtype=np.float64),
         box_counts=np.array(counts, dtype=np.int64), hist_counts=hist.numpy().astype(np.int64),
         lyap_dists=dists, mean=np.array(float(cur.mean())))


# ---------------------------------------------------------------- encoder / model
def randomize_bn(model, gen):
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=gen) * 1.5 + 0.25)
            m.weight.data.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=gen) * 0.1)


def sim_frames(N, count, stride):
    sim = SmokeSimulator((N, N), device="cpu")
    sim.add_incense_source([(N // 2, N // 2), (N // 4, N // 3)], [1.0, 1.7])
    out = []
    for t in range(count * stride):
        f = sim.simulate_step(add_fractal=False)
        if (t + 1) % stride == 0:
            out.append(f.numpy().copy())
    return np.stack(out)


def gen_encoder():
    torch.manual_seed(0)
    model = SmokePhysNet(input_dim=128)  # input_dim fixes AdaptiveAvgPool target (smokephys_net.py:31)
    gen = torch.Generator().manual_seed(7)
    with torch.no_grad():
        randomize_bn(model.input_encoder, gen)
    model.eval()
    enc = model.input_encoder
    w = {
        "conv1_w": enc[0].weight.detach().numpy(), "conv1_b": enc[0].bias.detach().numpy(),
        "bn1_w": enc[1].weight.detach().numpy(), "bn1_b": enc[1].bias.detach().numpy(),
        "bn1_mean": enc[1].running_mean.numpy(), "bn1_var": enc[1].running_var.numpy(),
        "conv2_w": enc[3].weight.detach().numpy(), "conv2_b": enc[3].bias.detach().numpy(),
        "bn2_w": enc[4].weight.detach().numpy(), "bn2_b": enc[4].bias.detach().numpy(),
        "bn2_mean": enc[4].running_mean.numpy(), "bn2_var": enc[4].running_var.numpy(),
    }
    save("encoder_weights.npz", **w)
    for N, nf in ((64, 2), (128, 2), (256, 1)):
        frames = sim_frames(N, nf, 10)
        # plus one dense frame (sim frames are mostly zero background)
        dense = torch.rand(1, N, N, generator=torch.Generator().manual_seed(100 + N)) * 1.8
        frames = np.concatenate([frames, dense.numpy()])
        x = torch.from_numpy(frames)[:, None]
        with torch.no_grad():
            enc_out = enc(x)
            feats = F.adaptive_avg_pool2d(enc_out, (32, 32))   # smokephys_net.py:87-91
            c1 = enc[2](enc[1](enc[0](x)))
        rec = dict(frames=frames, features=feats.numpy())
        if N == 64:
            rec["conv1_act"] = c1.numpy()[:1]       # [1,64,64,64] = 1 MiB raw
        save(f"encoder_io_{N}.npz", **rec)


def chaos_noise(seed, layers, B):
    """The 3 randn(B,1) draws per layer, in ChaosAttention.generate_chaos_field order
    (chaos_attention.py:50-52), for a CPU generator seeded with `seed`."""
    torch.manual_seed(seed)
    return torch.stack([torch.stack([torch.randn(B, 1) for _ in range(3)]) for _ in range(layers)])


def gen_model_small():
    torch.manual_seed(0)
    model = SmokePhysNet(input_dim=32, hidden_dim=64, num_layers=2, num_heads=4, output_channels=16)
    gen = torch.Generator().manual_seed(11)
    with torch.no_grad():
        randomize_bn(model, gen)
    model.eval()
    frames = sim_frames(64, 2, 10)
    x = torch.from_numpy(frames)[:, None]
    noise = chaos_noise(5, 2, 2)
    torch.manual_seed(5)
    with torch.no_grad():
        out = model(x, return_features=True)
    rec = {f"w::{k}": v.numpy() for k, v in model.state_dict().items()}
    rec.update(frames=frames, chaos_noise=noise.numpy(),
               reconstructed=out["reconstructed"].numpy(), physics_features=out["physics_features"].numpy(),
               latent_features=out["latent_features"].numpy(), intermediate_features=out["intermediate_features"].numpy())
    save("model_small.npz", **rec)

    # one training batch through train.py:66-91's loss decomposition (small model, train mode, dropout 0)
    torch.manual_seed(0)
    model = SmokePhysNet(input_dim=32, hidden_dim=64, num_layers=2, num_heads=4, output_channels=16)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    sd0 = {f"w::{k}": v.numpy().copy() for k, v in model.state_dict().items()}
    B = 2
    seq = torch.from_numpy(np.stack([sim_frames(128, 20, 1), sim_frames(128, 20, 1) * 0.5]))  # [B,20,128,128]
    inputs, targets = seq[:, 7:8], seq[:, 8:9]
    chaos_targets = torch.tensor([[0.01, 1.2, 3.0], [0.0, 1.0, 2.5]])
    noise = chaos_noise(9, 2, B)
    torch.manual_seed(9)
    reg = PhysicsRegularizer()
    outputs = model(inputs)
    recon = F.mse_loss(outputs["reconstructed"], targets)
    chaos = F.mse_loss(outputs["physics_features"], chaos_targets)
    pl = reg({"density": outputs["reconstructed"], "density_sequence": seq}, {"density": targets})
    phys = pl["total_physics_loss"]
    total = recon + 0.1 * chaos + 0.05 * phys
    total.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
    sd0.update(sequence_sel=seq[:, [0, 7, 8, 19]].numpy(), seq_mean_abs_dt=np.array(float(pl["continuity"])),
               inputs=inputs.numpy(), targets=targets.numpy(), chaos_targets=chaos_targets.numpy(),
               chaos_noise=noise.numpy(),
               losses=np.array([float(total), float(recon), float(phys), float(chaos)], dtype=np.float64),
               mass=np.array(float(pl["mass_conservation"])), grad_norm=np.array(float(gnorm)))
    save("train_batch.npz", **sd0)


def gen_model_full_checksums():
    torch.manual_seed(0)
    model = SmokePhysNet().eval()
    sd = model.state_dict()
    cs = {k: np.array([float(v.double().sum()), float(v.double().abs().sum())]) for k, v in sd.items()}
    frames = sim_frames(128, 1, 10)
    x = torch.from_numpy(frames)[:, None]
    noise = chaos_noise(3, 6, 1)
    torch.manual_seed(3)
    with torch.no_grad():
        out = model(x)
    save("model_full_checksums.npz", **{f"cs::{k}": v for k, v in cs.items()},
         frames=frames, chaos_noise=noise.numpy(), physics_features=out["physics_features"].numpy(),
         latent_features=out["latent_features"].numpy(),
         recon_sel=out["reconstructed"].numpy()[0, 0, ::8, ::8],
         nparams=np.array(sum(p.numel() for p in model.parameters())))


def gen_transformer_layer():
    """One reference ChaosTransformerLayer in eval mode (smokephys_net.py:136-168, chaos_attention.py:68-114): pins the
    build's body kernels (linear, chaos addend, attention, LayerNorm) to the reference's own arithmetic."""
    torch.manual_seed(21)
    layer = ChaosTransformerLayer(128, 2, chaos_strength=0.1).eval()
    gen = torch.Generator().manual_seed(22)
    B, L = 2, 128
    with torch.no_grad():
        for prm in layer.parameters():          # non-trivial LayerNorm affine, larger logits than the default init gives
            if prm.dim() == 1:
                prm.add_(torch.randn(prm.shape, generator=gen) * 0.1)
        layer.chaos_attention.q_proj.weight.mul_(3.0)
        layer.chaos_attention.k_proj.weight.mul_(3.0)
        x = torch.randn(B, L, 128, generator=gen)
        noise = chaos_noise(23, 1, B)[0]                      # [3,B,1]
        torch.manual_seed(23)
        attn = layer.chaos_attention(layer.norm1(x))
        torch.manual_seed(23)
        out = layer(x)
    rec = {f"w::{k}": v.numpy() for k, v in layer.state_dict().items()}
    rec.update(x=x.numpy(), noise=noise.numpy(), attention_out=attn.numpy(), layer_out=out.numpy())
    save("transformer_layer.npz", **rec)


# ---------------------------------------------------------------- interpolation helpers (navier_stokes.py:97-131)
def gen_interp():
    """The three public interpolation methods, called directly."""
    rng = np.random.RandomState(11)
    h, w = 48, 64
    ns = NavierStokesSimulator((h, w), device="cpu")
    rec = {}
    fields = {"cell": (h, w), "u": (h + 1, w), "v": (h, w + 1)}
    for name, (R, C) in fields.items():
        f = rng.randn(R, C).astype(np.float32)
        n = 4096
        # interior points, points exactly on integer coordinates, the exact upper edge (zero quirk), and out-of-range values
        y = rng.uniform(-3.0, R + 2.0, n).astype(np.float32)
        x = rng.uniform(-3.0, C + 2.0, n).astype(np.float32)
        y[:256] = rng.randint(0, R, 256).astype(np.float32)
        x[128:384] = rng.randint(0, C, 256).astype(np.float32)
        y[400:420] = R - 1
        x[410:440] = C - 1
        y[440:450] = np.float32(R - 1) - np.float32(1e-4)
        x[450:460] = np.float32(C - 1) - np.float32(1e-4)
        y[460:470] = [-0.0, -0.5, -1.0, -1e-7, 1e-7, 0.5, R - 0.5, R, R + 0.5, 1e6]
        x[470:480] = [-0.0, -0.5, -1.0, -1e-7, 1e-7, 0.5, C - 0.5, C, C + 0.5, -1e6]
        y2 = y.reshape(64, 64); x2 = x.reshape(64, 64)
        ft, yt, xt = torch.from_numpy(f), torch.from_numpy(y2), torch.from_numpy(x2)
        rec[f"{name}_field"], rec[f"{name}_y"], rec[f"{name}_x"] = f, y2, x2
        rec[f"{name}_bilinear"] = ns.bilinear_interpolate(ft, yt, xt).numpy()
        rec[f"{name}_interp_u"] = ns.interpolate_velocity_u(ft, yt, xt).numpy()
        rec[f"{name}_interp_v"] = ns.interpolate_velocity_v(ft, yt, xt).numpy()
    save("interp_64.npz", **rec)


# ---------------------------------------------------------------- the reference's on-disk dataset cache (data_loader.py:25-35)
def gen_ref_cache():
    """The pickle the reference writes; same seed as dataset_seed0_64.npz (so both fixtures describe the same samples)."""
    import tempfile, shutil
    d = tempfile.mkdtemp()
    try:
        np.random.seed(0)
        torch.manual_seed(0)
        SyntheticSmokeDataset(num_samples=2, grid_size=(64, 64), device="cpu", cache_path=os.path.join(d, "train_data.pkl"))
        shutil.copyfile(os.path.join(d, "train_data.pkl"), os.path.join(OUT, "ref_cache_64.pkl"))
        print(f"wrote ref_cache_64.pkl: {os.path.getsize(os.path.join(OUT, 'ref_cache_64.pkl'))/1024:.1f} KiB")
    finally:
        shutil.rmtree(d)


if __name__ == "__main__":
    which = sys.argv[1:] or ["stages", "backtrace", "traj", "fractal", "dataset", "chaos", "encoder", "model", "full", "layer", "interp", "cache"]
    fns = dict(interp=gen_interp, cache=gen_ref_cache, stages=gen_stages, backtrace=gen_backtrace, traj=gen_traj, fractal=gen_fractal,
               dataset=gen_dataset, chaos=gen_chaos_stats, encoder=gen_encoder, model=gen_model_small,
               full=gen_model_full_checksums, layer=gen_transformer_layer)
    with torch.no_grad():
        for w in which:
            if w in ("model",):
                continue
            fns[w]()
    if "model" in which:
        gen_model_small()
