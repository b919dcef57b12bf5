#!/usr/bin/env python3
"""Headline benchmark: simulated + encoded frames/s (BASELINE.json metric) on N GPUs of one node.

One "step" = one time step of the batched Navier-Stokes stepper for all B grids of this rank (buoyancy, diffusion,
Jacobi pressure projection, three advections, fractal frame emit) + the fused CNN encoder over the B emitted frames
([B,256,256] -> [B,128,32,32]), everything resident in HBM.  Workload at N=1: BASELINE.json configs[2]
(256^2 grid, batch 64, Jacobi-100).  Grids are independent: ranks own disjoint grids, no data-path collective
("scaling": "weak", batch 64 per GPU).

    python bench.py [--gpus N --steps K --warmup W]          (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.  `roofline*` use SURVEY.md 8(d)'s algorithmic figures: stencil 4*(27+3J) bytes per
cell per step against HBM 8 TB/s; encoder 153,728*N^2 flop per frame against the dense MFMA peak of the dtype.
`cpu_baseline` times the CPU oracle (a scalar C port of the reference path) on a bounded sample on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0, "i8x3": 5000.0}   # dense peaks (MI355X_MICROARCH.md)


def encoder_weights(seed=0):
    """Random-init input_encoder of the reference architecture (smokephys_net.py:24-32) with non-trivial BN stats."""
    torch.manual_seed(seed)
    enc = torch.nn.Sequential(torch.nn.Conv2d(1, 64, 7, padding=3), torch.nn.BatchNorm2d(64), torch.nn.ReLU(),
                              torch.nn.Conv2d(64, 128, 3, padding=1), torch.nn.BatchNorm2d(128), torch.nn.ReLU())
    g = torch.Generator().manual_seed(seed + 7)
    with torch.no_grad():
        for bn in (enc[1], enc[4]):
            bn.running_mean.copy_(torch.randn(bn.num_features, generator=g) * 0.2)
            bn.running_var.copy_(torch.rand(bn.num_features, generator=g) * 1.5 + 0.25)
            bn.weight.copy_(torch.rand(bn.num_features, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(bn.num_features, generator=g) * 0.1)
    from smokephysai_amd.models.encoder import encoder_weight_dict
    return {k: v.detach().clone() for k, v in encoder_weight_dict(enc).items()}


def draw_sources(B, N, seed):
    """Per grid the draw order of data_loader.py:49-58 (1-3 sources, x,y in [20,N-20), intensity U(0.5,2))."""
    rng = np.random.RandomState(seed)
    out = []
    for b in range(B):
        for _ in range(rng.randint(1, 4)):
            x = rng.randint(20, N - 20)
            y = rng.randint(20, N - 20)
            out.append((b, int(x), int(y), 8, float(rng.uniform(0.5, 2.0))))
    return out


def cpu_baseline(N, J, weights, budget_frames):
    """The CPU oracle (oracle/: C port of the reference path) on one grid, looped per frame like the reference:
    step() + fractal recomputed per frame (fractal_generator.py:55-56) + input_encoder + pools.  The stepper is
    scalar; the encoder is the row-vectorised OpenMP fp32 variant on up to 16 host threads (the GPU box's CPU share)."""
    import oracle
    threads = max(1, min(16, os.cpu_count() or 1))
    w = {k: v.cpu().numpy() for k, v in weights.items()}

    def run(nthreads, frames):
        oracle.set_threads(nthreads)
        sim = oracle.OracleSmokeSimulator((N, N), jacobi_iters=J, cache_fractal=False)
        sim.ns_solver.add_smoke_source(N // 2, N // 2, 8, 1.0)
        frame = sim.simulate_step(add_fractal=True)
        oracle.encoder_features_fast(frame[None], w, input_dim=128)          # warm-up (thread pool, page faults)
        t0 = time.perf_counter()
        t_sim = 0.0
        for _ in range(frames):
            ts = time.perf_counter()
            frame = sim.simulate_step(add_fractal=True)
            t_sim += time.perf_counter() - ts
            oracle.encoder_features_fast(frame[None], w, input_dim=128)
        return time.perf_counter() - t0, t_sim

    dt, t_sim = run(threads, budget_frames)
    n1 = max(2, budget_frames // 10)
    dt1, _ = run(1, n1)                                        # also with one thread
    return {"value": budget_frames / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{budget_frames} frames of one {N}x{N} grid (Jacobi-{J}, fractal recomputed per frame): scalar C stepper "
                      f"({t_sim / budget_frames * 1e3:.0f} ms/frame) + OpenMP fp32 C encoder on {threads} threads, {dt:.1f} s total",
            "single_thread": {"value": n1 / dt1, "unit": "frames/s", "cores": 1, "sample": f"{n1} frames, {dt1:.1f} s"},
            "host_cores_available": os.cpu_count(),
            "reference_on_8_cores_in_build_container": "6.45 frames/s (BASELINE.md: actual reference code, torch CPU/oneDNN)"}


def hbm_copy_gbs(dev):
    """Measured device-copy bandwidth (read + write bytes / time) of a 1 GiB fp32 tensor: the practical HBM ceiling
    beside the 8 TB/s spec peak used for roofline.frac (MI355X_MICROARCH.md quotes ~6.3 TB/s for a float4 copy)."""
    a = torch.empty(256 * 1024 * 1024, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(10):
        b.copy_(a)
    torch.cuda.synchronize(dev)
    return 2 * a.numel() * 4 * 10 / (time.perf_counter() - t0) / 1e9


def inference_ms(dev, N, frames, encoder_dtype):
    """Metric M2 (BASELINE.json): SmokePhysNet.forward wall time per frame, eval mode, full 27.8 M-parameter network
    (HIP encoder + transformer/heads, fp32-accurate), device-synchronised, at the reference's batch sizes (benchmark.py
    uses 4, inference.py uses 1) and at the simulation batch.  The headline figures replay the forward from a captured
    hipGraph (GraphedSmokePhysNet; bit-identical to the eager call); the eager figures are reported beside them."""
    from smokephysai_amd.models import SmokePhysNet, GraphedSmokePhysNet
    torch.manual_seed(0)
    model = SmokePhysNet(encoder_dtype=encoder_dtype).to(dev).eval()
    graphed = GraphedSmokePhysNet(model)
    res, eager = {}, {}
    with torch.no_grad():
        for bs in (1, 4, frames.shape[0]):
            x = frames[:bs, None].contiguous()
            for name, fwd, out in (("eager", model, eager), ("graph", graphed, res)):
                for _ in range(2):
                    fwd(x)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                reps = 10 if bs <= 4 else 5
                for _ in range(reps):
                    fwd(x)
                torch.cuda.synchronize(dev)
                out[f"batch{bs}"] = (time.perf_counter() - t0) / reps / bs * 1e3
    res["eager"] = eager
    # ~68.7 GFLOP per 256^2 frame (61.1 at 128^2) for the whole forward, reference formulation
    gflop = {256: 68.7, 128: 61.1}.get(N)
    if gflop:
        res["counted_TFLOPs_at_sim_batch"] = gflop / res[f"batch{frames.shape[0]}"]
    res["body"] = ("libsmokehip split-bf16 kernels: fused encoder, token linears (bias/pos-embed/chaos-term/GELU/residual epilogues), "
                   "flash attention, chaos addend, LayerNorm, conv reconstruction head")
    res["note"] = f"{N}x{N} frames; hipGraph replay (eager launch beside it); reference README: 610.92 ms/frame (hardware unstated)"
    return res


def train_step_ms(dev, N, B, steps=3):
    """train.py's step (zero_grad, batch_losses, backward, clip 1.0, AdamW) on a device-built batch; ms per step."""
    import train
    from smokephysai_amd.models import SmokePhysNet
    from smokephysai_amd.models.physics_regularizer import PhysicsRegularizer
    torch.manual_seed(0)
    model = SmokePhysNet().to(dev).train()
    reg = PhysicsRegularizer()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    seq = torch.rand(B, 20, N, N, device=dev)
    batch = {"input": seq[:, 9:10].contiguous(), "target": seq[:, 10:11].contiguous(),
             "chaos_features": torch.rand(B, 3, device=dev), "sequence": seq}

    def step():
        opt.zero_grad()
        total = train.batch_losses(model, reg, batch, dev)[0]
        total.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        opt.step()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return {"ms_per_step": (time.perf_counter() - t0) / steps * 1e3, "batch": B, "grid": N,
            "note": "forward + backward + clip + AdamW; linear GEMMs, attention, LayerNorm and the encoder's BatchNorm/ReLU/pool on libsmokehip, "
                    "convolutions / GELU / dropout on PyTorch-ROCm"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64, help="grids per GPU")
    ap.add_argument("--jacobi", type=int, default=100)
    ap.add_argument("--encoder-dtype", default="bf16x3", choices=["f32", "bf16x3", "bf16", "i8x3"],
                    help="bf16x3 (headline; BASELINE configs[2] says \"encoder bf16\"): split-bf16 MFMA, fp32 accumulate, "
                         "features within 3e-6 (max-norm) of the reference; i8x3: 16-bit fixed point on int8 MFMA, within "
                         "3.5e-5 and ~1.6x faster. The other one of these two is timed too and reported under alt.")
    ap.add_argument("--cpu-frames", type=int, default=240, help="frames in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-encode", action="store_true", help="stencil only (diagnostic; not the headline metric)")
    ap.add_argument("--no-alt", action="store_true", help="time only --encoder-dtype (profiling runs)")
    ap.add_argument("--no-inference", action="store_true", help="skip the per-frame inference-ms measurement (metric M2)")
    ap.add_argument("--train-step", action="store_true",
                    help="also time train.py's optimisation step (BASELINE configs[3]'s per-GPU shape: --batch frames of --grid^2, full model; "
                         "adds about a minute: MIOpen tunes its convolutions on first use)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    # one process per GPU; SMK_BENCH_BACKEND=gloo lets a 1-GPU box rehearse the N>1 control path (ranks share cuda:0)
    backend = os.environ.get("SMK_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)       # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    from smokephysai_amd.models.encoder import HipEncoder
    from smokephysai_amd.physics import SmokeSimulator

    B, N, J, K, W = args.batch, args.grid, args.jacobi, args.steps, args.warmup
    sim = SmokeSimulator((N, N), device=dev, batch_size=B, jacobi_iters=J)
    sim.ns_solver.add_smoke_sources(draw_sources(B, N, seed=rank))     # each rank owns its own B grids
    weights = encoder_weights(0)
    enc = HipEncoder(weights, device=dev)
    frame = torch.empty(B, N, N, device=dev)

    def make_step(dtype):
        def step(ev=None):
            if ev: ev[0].record()
            sim.ns_solver.step_into(frame, 1, add_fractal=True, fractal_intensity=0.05)
            if ev: ev[1].record()
            if args.no_encode:
                feats = None
            elif dtype == "f32":
                feats = enc(frame, input_dim=128, dtype="f32")                        # [B,128,32,32]
            else:
                feats = enc.tokens(frame, input_dim=128, dtype=dtype)                 # same features, token-major [B,1024,128]
            if ev: ev[2].record()
            return feats
        return step

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed(dtype):
        """W warm-up steps, then exactly K timed steps between barrier + synchronize; max over ranks."""
        step = make_step(dtype)
        for _ in range(W):
            step()
        events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            feats = step(events[k])
        torch.cuda.synchronize(); barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert torch.isfinite(frame).all() and (feats is None or torch.isfinite(feats).all())
        ms_sim = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))      # HIP events on the launch stream
        ms_enc = float(np.mean([e[1].elapsed_time(e[2]) for e in events]))
        return elapsed, ms_sim, ms_enc

    elapsed, ms_sim, ms_enc = timed(args.encoder_dtype)
    alt = None
    if world == 1 and not args.no_encode and not args.no_alt and args.encoder_dtype in ("i8x3", "bf16x3"):
        other = "bf16x3" if args.encoder_dtype == "i8x3" else "i8x3"
        a_el, a_sim, a_enc = timed(other)
        alt = {"encoder_dtype": other, "value": B * K / a_el, "unit": "frames/s", "ms_per_step": a_el / K * 1e3,
               "ms_encode_per_step": a_enc, "counted_TFLOPs": B * 153728.0 * N * N / (a_enc * 1e-3) / 1e12}

    if rank == 0:
        frames_total = world * B * K
        stencil_bytes = B * N * N * 4.0 * (27 + 3 * J)          # algorithmic bytes per stencil pass
        enc_flops = B * 153728.0 * N * N                         # algorithmic flop per encoder launch
        sten_gbs = stencil_bytes / (ms_sim * 1e-3) / 1e9
        roof_stencil = {"bound": "hbm", "kernel": "stencil pass (all kernels of one time step, B grids)",
                        "achieved": sten_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sten_gbs / HBM_PEAK_GBS,
                        "traffic": None, "ms_per_launch": ms_sim}
        out = {"metric": "simulated+encoded frames/sec at 256^2 grid, batch 64", "value": frames_total / elapsed,
               "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": {"f32": "f32", "bf16x3": "f32 stencil + bf16x3 encoder (split-bf16 MFMA, fp32 accumulate)",
                         "bf16": "f32 stencil + bf16 encoder",
                         "i8x3": "f32 stencil + i8x3 encoder (16-bit fixed point on int8 MFMA, i32 accumulate)"}[args.encoder_dtype],
               "data": "synthetic",
               "config": {"workload": f"configs[2]: {N}x{N} grid, batch {B} per GPU, Jacobi-{J} project, fractal frame emit, "
                                      f"CNN encoder {args.encoder_dtype} -> [B,128,32,32]",
                          "grid": N, "batch_per_gpu": B, "jacobi_iters": J, "encoder_dtype": args.encoder_dtype,
                          "parallelism": f"independent grids sharded over {world} GPU(s), no data-path collective"},
               "sim_only_frames_per_s": B / (ms_sim * 1e-3), "ms_sim_per_step": ms_sim}
        if not args.no_encode:
            tf = enc_flops / (ms_enc * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.encoder_dtype]
            kname = {"bf16x3": "k_encoder_b16 (split-bf16 on v_mfma_f32_16x16x32_bf16)", "bf16": "k_encoder_bf16<false>",
                     "i8x3": "k_encoder_i8", "f32": "k_encoder_f32"}[args.encoder_dtype]
            roof_enc = {"bound": "mfma", "kernel": f"{kname}: fused conv1+conv2+pool, B frames",
                        "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
                        "ms_per_launch": ms_enc,
                        "note": "achieved counts SURVEY 8(d)'s algorithmic flops; the x3 modes execute 3 MFMA products per "
                                "counted multiply (split operands), so matrix-pipe work is 3x the counted figure"}
            out.update({"encode_only_frames_per_s": B / (ms_enc * 1e-3), "ms_encode_per_step": ms_enc,
                        "roofline": roof_enc if ms_enc >= ms_sim else roof_stencil,
                        "roofline_stencil": roof_stencil, "roofline_encoder": roof_enc})
        else:
            out["metric"] += " (DIAGNOSTIC: stencil only, not the headline metric)"
            out["roofline"] = roof_stencil
        if alt is not None:
            out["alt"] = alt
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):                                  # filled by tools/pmc_traffic.py from rocprofv3 --pmc passes
            tr = json.load(open(pmc))
            for key, kname in (("roofline_stencil", "stencil"), ("roofline_encoder", "encoder")):
                if key in out and kname in tr:
                    out[key]["traffic"] = tr[kname]
        sq = os.path.join(ROOT, "profiles", "pmc_mfma.json")     # tools/pmc_encoder.sh: matrix-pipe busy cycles of the headline kernel
        if os.path.exists(sq) and "roofline_encoder" in out and args.encoder_dtype == "bf16x3":
            out["roofline_encoder"]["pmc"] = json.load(open(sq))
        if world == 1:
            out["hbm_copy_measured_GBs"] = hbm_copy_gbs(dev)
        if world == 1 and not args.no_encode and not args.no_inference:
            out["inference_ms_per_frame"] = inference_ms(dev, N, frame, args.encoder_dtype)
        if world == 1 and args.train_step:
            out["train_step"] = train_step_ms(dev, N, B)
        if world == 1 and args.cpu_frames > 0:
            out["cpu_baseline"] = cpu_baseline(N, J, weights, args.cpu_frames)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
