#!/usr/bin/env python3
"""Headline benchmark: simulated + encoded frames/s (BASELINE.json metric) on N GPUs of one node.

One "step" = one time step of the batched Navier-Stokes stepper for all B grids of this rank (buoyancy, diffusion,
Jacobi pressure projection, three advections, fractal frame emit) + the fused CNN encoder over the B emitted frames
([B,256,256] -> [B,128,32,32]), everything resident in HBM.  Workload at N=1: BASELINE.json configs[2]
(256^2 grid, batch 64, Jacobi-100).  Grids are independent: ranks own disjoint grids, no data-path collective
("scaling": "weak", batch 64 per GPU).

    python bench.py [--gpus N --steps K --warmup W]

N>1 runs one process per GPU over RCCL.  Either the caller starts the ranks (python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...: RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* come from the environment) or plain
`python bench.py --gpus N` starts them itself: with WORLD_SIZE unset the parent spawns torch.distributed.run as a child
process BEFORE anything touches the GPU, relays the children's output (rank 0's single JSON line) and exits with their
code (`--dry-run` prints the launch command instead).  The data path has no collective; the `train_step` block is the
one place a collective runs: train.py's optimisation step under DistributedDataParallel (configs[3]'s per-GPU shape,
27.8 M fp32 gradients = 111 MB all-reduced per step over RCCL), timed with the same barrier + max-over-ranks rule.

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


def usable_host_cores():
    """Cores this process may actually use: its affinity mask, cut by a cgroup CPU quota if one is set (a GPU box may show every core
    of the host while the job owns a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, quota) if quota else n), quota


def cpu_baseline(N, J, weights, budget_frames):
    """The CPU oracle (oracle/: C port of the reference path) on one grid, looped per frame like the reference (no batching advantage):
    step() + fractal recomputed per frame (fractal_generator.py:55-56) + input_encoder + pools.  SURVEY 8(d): on ALL host cores this
    process may use, and on one thread, core count printed.  The stepper is scalar (the reference's elementwise torch ops on 65,536
    cells do not thread either); the encoder is oracle/encoder_fast.c -- register-blocked fp32 micro-kernel, AVX-512 / AVX2+FMA picked at
    load time, OpenMP over (channel block, row) tasks."""
    import oracle
    threads, quota = usable_host_cores()
    w = {k: v.cpu().numpy() for k, v in weights.items()}

    def run(nthreads, frames, budget_s):
        oracle.set_threads(nthreads)
        sim = oracle.OracleSmokeSimulator((N, N), jacobi_iters=J, cache_fractal=False)
        sim.ns_solver.add_smoke_source(N // 2, N // 2, 8, 1.0)
        frame = sim.simulate_step(add_fractal=True)
        for _ in range(2):
            oracle.encoder_features_fast(frame[None], w, input_dim=128)      # warm-up (thread pool, page faults)
        t0 = time.perf_counter()
        t_sim, done = 0.0, 0
        while done < frames and (done < 8 or time.perf_counter() - t0 < budget_s):
            ts = time.perf_counter()
            frame = sim.simulate_step(add_fractal=True)
            t_sim += time.perf_counter() - ts
            oracle.encoder_features_fast(frame[None], w, input_dim=128)
            done += 1
        return time.perf_counter() - t0, t_sim, done

    dt, t_sim, n = run(threads, budget_frames, 12.0)
    dt1, t_sim1, n1 = run(1, max(2, budget_frames // 10), 10.0)           # also with one thread
    return {"value": n / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} frames of one {N}x{N} grid (Jacobi-{J}, fractal recomputed per frame): scalar C stepper + fractal "
                      f"({t_sim / n * 1e3:.1f} ms/frame) + fp32 C encoder ({oracle.encoder_fast_isa()} micro-kernel, OpenMP) on {threads} threads "
                      f"({(dt - t_sim) / n * 1e3:.1f} ms/frame), {dt:.1f} s total",
            "single_thread": {"value": n1 / dt1, "unit": "frames/s", "cores": 1,
                              "sample": f"{n1} frames, {dt1:.1f} s ({t_sim1 / n1 * 1e3:.1f} ms stepper + {(dt1 - t_sim1) / n1 * 1e3:.1f} ms encoder per frame)"},
            "host_cores_available": threads, "host_cores_visible": os.cpu_count(), "cgroup_cpu_quota_cores": quota,
            "encoder_isa": oracle.encoder_fast_isa(),
            "reference_on_8_cores_in_build_container": "6.45 frames/s (BASELINE.md: actual reference code, torch CPU/oneDNN)"}


def dataset_leg(dev, N=128, samples=512, cpu_budget_s=8.0):
    """SyntheticSmokeDataset generation -- per sample 1-3 sources, 20 simulate_step() frames with the fractal
    emit, get_chaos_features() at t >= 10 (Lyapunov over the never-cleared history, box counts, histogram entropy), as the reference's
    data_loader.py:37-99 does it (1.43 s per sample at 128^2 on 8 cores).  Here: batched stepper (64 samples per launch
    chain), one pass of each HIP reduction and one device-to-host copy per chunk; sequences stay on the device.  Beside it the CPU oracle
    looped per sample like the reference (bounded sample)."""
    import oracle
    from smokephysai_amd.utils.data_loader import SyntheticSmokeDataset
    np.random.seed(0)
    SyntheticSmokeDataset(num_samples=64, grid_size=(N, N), device=dev)                # warm-up (kernels, constants, allocator)
    torch.cuda.synchronize(dev)
    np.random.seed(0)
    t0 = time.perf_counter()
    ds = SyntheticSmokeDataset(num_samples=samples, grid_size=(N, N), device=dev)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    assert len(ds) == samples and all(np.isfinite(list(d["chaos_features"].values())).all() for d in ds.data)
    del ds
    # the oracle, one sample at a time (reference loop: setup_grid, sources, 20 steps with the fractal recomputed per frame, labels)
    np.random.seed(0)
    sim = oracle.OracleSmokeSimulator((N, N), jacobi_iters=20, cache_fractal=False)
    c0 = time.perf_counter()
    done = 0
    while done < 2 or time.perf_counter() - c0 < cpu_budget_s:
        sim.ns_solver.setup_grid()
        pos, inten = oracle.draw_sources((N, N))
        sim.add_incense_source(pos, inten)
        for t in range(20):
            sim.simulate_step(add_fractal=True)
            if t >= 10:
                sim.get_chaos_features()
        done += 1
    cdt = time.perf_counter() - c0
    return {"workload": f"SyntheticSmokeDataset: {samples} samples of {N}x{N}, 20 frames each + chaos labels, Jacobi-20, sim_batch 64",
            "value": samples / dt, "unit": "samples/s", "ms_per_sample": dt / samples * 1e3, "frames_per_s": samples * 20 / dt,
            "cpu_port": {"value": done / cdt, "unit": "samples/s", "ms_per_sample": cdt / done * 1e3, "cores": 1,
                         "sample": f"{done} samples looped one at a time on the C oracle, {cdt:.1f} s"},
            "reference_on_8_cores_in_build_container": "1.43 s per sample at 128^2 (BASELINE.md: actual reference code)"}


def config4_measure(dev, steps=6, rank=0, world=1):
    """This rank's share of configs[4] (8 / world volumes of 512 x 512 x 64, Jacobi-20, SPEC_3D.md) as the PIPELINE it is: every timed step is
    one `step_into` of the stepper followed by HipEncoder3D on all the volumes that step emitted, HIP events on the launch stream around
    both parts.  Returns ms per step of: the stepper, the encoder (per volume), the whole step (event to event) -- and the encoder's time
    per volume on DENSE input (U(0, 1.8), the fixtures' dense-frame distribution: matrix-core power depends on the data).  No collective."""
    from smokephysai_amd.models import HipEncoder3D
    from smokephysai_amd.physics import NavierStokesSimulator3D
    B_total, D, H, W, J = 8, 64, 512, 512, 20
    B = B_total // world
    sim = NavierStokesSimulator3D((D, H, W), device=dev, batch_size=B, jacobi_iters=J)
    rng = np.random.RandomState(4 + rank)
    sim.add_smoke_sources([(b, int(rng.randint(40, W - 40)), int(rng.randint(40, H - 40)), int(rng.randint(10, D - 10)), 8,
                            float(rng.uniform(0.5, 2.0))) for b in range(B) for _ in range(3)])
    g = torch.Generator().manual_seed(0)
    w = {"conv1_w": torch.randn(64, 1, 7, 7, 7, generator=g) * 0.05, "conv1_b": torch.randn(64, generator=g) * 0.1,
         "bn1_w": torch.rand(64, generator=g) + 0.5, "bn1_b": torch.randn(64, generator=g) * 0.1, "bn1_mean": torch.randn(64, generator=g) * 0.2,
         "bn1_var": torch.rand(64, generator=g) + 0.3, "conv2_w": torch.randn(128, 64, 3, 3, 3, generator=g) * 0.03,
         "conv2_b": torch.randn(128, generator=g) * 0.1, "bn2_w": torch.rand(128, generator=g) + 0.5, "bn2_b": torch.randn(128, generator=g) * 0.1,
         "bn2_mean": torch.randn(128, generator=g) * 0.2, "bn2_var": torch.rand(128, generator=g) + 0.3}
    enc = HipEncoder3D(w, device=dev)
    frame = torch.empty(B, D, H, W, device=dev)
    for _ in range(2):                                       # untimed: buffers, first-use setup, clocks
        sim.step_into(frame, 1)
        feats = enc(frame)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    torch.cuda.synchronize(dev)
    for k in range(steps):
        ev[k][0].record()
        sim.step_into(frame, 1)
        ev[k][1].record()
        feats = enc(frame)
        ev[k][2].record()
    torch.cuda.synchronize(dev)
    assert torch.isfinite(frame).all() and float(frame.abs().sum()) > 0
    assert feats.shape == (B, 128, 32, 32) and torch.isfinite(feats).all()
    ms_sim = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    ms_enc = float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) / B
    ms_step = ev[0][0].elapsed_time(ev[-1][2]) / steps      # first record to last record: launch gaps between the parts included
    dense = torch.rand(1, D, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 1.8
    enc(dense)
    d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    d0.record()
    for _ in range(3):
        enc(dense)
    d1.record()
    torch.cuda.synchronize(dev)
    return ms_sim, ms_enc, ms_step, d0.elapsed_time(d1) / 3


def sim3d_counters():
    """HBM bytes per configs[4] stepper step from the committed rocprofv3 --pmc passes (tools/pmc_sim3d.sh -> profiles/r04/sim3d_pmc.json):
    replayed, not measured live, with the stamp of the sources they were taken on."""
    f = os.path.join(ROOT, "profiles", "r04", "sim3d_pmc.json")
    if not os.path.exists(f):
        return None
    d = json.load(open(f))
    st = d.get("stamp", {})
    return {"bytes_per_step": d["per_step"]["hbm_bytes"], "fetch": d["per_step"]["hbm_fetch_bytes"], "write": d["per_step"]["hbm_write_bytes"],
            "kernel_us_per_step_under_profiler": d["per_step"]["kernel_us"],
            "by_kernel": {k: {"launches_per_step": e.get("launches_per_step"), "us": e.get("us_per_launch_trace"),
                              "bytes": round(e.get("hbm_fetch_bytes_per_launch", 0) + e.get("hbm_write_bytes_per_launch", 0))}
                          for k, e in d["kernels"].items()},
            "source": {"file": "profiles/r04/sim3d_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x 2 as "
                               "MI355X_MICROARCH.md prescribes for gfx950: it counts the L2's fabric-side requests, Infinity-Cache hits included)",
                       "stamp": st, "stale": st.get("csrc_sha256") != source_stamp()["csrc_sha256"]}}


def config4_block(ms, ms_enc, ms_step, ms_enc_dense, steps=6, world=1):
    """The `config4` block of the bench line from the (max-over-ranks) times of the pipelined steps."""
    B_total, D, H, W, J = 8, 64, 512, 512, 20
    B = B_total // world
    cells = B * D * H * W
    alg = cells * 4.0 * (37 + 3 * J)
    gbs = alg / (ms * 1e-3) / 1e9
    enc_flop = 2.0 * D * H * W * (343 * 64 + 27 * 64 * 128)
    roof_st = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
               "note": "per GPU.  achieved / frac = PASS-MODEL bytes 4*(37+3J) per cell (SPEC_3D.md section 7) over the measured time: work done "
                       "per second in the survey's unit, NOT utilisation of the pins -- the launches move fewer bytes (diffusion + divergence "
                       "fused, four Jacobi sweeps per launch, gradient subtraction + four advections fused).  traffic / measured_GBs / "
                       "frac_measured = HBM-side bytes from the rocprofv3 counters over the live time"}
    c = sim3d_counters()
    if c is not None and world == 1:
        roof_st.update({"traffic": c["bytes_per_step"], "traffic_split": {"fetch": c["fetch"], "write": c["write"]}, "traffic_by_kernel": c["by_kernel"],
                        "traffic_source": c["source"], "measured_GBs": c["bytes_per_step"] / (ms * 1e-3) / 1e9,
                        "frac_measured": c["bytes_per_step"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "compulsory_bytes_per_step": cells * 4.0 * (9 + 3 * ((J + 3) // 4) + 10),
                        "compulsory_note": "4 in + 5 out (diffuse + divergence), ceil(J / 4) Jacobi launches x (p, div in; p out), 5 in + 5 out (advection with the gradient subtraction and the frame)"})
    tf = enc_flop / (ms_enc * 1e-3) / 1e12
    tfd = enc_flop / (ms_enc_dense * 1e-3) / 1e12
    return {"workload": f"configs[4]: {W}x{H}x{D} grid, batch {B_total} ({B} per GPU), Jacobi-{J} (SPEC_3D.md), conv3d encoder -> [B,128,32,32]",
            "value": B_total / (ms_step * 1e-3), "unit": "volumes/s (simulated + encoded, whole job)", "ms_per_step": ms_step, "n_gpus": world,
            "timing": "pipelined: every timed step = the stepper's step then the encoder on the volumes it emitted, one stream, HIP events; "
                      "ms_per_step = first event to last event / steps",
            "volumes_per_gpu": B, "parallelism": f"independent volumes sharded over {world} GPU(s), no data-path collective; times = max over ranks",
            "ms_sim_per_step": ms, "sim_only_volumes_per_s": B_total / (ms * 1e-3), "ms_encode_per_volume": ms_enc,
            "ms_encode_per_volume_dense": ms_enc_dense, "steps": steps, "dtype": "f32 stencil + bf16x3 GEMM",
            "cells_per_step": cells, "algorithmic_bytes_per_step": alg,
            "launches_per_step": "1 + ceil(J / 4) + 1: buoyancy + diffusion + divergence (one z-marching launch), Jacobi in 4-sweep temporally blocked "
                                 "launches (J = 20: five, through a third pressure buffer), gradient subtraction + the four advections (one z-marching launch)",
            "roofline_stencil": roof_st,
            "roofline_encoder": {"bound": "mfma", "kernel": "conv1: k_conv3d_s7_march (weights in registers, fragment tables in LDS), conv2 + depth pooling: k_conv3d_march "
                                                            "(three input planes in LDS, 27 taps per plane from there); k_pool3d_accum on the depth sums",
                                 "achieved": tf, "peak": MFMA_PEAK_TFLOPS["bf16x3"], "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS["bf16x3"],
                                 "achieved_dense": tfd, "frac_dense": tfd / MFMA_PEAK_TFLOPS["bf16x3"],
                                 "note": "per GPU; achieved = the stepper's own volumes (mostly background), achieved_dense = U(0, 1.8) volumes -- the "
                                         "figure the kernel profile under profiles/ (tools/pmc_enc3d.sh, random data) corresponds to; the matrix pipe "
                                         "clocks lower on dense operands.  Features 4e-6 from the fp64 oracle (tolerance 1e-4); the x3 split executes 3 "
                                         "MFMA products per counted multiply (frac 0.33 = the matrix pipe full); conv1's output (4.3 GB per volume) is "
                                         "the one activation still written to HBM"}}


def config4_leg(dev, steps=6, rank=0, world=1, dist=None, backend="nccl"):
    """BASELINE configs[4]: 8 grids of 512 x 512 x 64, Jacobi-20, SPEC_3D.md semantics, through smk_sim3d_step, then the conv3d encoder
    (SPEC_3D.md section 8) on all emitted volumes.  N > 1 (configs[4] is quoted on 8 GPUs): the 8 volumes shard over the ranks with no
    data-path collective (8 / N per GPU; every rank runs this leg); each component is the MAX over ranks.  A rank whose share fails still
    takes part in the one collective of the leg (a failure flag travels with the times), so the others never wait for it."""
    err = None
    vals = [0.0, 0.0, 0.0, 0.0]
    try:
        vals = list(config4_measure(dev, steps, rank, world))
    except Exception as e:        # noqa: BLE001 -- reported in the block; the headline line must still be printed
        if dist is None:
            raise
        err = f"rank {rank}: {type(e).__name__}: {e}"
    if dist is not None:
        t = torch.tensor(vals + [0.0 if err is None else 1.0], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if float(t[4].item()) > 0:
            return {"error": err or "another rank failed in its share of the leg", "n_gpus": world}
        vals = [float(x) for x in t[:4].tolist()]
    return config4_block(*vals, steps=steps, world=world)


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
                for _ in range(5):
                    fwd(x)
                reps = 30 if bs <= 4 else 8
                groups = []
                for _ in range(3):                           # median of three timed groups (a 10-rep single group swung by +-8 % at batch 1)
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        fwd(x)
                    torch.cuda.synchronize(dev)
                    groups.append((time.perf_counter() - t0) / reps / bs * 1e3)
                out[f"batch{bs}"] = sorted(groups)[1]
    res["eager"] = eager
    # ~68.7 GFLOP per 256^2 frame (61.1 at 128^2) for the whole forward, reference formulation
    gflop = {256: 68.7, 128: 61.1}.get(N)
    if gflop:
        res["counted_TFLOPs_at_sim_batch"] = gflop / res[f"batch{frames.shape[0]}"]
    res["body"] = ("libsmokehip split-bf16 kernels: fused encoder, token linears (bias/pos-embed/chaos-term/GELU/residual epilogues), "
                   "flash attention, chaos addend, LayerNorm, conv reconstruction head")
    res["note"] = f"{N}x{N} frames; hipGraph replay (eager launch beside it); reference README: 610.92 ms/frame (hardware unstated)"
    return res


def source_stamp():
    """sha256 over the kernel sources + ABI header (what the committed rocprofv3 counter extracts under profiles/ were
    taken on; tools/stamp_profiles.py writes the same hash into them).  The built .so is hashed too, for the record."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "smokephysai_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "smokephysai_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "smokephysai_amd", "csrc", "Makefile"), os.path.join(ROOT, "include", "smokehip.h")])
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    so = os.path.join(ROOT, "smokephysai_amd", "libsmokehip.so")
    lib = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16] if os.path.exists(so) else None
    return {"csrc_sha256": h.hexdigest()[:16], "lib_sha256": lib}


def attach_counters(out, encoder_dtype):
    """roofline*.traffic / .pmc come from rocprofv3 --pmc passes (tools/profile.sh, tools/pmc_encoder.sh) committed under
    profiles/: they are REPLAYED here, not measured live, so each carries the stamp of the sources it was taken on and
    `stale` says whether this build's kernel sources differ from that stamp."""
    now = source_stamp()
    out["build_stamp"] = now
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        tr = json.load(open(pmc))
        st = tr.get("stamp", {})
        stale = st.get("csrc_sha256") != now["csrc_sha256"]
        for key, kname in (("roofline_stencil", "stencil"), ("roofline_encoder", "encoder")):
            if key in out and kname in tr and tr.get("encoder_dtype", "bf16x3") == encoder_dtype:
                r = out[key]
                r["traffic"] = tr[kname]
                r["traffic_source"] = {"file": "profiles/pmc_traffic.json (replayed from committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)",
                                       "stamp": st, "stale": stale}
                # the honest HBM figure: bytes seen at the counters / live launch time, against the 8 TB/s pin rate
                r["measured_GBs"] = tr[kname] / (r["ms_per_launch"] * 1e-3) / 1e9
                r["frac_measured"] = r["measured_GBs"] / HBM_PEAK_GBS
        if "roofline_stencil" in out and "detail" in tr:
            out["roofline_stencil"]["traffic_by_kernel"] = {
                k: round(v.get("fetch_bytes_per_dispatch", 0) + v.get("write_bytes_per_dispatch", 0))
                for k, v in tr["detail"].items() if any(s in k for s in ("k_jacobi", "k_buoy", "k_advect", "k_grad", "k_div", "k_step"))}
    sq = os.path.join(ROOT, "profiles", "pmc_mfma.json")
    if os.path.exists(sq) and "roofline_encoder" in out and encoder_dtype == "bf16x3":
        d = json.load(open(sq))
        d["stale"] = d.get("stamp", {}).get("csrc_sha256") != now["csrc_sha256"]
        out["roofline_encoder"]["pmc"] = d


def train_step_leg(dev, N, B, dist, world, backend, steps=4, res=None, force_dist=False):
    """train.py's step (zero_grad, batch_losses, backward, clip 1.0, AdamW) on a device-built batch of B frames per rank
    (BASELINE configs[3]'s per-GPU shape), the model wrapped by utils.distributed.wrap_ddp exactly like train.py:main does:
    with N > 1 ranks the 27.8 M fp32 gradients (111 MB) are all-reduced over RCCL in 64 MB buckets overlapped with backward.
    Timing rule of the headline: barrier + synchronize on both sides, max over ranks."""
    import train
    from smokephysai_amd.models import SmokePhysNet
    from smokephysai_amd.models.physics_regularizer import PhysicsRegularizer
    from smokephysai_amd.utils.distributed import ddp_bucket_report, wrap_ddp
    res = {} if res is None else res                         # filled as the leg goes: the watchdog prints what is there if a later part hangs
    if dist is None and force_dist:
        # N = 1 on the REAL backend: a one-rank RCCL group, DistributedDataParallel(device_ids=...), both exchange hooks and the
        # persistent projection in one process -- the N-rank step's code path, so that the 8-GPU run is not its first contact with RCCL
        import datetime
        import torch.distributed as dist
        from smokephysai_amd.utils.distributed import init_distributed
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        res["stage"] = "init_process_group(nccl, world_size=1)"
        init_distributed("nccl", force=True)
        backend = "nccl"
    rank = dist.get_rank() if dist is not None else 0
    torch.manual_seed(0)
    model = SmokePhysNet().to(dev).train()
    res["stage"] = "wrap_ddp"
    ddp = wrap_ddp(model, dev, force=force_dist)
    reg = PhysicsRegularizer()
    opt = torch.optim.AdamW(ddp.parameters(), lr=1e-3, weight_decay=0.01)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)                 # every rank its own frames
    seq = torch.rand(B, 20, N, N, device=dev, generator=g)
    batch = {"input": seq[:, 9:10].contiguous(), "target": seq[:, 10:11].contiguous(),
             "chaos_features": torch.rand(B, 3, device=dev, generator=g), "sequence": seq}

    def step(sync=True):
        import contextlib
        opt.zero_grad()
        ctx = contextlib.nullcontext() if sync or ddp is model else ddp.no_sync()
        with ctx:
            total = train.batch_losses(ddp, reg, batch, dev)[0]
            total.backward()
        torch.nn.utils.clip_grad_norm_(ddp.parameters(), max_norm=1.0)
        opt.step()

    def timed(n, sync=True):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(sync)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el / n * 1e3

    res["stage"] = "warm-up steps"
    for _ in range(3):
        step()
    res["stage"] = "timed steps"
    # two blocks of `steps`; the lower one is the step's time (a block that contains a one-off stall -- the allocator growing, DDP's bucket
    # rebuild after its first iterations: +25 ms per step measured on one run in three -- is not the steady state); both are reported
    blocks = [timed(steps), timed(steps)]
    ms = min(blocks)
    res["ms_per_step_blocks"] = blocks
    nparam = sum(p.numel() for p in model.parameters())
    exchanged = dist is not None
    res.update({"ms_per_step": ms, "frames_per_s": world * B / (ms * 1e-3), "batch_per_gpu": B, "global_batch": world * B, "grid": N,
           "rccl_ranks": world if (dist is not None and backend == "nccl") else 0, "ranks": world,
           "collective_backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend) if dist is not None else None,
           "grad_bytes_fp32": nparam * 4,
           "ddp_wrapped": ddp is not model,
           "note": ("forward + backward" + (f" + gradient all-reduce over {world} rank(s) (DistributedDataParallel on {backend}, 64 MB buckets, "
                                            "overlapped with backward)" if exchanged and ddp is not model
                                            else " (one process, no process group: NO gradient exchange ran)") +
                    " + clip + AdamW; linear GEMMs, attention, LayerNorm and the encoder's BatchNorm/ReLU/pool, both of its convolutions, "
                    "GELU/dropout/residual of the FFN on libsmokehip; the decoder head's transposed convolutions, the loss and AdamW on PyTorch-ROCm")})
    if dist is not None:
        res["stage"] = "no_sync steps"
        res["ms_per_step_no_allreduce"] = timed(steps, sync=False)          # same step under ddp.no_sync(): what the exchange costs
        res["ddp_buckets"] = ddp_bucket_report(ddp)
        # the gradient exchange alone: one flat fp32 all-reduce of the same byte count (algorithm bandwidth = bytes / time;
        # bus bandwidth = 2 (n-1)/n of that -- a ring on xGMI is bound by one 153 GB/s link)
        res["stage"] = "flat all-reduce"
        flat = torch.zeros(nparam, device=dev if backend == "nccl" else "cpu")
        for _ in range(2):
            dist.all_reduce(flat)
        def best_group(f, groups=4, n=5):                  # seconds per call: the lowest of `groups` means over n calls (host-side noise: at
            best = float("inf")                              # one rank both exchanges are tens of microseconds of launch path)
            for _ in range(groups):
                torch.cuda.synchronize(); dist.barrier()
                t0 = time.perf_counter()
                for _ in range(n):
                    f()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / n)
            return best
        ar = best_group(lambda: dist.all_reduce(flat))
        res["allreduce_flat"] = {"bytes": nparam * 4, "ms": ar * 1e3, "algbw_GBs": nparam * 4 / ar / 1e9,
                                 "busbw_GBs": nparam * 4 / ar / 1e9 * 2 * (world - 1) / world}
        # SURVEY 8f-3's alternative on the same bytes: all-to-all of the shards + ordered local sum + all-gather (every xGMI link busy
        # for two hops instead of a ring's 2 (N-1) dependent hops) -- first as flat collectives, then as the DDP hook inside the step
        res["stage"] = "flat direct exchange"
        from smokephysai_amd.utils.distributed import DirectExchangeState, direct_exchange_hook

        class _Bucket:                                       # what DDP hands the hook: the flat gradient bucket
            def __init__(self, t): self.t = t
            def buffer(self): return self.t
        dstate = DirectExchangeState(None, None)

        def direct_once():                                   # the hook itself on the whole gradient as ONE bucket (111 MB)
            direct_exchange_hook(dstate, _Bucket(flat)).wait()
        for _ in range(2):
            direct_once()
        dr = best_group(direct_once)
        res["direct_exchange_flat"] = {"bytes": nparam * 4, "ms": dr * 1e3, "algbw_GBs": nparam * 4 / dr / 1e9,
                                       "what": "utils.distributed.direct_exchange_hook on the flat gradient: all_to_all_single straight from the bucket, "
                                               "smk_reduce_shards (one launch: rank-order fp32 sum / N of the owner's shard), all_gather_into_tensor "
                                               "into the bucket; one rank: nothing to move"}
        del ddp, flat
        import gc
        gc.collect()                                         # (the first wrapper's reducer hooks go with it)
        res["stage"] = "direct-exchange steps"
        ddp = wrap_ddp(model, dev, grad_exchange="direct", force=force_dist)
        for _ in range(2):
            step()
        res["direct_exchange_step"] = {"ms_per_step": timed(steps), "ddp_buckets": ddp_bucket_report(ddp)}
    res.pop("stage", None)
    return res


def second_config(dev, rank, K, W):
    """BASELINE configs[1] -- 128x128 grid, batch 32, Jacobi-20 (the reference's own J), CNN encoder fp32 -- through the same
    step (stepper + encoder per step, HBM-resident), so that it is driver-timed too; this rank's HIP events only."""
    from smokephysai_amd.models.encoder import HipEncoder
    from smokephysai_amd.physics import SmokeSimulator
    B, N, J = 32, 128, 20
    sim = SmokeSimulator((N, N), device=dev, batch_size=B, jacobi_iters=J)
    sim.ns_solver.add_smoke_sources(draw_sources(B, N, seed=100 + rank))
    enc = HipEncoder(encoder_weights(0), device=dev)
    frame = torch.empty(B, N, N, device=dev)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
    for k in range(-W, K):
        if k >= 0: ev[k][0].record()
        sim.ns_solver.step_into(frame, 1, add_fractal=True, fractal_intensity=0.05)
        if k >= 0: ev[k][1].record()
        feats = enc(frame, input_dim=128, dtype="f32")
        if k >= 0: ev[k][2].record()
    torch.cuda.synchronize()
    assert torch.isfinite(feats).all()
    ms_sim = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    ms_enc = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    ms = float(np.mean([e[0].elapsed_time(e[2]) for e in ev]))
    sten = B * N * N * 4.0 * (27 + 3 * J) / (ms_sim * 1e-3) / 1e9
    tf = B * 153728.0 * N * N / (ms_enc * 1e-3) / 1e12
    return {"workload": f"configs[1]: {N}x{N} grid, batch {B}, Jacobi-{J}, CNN encoder fp32", "value": B / (ms * 1e-3), "unit": "frames/s",
            "ms_per_step": ms, "ms_sim_per_step": ms_sim, "ms_encode_per_step": ms_enc, "steps": K, "dtype": "f32",
            "timing": "HIP events on the launch stream (mean over the timed steps of this process)",
            "roofline_stencil": {"bound": "hbm", "achieved": sten, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sten / HBM_PEAK_GBS},
            "roofline_encoder": {"bound": "mfma", "kernel": "k_encoder_f32 (v_mfma_f32_32x32x2_f32)", "achieved": tf,
                                 "peak": MFMA_PEAK_TFLOPS["f32"], "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS["f32"]}}


def parse_args(argv=None):
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
    ap.add_argument("--no-config4", action="store_true", help="skip the configs[4] block (3-D stepper, 512x512x64 x 8)")
    ap.add_argument("--no-dataset", action="store_true", help="skip the dataset-generation leg (samples/s at 128^2, SURVEY 8a row 16)")
    ap.add_argument("--no-config1", action="store_true", help="skip the secondary configs[1] block (128^2 x 32, Jacobi-20, fp32 encoder)")
    ap.add_argument("--train-step", dest="train_step", action="store_true", default=True,
                    help="time train.py's optimisation step (BASELINE configs[3]'s per-GPU shape: --batch frames of --grid^2, full model; "
                         "under DistributedDataParallel when N > 1: the RCCL gradient all-reduce).  On by default at every N, so that the "
                         "1 -> N curve of the DDP step has its N = 1 point (adds about 20 s)")
    ap.add_argument("--no-train-step", dest="train_step", action="store_false", help="skip the train-step leg (profiling runs)")
    ap.add_argument("--force-dist", dest="force_dist", action="store_true", default=True,
                    help="N = 1 only: run the train-step leg in a ONE-rank RCCL process group under DistributedDataParallel (default), so that "
                         "the collective code path of the N-rank job -- RCCL init, DDP buckets, both exchange hooks -- runs on the real backend")
    ap.add_argument("--no-force-dist", dest="force_dist", action="store_false", help="N = 1: the bare model, no process group")
    ap.add_argument("--train-step-limit", type=float, default=240.0,
                    help="seconds after which a stuck train-step leg is abandoned: rank 0 prints the headline line without it")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch only: rendezvous port (0 = pick a free one)")
    ap.add_argument("--dry-run", action="store_true", help="self-launch only: print the child command as JSON and exit")
    return ap.parse_args(argv)


def launch_command(args, argv):
    """The child command of a self-launched N>1 run: one torch.distributed.run agent that starts one rank per GPU."""
    port = args.master_port
    if not port:
        import socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    child = [a for a in argv if a != "--dry-run"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + child


def self_launch(args, argv):
    """`python bench.py --gpus N` with no torchrun environment: start the ranks as fresh child processes.  This parent has
    not touched the GPU (importing torch does not initialise HIP) and never execs: it waits, relays, and returns the code."""
    import subprocess
    cmd = launch_command(args, argv)
    if args.dry_run:
        print(json.dumps({"launch": cmd, "n_gpus": args.gpus}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    lines = 0
    for line in proc.stdout:                                     # rank 0's single JSON line (anything else is passed through to stderr)
        if line.startswith("{") and '"metric"' in line:
            sys.stdout.write(line); sys.stdout.flush(); lines += 1
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"bench.py: expected one JSON line from rank 0, saw {lines}", file=sys.stderr)
        return 1
    return rc


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_run):
        raise SystemExit(self_launch(args, argv))

    # the train-step leg's MIOpen convolutions (the reconstruction head): a find-db of the bench's own, so that entries another run left in
    # the account's (a deterministic run's restricted solvers: smokephysai_amd/utils/miopen_db.py) cannot slow the measured step
    from smokephysai_amd.utils.miopen_db import use_private_find_db
    use_private_find_db("smokephys_bench")

    # stdout carries ONE JSON line and nothing else: native libraries write there too (RCCL prints a five-line version banner when a
    # process group is created), so fd 1 is pointed at stderr for the life of the process and the line goes to a duplicate of the real stdout
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    # one process per GPU; SMK_BENCH_BACKEND=gloo lets a 1-GPU box rehearse the N>1 control path (ranks share cuda:0)
    backend = os.environ.get("SMK_BENCH_BACKEND", "nccl")
    if backend != "nccl" and world > 1:
        # rehearsal ranks SHARE one card: the single-launch projection needs all its workgroups resident at once (one process per GPU is
        # the deployment model), so two processes on one device take the multi-launch form (read once, when libsmokehip loads)
        os.environ.setdefault("SMK_JACOBI_PERSIST", "0")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=600))       # RCCL over xGMI
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=600))

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

    # Leg order: the secondary legs (configs[1], the other encoder arithmetic) run BEFORE the headline on every rank, at every N: a fresh
    # process's first ~20 launches run 10-25 % slower while the clocks ramp (profiles/README.md), and a 20-step timed region is short
    # enough to see it.  Each leg does its own W warm-up steps and times exactly K steps; the order is printed as "leg_order".
    leg_order = []
    config1 = None
    if not args.no_config1 and not args.no_encode:
        config1 = second_config(dev, rank, K, W)
        leg_order.append("config1")
    alt = None
    if not args.no_encode and not args.no_alt and args.encoder_dtype in ("i8x3", "bf16x3"):
        other = "bf16x3" if args.encoder_dtype == "i8x3" else "i8x3"
        a_el, a_sim, a_enc = timed(other)
        alt = {"encoder_dtype": other, "value": world * B * K / a_el, "unit": "frames/s", "ms_per_step": a_el / K * 1e3,
               "ms_encode_per_step": a_enc, "counted_TFLOPs": B * 153728.0 * N * N / (a_enc * 1e-3) / 1e12}
        leg_order.append("alt")
    elapsed, ms_sim, ms_enc = timed(args.encoder_dtype)
    leg_order.append("headline")
    ms_enc_dense = None
    if not args.no_encode and rank == 0:
        # the same encoder launch on DENSE frames (U(0, 1.8), the fixtures' dense-frame distribution): the simulated frames above are
        # > 99 % background and matrix-core power (hence clock) depends on the operand data
        dense = torch.rand(B, N, N, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) * 1.8
        run = (lambda: enc(dense, input_dim=128, dtype="f32")) if args.encoder_dtype == "f32" else (lambda: enc.tokens(dense, input_dim=128, dtype=args.encoder_dtype))
        for _ in range(max(W, 2)):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms_enc_dense = e0.elapsed_time(e1) / K
    config4_sharded = None
    if world > 1 and 8 % world == 0 and not args.no_config4 and not args.no_encode:
        config4_sharded = config4_leg(dev, rank=rank, world=world, dist=dist, backend=backend)      # every rank: 8 / N volumes each
        leg_order.append("config4")

    out = None
    secondary_rc = 0
    if rank == 0:
        frames_total = world * B * K
        stencil_bytes = B * N * N * 4.0 * (27 + 3 * J)          # algorithmic bytes per stencil pass
        enc_flops = B * 153728.0 * N * N                         # algorithmic flop per encoder launch
        sten_gbs = stencil_bytes / (ms_sim * 1e-3) / 1e9
        plan = sim.ns_solver.jacobi_plan()["projection"]
        roof_stencil = {"bound": "hbm", "kernel": "stencil pass (all kernels of one time step, B grids)",
                        "achieved": sten_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sten_gbs / HBM_PEAK_GBS,
                        "traffic": None, "ms_per_launch": ms_sim,
                        "note": "achieved = SURVEY 8(d)'s pass-model bytes 4*(27+3J) per cell over the measured time; it exceeds the pin rate "
                                "because the J Jacobi sweeps run register/LDS-resident and never touch HBM -- frac is therefore NOT a bound. "
                                "frac_measured (counter bytes / time / 8 TB/s) is the HBM figure; on_chip_bound names what limits the sweeps"}
        if plan:
            roof_stencil["on_chip_bound"] = plan
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
            if args.encoder_dtype in ("bf16x3", "i8x3"):
                roof_enc["mfma_work_frac"] = 3.0 * tf / peak if args.encoder_dtype == "bf16x3" else None
            if ms_enc_dense is not None:
                tfd = enc_flops / (ms_enc_dense * 1e-3) / 1e12
                out["encode_only_dense"] = {"input": f"{B} frames of U(0, 1.8) at {N}x{N} (every pixel non-zero), same launch as the headline's encoder",
                                            "ms_per_launch": ms_enc_dense, "frames_per_s": B / (ms_enc_dense * 1e-3), "counted_TFLOPs": tfd,
                                            "frac": tfd / peak, "vs_simulated_frames": ms_enc_dense / ms_enc}
                roof_enc["achieved_dense"] = tfd
                roof_enc["frac_dense"] = tfd / peak
            out.update({"encode_only_frames_per_s": B / (ms_enc * 1e-3), "ms_encode_per_step": ms_enc,
                        "roofline": roof_enc if ms_enc >= ms_sim else roof_stencil,
                        "roofline_stencil": roof_stencil, "roofline_encoder": roof_enc})
        else:
            out["metric"] += " (DIAGNOSTIC: stencil only, not the headline metric)"
            out["roofline"] = roof_stencil
        if alt is not None:
            out["alt"] = alt
        attach_counters(out, args.encoder_dtype)
        if world == 1:
            out["hbm_copy_measured_GBs"] = hbm_copy_gbs(dev)
        # secondary legs: a failure in one is reported in its block (and in the exit code); the headline line is still printed
        def secondary(name, fn):
            try:
                out[name] = fn()
                return 0
            except Exception as e:        # noqa: BLE001
                out[name] = {"error": f"{type(e).__name__}: {e}"[:400]}
                return 4
        if world == 1 and not args.no_encode and not args.no_inference:
            secondary_rc |= secondary("inference_ms_per_frame", lambda: inference_ms(dev, N, frame, args.encoder_dtype))
        if config1 is not None:
            out["config1"] = config1
        if world == 1 and not args.no_dataset:
            secondary_rc |= secondary("dataset", lambda: dataset_leg(dev))
        if world == 1 and not args.no_config4:
            secondary_rc |= secondary("config4", lambda: config4_leg(dev))
        elif config4_sharded is not None:
            out["config4"] = config4_sharded
            secondary_rc |= 4 if "error" in config4_sharded else 0
        if world == 1 and args.cpu_frames > 0:
            secondary_rc |= secondary("cpu_baseline", lambda: cpu_baseline(N, J, weights, args.cpu_frames))
        # what the device had run when the headline's timed region began: the W warm-up steps of the headline leg itself plus the full
        # secondary legs in front of it (a fresh process's first launches run slower while the clocks ramp; README "leg order")
        out["effective_warmup_steps"] = {"headline_leg": W, "config1_leg_before": (W + K) if config1 is not None else 0,
                                         "alt_leg_before": (W + K) if alt is not None else 0,
                                         "total_steps_before_timed_region": W + ((W + K) if config1 is not None else 0) + ((W + K) if alt is not None else 0)}
        out["leg_order"] = leg_order + [k for k in ("inference_ms_per_frame", "dataset", "config4", "cpu_baseline") if k in out and k not in leg_order] + (["train_step"] if args.train_step else [])

    exit_code = secondary_rc if rank == 0 else 0
    if args.train_step:
        # A collective that never completes on one rank would hang every rank: a watchdog on each rank abandons the leg at the same
        # deadline.  Rank 0 still prints the headline line it holds (with the partial train_step block), and then EVERY rank exits
        # non-zero: each has touched the GPU, and an abandoned leg must not look like a clean run to torchrun or the driver.  One lock
        # decides between "finished" and "abandoned", so the line is printed exactly once.
        import threading
        finish = threading.Lock()
        state = {"done": False}
        ts_partial = {}

        def abandon():
            with finish:
                if state["done"]:
                    return
                state["done"] = True
                if rank == 0:
                    out["train_step"] = dict(ts_partial, error=f"abandoned after {args.train_step_limit:.0f} s (watchdog) in stage "
                                                                f"{ts_partial.get('stage')!r}; the keys beside this one were measured before the part that did not return")
                    emit(out)
                os._exit(3)
        dog = threading.Timer(args.train_step_limit, abandon)
        dog.daemon = True
        dog.start()
        del sim, enc
        torch.cuda.empty_cache()
        try:
            ts = train_step_leg(dev, N, B, dist, world, backend, res=ts_partial, force_dist=args.force_dist and world == 1)
        except Exception as e:                                   # the headline number must survive a failure of this leg ...
            ts = dict(ts_partial, error=f"{type(e).__name__}: {e}"[:400])
            exit_code = 4                                        # ... but the run is not clean
        with finish:                                             # (held by the watchdog while it prints and exits: then this never returns)
            state["done"] = True
        dog.cancel()
        if rank == 0:
            out["train_step"] = ts
    if rank == 0:
        emit(out)
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        tdist.destroy_process_group()
    if exit_code:
        sys.stdout.flush()
        raise SystemExit(exit_code)


if __name__ == "__main__":
    main()
