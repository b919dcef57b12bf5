"""CPU tests of the host-side mirror: source draw order, label statistics, model/state-dict surface, the train.py loss
decomposition against the reference's captured scalars, shard partitioning, and the N>1 gradient path (gloo, 2 ranks)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from smokephysai_amd.models import PhysicsRegularizer, SmokePhysNet
from smokephysai_amd.utils.data_loader import draw_source_configs
from smokephysai_amd.utils.distributed import shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("N,nsamp", [(64, 2), (128, 1)])
def test_source_draw_order_matches_reference(golden, N, nsamp):
    g = golden(f"dataset_seed0_{N}.npz")
    np.random.seed(0)
    cfgs = draw_source_configs(nsamp, (N, N))
    for i, c in enumerate(cfgs):
        np.testing.assert_array_equal(np.array(c["positions"], dtype=np.int64), g[f"s{i}_positions"])
        np.testing.assert_array_equal(np.array(c["intensities"]), g[f"s{i}_intensities"])


def test_chaos_labels_with_shared_history_quirk(golden):
    """Host label logic on the reference's own frames (stats from the CPU oracle): sample 1's Lyapunov window reaches
    into sample 0 because the reference never clears the history (data_loader.py:46 resets only the solver)."""
    import oracle
    from smokephysai_amd.utils.data_loader import labels_from_stats
    g = golden("dataset_seed0_64.npz")
    s0, s1 = g["s0_sequence"], g["s1_sequence"]

    def stats(seq, prev):
        ext = seq if prev is None else np.concatenate([prev[-19:], seq])
        d = np.array([np.float32(np.sqrt(np.sum((ext[i + 1].astype(np.float64) - ext[i]) ** 2))) for i in range(len(ext) - 1)])
        o = oracle.OracleSmokeSimulator((64, 64))
        box = np.stack([o.box_counts(f) for f in seq[10:]])
        hist = np.stack([o.hist_counts(f) for f in seq[10:]])
        return d, box, hist, len(ext) - len(seq)

    d, box, hist, off = stats(s0, None)
    avg0, f0 = labels_from_stats(d, box, hist, 20, off)
    d, box, hist, off = stats(s1, s0)
    avg1, f1 = labels_from_stats(d, box, hist, 20, off)
    for avg, ref in ((avg0, g["s0_chaos"]), (avg1, g["s1_chaos"])):
        got = [avg["lyapunov_exponent"], avg["fractal_dimension"], avg["entropy"]]
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-7)
    assert len(f0) == 10 and sum(f["lyapunov_exponent"] == 0.0 for f in f0) >= 9     # only t=19 has 20 frames
    d, box, hist, off = stats(s1, None)
    avg1_noquirk, _ = labels_from_stats(d, box, hist, 20, off)
    assert avg1_noquirk["lyapunov_exponent"] != avg1["lyapunov_exponent"] or avg1["lyapunov_exponent"] == 0.0


def test_chaos_scalar_formulas_vs_reference(golden):
    from smokephysai_amd.physics.smoke_simulator import (entropy_from_hist, fractal_dimension_from_counts,
                                                         lyapunov_from_norms)
    g = golden("chaos_stats_64.npz")
    assert abs(lyapunov_from_norms(g["lyap_dists"]) - g["feats"][0]) < 1e-9
    assert abs(fractal_dimension_from_counts(g["box_counts"]) - g["feats"][1]) < 1e-9
    assert abs(entropy_from_hist(g["hist_counts"]) - g["feats"][2]) < 1e-5


def _small_model(g):
    model = SmokePhysNet(input_dim=32, hidden_dim=64, num_layers=2, num_heads=4, output_channels=16)
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w::")}
    assert list(model.state_dict().keys()) == list(sd.keys())          # same key names AND order as the reference
    model.load_state_dict(sd)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model


def test_train_loss_decomposition_matches_reference(golden):
    """One seeded batch through train.py's loss decomposition: the four scalars and the grad norm."""
    import train
    g = golden("train_batch.npz")
    model = _small_model(g).train()
    seq = torch.zeros(2, 20, 128, 128)
    # the continuity term only needs mean|dt| of the sequence: rebuild a sequence with the captured statistic
    batch = {"input": torch.from_numpy(g["inputs"]), "target": torch.from_numpy(g["targets"]),
             "chaos_features": torch.from_numpy(g["chaos_targets"]), "sequence": seq}
    reg = PhysicsRegularizer()
    total, recon, phys, chaos = train.batch_losses(model, reg, batch, "cpu", chaos_noise=torch.from_numpy(g["chaos_noise"]))
    ref_total, ref_recon, ref_phys, ref_chaos = g["losses"]
    assert abs(float(recon) - ref_recon) / ref_recon < 1e-5
    assert abs(float(chaos) - ref_chaos) / ref_chaos < 1e-5
    mass = float(phys)                                                    # sequence of zeros -> continuity 0
    assert abs(mass - float(g["mass"])) / float(g["mass"]) < 1e-5
    assert abs(mass + float(g["seq_mean_abs_dt"]) - ref_phys) / ref_phys < 1e-5
    full = float(recon) + 0.1 * float(chaos) + 0.05 * (mass + float(g["seq_mean_abs_dt"]))
    assert abs(full - ref_total) / ref_total < 1e-5
    total.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
    assert abs(float(gnorm) - float(g["grad_norm"])) / float(g["grad_norm"]) < 1e-4


def test_eval_forward_has_no_cpu_fallback(golden):
    model = _small_model(golden("train_batch.npz")).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        with torch.no_grad():
            model(torch.zeros(1, 1, 64, 64))


def test_full_model_init_matches_reference_checksums(golden):
    """torch.manual_seed(0) + default constructor consumes the RNG in the reference's order: identical weights."""
    g = golden("model_full_checksums.npz")
    torch.manual_seed(0)
    model = SmokePhysNet()
    assert sum(p.numel() for p in model.parameters()) == int(g["nparams"]) == 27782890
    for k, v in model.state_dict().items():
        cs = g[f"cs::{k}"]
        got = np.array([float(v.double().sum()), float(v.double().abs().sum())])
        np.testing.assert_allclose(got, cs, rtol=1e-12, atol=1e-9, err_msg=k)


def test_encoder_route_selection_on_host_tensors():
    """Train mode always means batch-statistics semantics (with or without grad); eval off-GPU is the fused-kernel route, which refuses."""
    model = SmokePhysNet(input_dim=32, hidden_dim=64, num_layers=1, num_heads=4)
    x = torch.zeros(1, 1, 64, 64)
    assert model.train()._encoder_route(x) == "train"
    with torch.no_grad():
        assert model._encoder_route(x) == "train"
    assert model.eval()._encoder_route(x) == "hip"
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        with torch.no_grad():
            model(x)


def test_model_deepcopy_drops_the_device_mirrors():
    """copy.deepcopy / pickling must not share (or try to pickle) libsmokehip handles: the copy rebuilds them on first use."""
    import copy
    import pickle
    from smokephysai_amd.models.linear import TrainableHipLinear
    model = SmokePhysNet(input_dim=32, hidden_dim=64, num_layers=1, num_heads=4)
    clone = pickle.loads(pickle.dumps(copy.deepcopy(model)))
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), clone.state_dict().values()))
    assert list(model.state_dict()) == list(clone.state_dict())
    assert clone._hip is None and clone._hip_dec is None and clone._hip_body is not model._hip_body
    lin = clone.chaos_layers[0].ffn[0]
    assert isinstance(lin, TrainableHipLinear) and lin.hip_train and "_hip_fwd" not in lin.__dict__
    # on CPU tensors the trainable linear is plain F.linear (the HIP route needs a ROCm device)
    x = torch.randn(3, 64)
    assert torch.equal(lin(x), torch.nn.functional.linear(x, lin.weight, lin.bias))


# ---------------------------------------------------------------- N>1: DDP gradient step on 2 gloo ranks
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _ddp_worker(rank, world, port, golden_path, out_path, exchange="rccl"):
    import torch.distributed as dist
    from smokephysai_amd.utils.distributed import all_reduce_mean_scalars, init_distributed, wrap_ddp
    import train
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    r, w, _ = init_distributed("gloo")
    assert (r, w) == (rank, world)
    g = dict(np.load(golden_path))
    model = _small_model(g).train()
    for m in model.modules():                       # per-sample-independent BN so that DDP == single-process full batch
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    ddp = wrap_ddp(model, "cpu", grad_exchange=exchange)
    lo, hi = shard_range(2, rank, world)
    batch = {"input": torch.from_numpy(g["inputs"][lo:hi]), "target": torch.from_numpy(g["targets"][lo:hi]),
             "chaos_features": torch.from_numpy(g["chaos_targets"][lo:hi]), "sequence": torch.zeros(hi - lo, 20, 8, 8)}
    noise = torch.from_numpy(g["chaos_noise"][:, :, lo:hi])
    total, *_ = train.batch_losses(ddp, PhysicsRegularizer(), batch, "cpu", chaos_noise=noise)
    total.backward()
    mean_loss = all_reduce_mean_scalars([total.item()], "cpu")[0]
    from smokephysai_amd.utils.distributed import ddp_bucket_report
    if rank == 0:
        torch.save({"grads": {k: p.grad.clone() for k, p in model.named_parameters()}, "loss": mean_loss,
                    "report": ddp_bucket_report(ddp)}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["rccl", "direct"])
def test_ddp_gradients_equal_single_process(golden, tmp_path, exchange):
    """2 gloo ranks, one sample each: the averaged gradients equal the single-process full-batch gradients -- with the backend's
    all-reduce and with the direct reduce-scatter + all-gather hook (utils/distributed.direct_exchange_hook)."""
    import torch.multiprocessing as mp
    import train
    gpath = os.path.join(os.path.dirname(__file__), "golden", "train_batch.npz")
    out = str(tmp_path / "ddp.pt")
    mp.spawn(_ddp_worker, args=(2, _free_port(), gpath, out, exchange), nprocs=2, join=True)
    res = torch.load(out)
    assert ("direct" in res["report"]["grad_exchange"]) == (exchange == "direct")
    g = golden("train_batch.npz")
    model = _small_model(g).train()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    batch = {"input": torch.from_numpy(g["inputs"]), "target": torch.from_numpy(g["targets"]),
             "chaos_features": torch.from_numpy(g["chaos_targets"]), "sequence": torch.zeros(2, 20, 8, 8)}
    # the mass term is an MSE over the batch of sums -> mean over ranks of per-rank means equals the full-batch mean
    total, *_ = train.batch_losses(model, PhysicsRegularizer(), batch, "cpu", chaos_noise=torch.from_numpy(g["chaos_noise"]))
    total.backward()
    assert abs(res["loss"] - float(total)) / abs(float(total)) < 1e-5
    # relative to the global gradient scale (k_proj.bias has an exactly-zero gradient: softmax ignores a key bias)
    scale = max(float(p.grad.abs().max()) for p in model.parameters())
    for k, p in model.named_parameters():
        assert float((p.grad - res["grads"][k]).abs().max()) / scale < 5e-5, k     # fp32 reduction-order noise (observed 1.5e-5)


def _multibucket_worker(rank, world, port, out_path):
    import torch.distributed as dist
    from smokephysai_amd.utils.distributed import DirectExchangeState, direct_exchange_hook, init_distributed
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    init_distributed("gloo")
    grads = {}
    for mode in ("allreduce", "direct", "direct_bf16"):
        torch.manual_seed(0)
        m = torch.nn.Sequential(*[torch.nn.Linear(64, 64) for _ in range(12)])
        d = torch.nn.parallel.DistributedDataParallel(m, bucket_cap_mb=0.02)       # ~10 buckets per backward
        st = DirectExchangeState(None, torch.bfloat16 if mode == "direct_bf16" else None)
        if mode != "allreduce":
            d.register_comm_hook(st, direct_exchange_hook)
        g = torch.Generator().manual_seed(100 + rank)
        for _ in range(3):
            m.zero_grad()
            d(torch.randn(9, 64, generator=g)).pow(2).mean().backward()
        grads[mode] = torch.cat([p.grad.flatten() for p in m.parameters()])
        grads[mode + "_calls"] = st.calls
    if rank == 0:
        torch.save(grads, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_direct_exchange_with_several_buckets_in_flight(tmp_path, world):
    """The hook never waits on a collective inside a callback, so many buckets per backward cannot deadlock a backend that runs
    callbacks on its worker threads; three steps of a 12-layer stack in ~10 buckets give the all-reduce's gradients.  Three ranks: shard
    sizes that do not divide the buckets (the tail all-reduce carries the remainder) and a mean by a non-power-of-two."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "mb.pt")
    mp.spawn(_multibucket_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    res = torch.load(out)
    assert res["direct_calls"] >= 10
    assert float((res["direct"] - res["allreduce"]).abs().max()) <= 1e-6 * float(res["allreduce"].abs().max())
    # the bf16 wire mode (config key grad_exchange: direct_bf16): both hops carry bf16, the owner's sum stays fp32.  Every element that went
    # through a shard is a bf16 value (the < 4 N elements per bucket left to the tail all-reduce stay fp32), and it is the bf16 rounding of
    # the mean of bf16-rounded contributions: within 2^-8 of the fp32 mean's magnitude scale
    b, ref = res["direct_bf16"], res["allreduce"]
    assert res["direct_bf16_calls"] >= 10
    is_bf16 = b == b.to(torch.bfloat16).to(torch.float32)
    assert float(is_bf16.float().mean()) > 0.9
    assert float((b - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    assert float((b - ref).abs().max()) > 0                                       # it is not the fp32 path under another name


# ---------------------------------------------------------------- N>1: SyncBatchNorm == single-process whole-batch BatchNorm
def _syncbn_worker(rank, world, port, golden_path, out_path):
    import torch.distributed as dist
    from smokephysai_amd.utils.distributed import init_distributed, wrap_ddp
    import train
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    init_distributed("gloo")
    g = dict(np.load(golden_path))
    model = _small_model(g).train()                       # BatchNorm in TRAIN mode: batch statistics
    ddp = wrap_ddp(model, "cpu", sync_bn=True)
    from smokephysai_amd.models.sync_bn import SyncBatchNorm2d
    assert sum(isinstance(m, SyncBatchNorm2d) for m in model.modules()) == 4 and list(model.state_dict()) == [k[3:] for k in g if k.startswith("w::")]
    lo, hi = shard_range(2, rank, world)
    batch = {"input": torch.from_numpy(g["inputs"][lo:hi]), "target": torch.from_numpy(g["targets"][lo:hi]),
             "chaos_features": torch.from_numpy(g["chaos_targets"][lo:hi]), "sequence": torch.zeros(hi - lo, 20, 8, 8)}
    noise = torch.from_numpy(g["chaos_noise"][:, :, lo:hi])
    total, *_ = train.batch_losses(ddp, PhysicsRegularizer(), batch, "cpu", chaos_noise=noise)
    total.backward()
    if rank == 0:
        torch.save({"grads": {k: p.grad.clone() for k, p in model.named_parameters()},
                    "buffers": {k: b.clone() for k, b in model.named_buffers()}}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_sync_batchnorm_ddp_equals_single_process_full_batch(golden, tmp_path):
    """2 gloo ranks, one sample each, train-mode BatchNorm with sync_bn=True: parameter gradients and BatchNorm running statistics equal
    the single-process run on the full batch of 2 (the reference's semantics, train.py:59-93 in one process)."""
    import torch.multiprocessing as mp
    import train
    gpath = os.path.join(os.path.dirname(__file__), "golden", "train_batch.npz")
    out = str(tmp_path / "syncbn.pt")
    mp.spawn(_syncbn_worker, args=(2, _free_port(), gpath, out), nprocs=2, join=True)
    res = torch.load(out)
    g = golden("train_batch.npz")
    model = _small_model(g).train()
    batch = {"input": torch.from_numpy(g["inputs"]), "target": torch.from_numpy(g["targets"]),
             "chaos_features": torch.from_numpy(g["chaos_targets"]), "sequence": torch.zeros(2, 20, 8, 8)}
    total, *_ = train.batch_losses(model, PhysicsRegularizer(), batch, "cpu", chaos_noise=torch.from_numpy(g["chaos_noise"]))
    total.backward()
    scale = max(float(p.grad.abs().max()) for p in model.parameters())
    for k, p in model.named_parameters():
        assert float((p.grad - res["grads"][k]).abs().max()) / scale < 2e-4, k
    for k, b in model.named_buffers():
        if b.dtype.is_floating_point:
            assert torch.allclose(b, res["buffers"][k], rtol=1e-4, atol=1e-6), k
        else:
            assert torch.equal(b, res["buffers"][k]), k
    # and WITHOUT the exchange the per-rank statistics give different gradients (the test has teeth)
    m2 = _small_model(g).train()
    half = {k: v[:1] for k, v in batch.items()}
    t2, *_ = train.batch_losses(m2, PhysicsRegularizer(), half, "cpu", chaos_noise=torch.from_numpy(g["chaos_noise"][:, :, :1]))
    t2.backward()
    w = "input_encoder.0.weight"
    gw, gw2 = dict(model.named_parameters())[w].grad, dict(m2.named_parameters())[w].grad
    assert float((gw2 - gw).abs().max()) / float(gw.abs().max()) > 0.05


def _uneven_worker(rank, world, port, out_path):
    """Ranks with 3 and 2 batches per epoch must run the same number of optimisation steps (DDP hang)."""
    import torch.distributed as dist
    from smokephysai_amd.utils.distributed import init_distributed, wrap_ddp
    import train
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    init_distributed("gloo")
    torch.manual_seed(0)
    model = SmokePhysNet(input_dim=32, hidden_dim=64, num_layers=1, num_heads=4, output_channels=16)
    ddp = wrap_ddp(model, "cpu")
    n = 5 if rank == 0 else 4                             # 9 samples over 2 ranks, batch 2 -> 3 vs 2 batches
    g = torch.Generator().manual_seed(rank)
    items = [{"input": torch.rand(1, 64, 64, generator=g), "target": torch.rand(1, 128, 128, generator=g),
              "chaos_features": torch.rand(3, generator=g), "sequence": torch.rand(20, 8, 8, generator=g)} for _ in range(n)]
    loader = torch.utils.data.DataLoader(items, batch_size=2, shuffle=False)
    opt = torch.optim.AdamW(ddp.parameters(), lr=1e-4)
    m = train.train_epoch(ddp, loader, opt, PhysicsRegularizer(), "cpu", 0, train._NullWriter())
    if rank == 0:                                         # (validate_epoch runs the eval-mode HIP path: GPU only)
        torch.save({"train": m}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_uneven_shards_run_equal_step_counts(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "uneven.pt")
    mp.spawn(_uneven_worker, args=(2, _free_port(), out), nprocs=2, join=True)          # a mismatch would hang in DDP's all-reduce
    res = torch.load(out)
    assert set(res["train"]) == {"total_loss", "recon_loss", "physics_loss", "chaos_loss"}
    assert all(np.isfinite(v) for v in res["train"].values())


def _one_rank_worker(rank, port, out_path):
    """A ONE-rank process group (gloo here; RCCL on the GPU box, tests/test_rccl_world1.py): init_distributed(force=True) must create it,
    wrap_ddp(force=True) must wrap, and both exchange forms must leave the gradients exactly as the bare model computes them."""
    import torch.distributed as dist
    from smokephysai_amd.utils.distributed import ddp_bucket_report, init_distributed, wrap_ddp
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    assert init_distributed("gloo") == (0, 1, 0) and not dist.is_initialized()          # unforced: no group for one process
    assert init_distributed("gloo", force=True) == (0, 1, 0) and dist.is_initialized() and dist.get_world_size() == 1
    res = {}
    for mode in ("bare", "rccl", "direct"):
        torch.manual_seed(3)
        m = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.GELU(), torch.nn.Linear(64, 8))
        assert wrap_ddp(m, "cpu") is m                                                  # unforced: a one-rank group leaves the model bare
        d = m if mode == "bare" else wrap_ddp(m, "cpu", grad_exchange=mode, force=True)
        if mode != "bare":
            assert isinstance(d, torch.nn.parallel.DistributedDataParallel) and ddp_bucket_report(d)["world_size"] == 1
        x = torch.randn(16, 32, generator=torch.Generator().manual_seed(5))
        d(x).square().mean().backward()
        res[mode] = torch.cat([p.grad.flatten() for p in m.parameters()])
    dist.destroy_process_group()
    torch.save({k: v for k, v in res.items()}, out_path)


def test_forced_one_rank_group_wraps_and_changes_nothing(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "one_rank.pt")
    mp.spawn(_one_rank_worker, args=(_free_port(), out), nprocs=1, join=True)
    res = torch.load(out)
    assert torch.equal(res["rccl"], res["bare"]) and torch.equal(res["direct"], res["bare"]) and float(res["bare"].abs().sum()) > 0


def test_private_miopen_find_db(tmp_path, monkeypatch):
    """utils/miopen_db.py: a purpose-named find-db directory unless the user already chose one (deterministic runs must not share the
    account's ordinary find-db: the entries they write slow every later run)."""
    from smokephysai_amd.utils.miopen_db import use_private_find_db
    monkeypatch.setenv("HOME", str(tmp_path))
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    d = use_private_find_db("deterministic")
    assert d == os.path.join(str(tmp_path), ".config", "miopen_deterministic") and os.path.isdir(d) and os.environ["MIOPEN_USER_DB_PATH"] == d
    assert use_private_find_db("smokephys_bench") == d                     # already set: the first choice stands
    assert use_private_find_db("smokephys_bench", force=True).endswith("miopen_smokephys_bench")
