"""One process per GPU over RCCL (torch.distributed backend "nccl" on ROCm); gloo on CPU for tests.

The hot path needs no collective: grids and frames are independent, so ranks own disjoint contiguous blocks of
grids/samples (shard_range).  The only exchange step is the training gradient all-reduce (DDP, bucketed and
overlapped with backward), SURVEY.md 8(e).
"""
import os
from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of n independent units owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_distributed(backend: str = None, force: bool = False):
    """Initialise from the torchrun environment (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*); no-op for one process unless `force`
    (a world of one rank on the real backend: the collective path of the N-rank job, rehearsed on one GPU).
    Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world == 1:                                      # nobody else has to find the store: take any free port
                import socket
                s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
            else:
                os.environ["MASTER_PORT"] = "29500"
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


class DirectExchangeState:
    """State of `direct_exchange_hook`: the process group, the wire dtype of the shards (None = the gradients' own fp32) and what the
    hook has moved so far (for logging)."""

    def __init__(self, process_group=None, wire_dtype=None):
        self.process_group = process_group
        self.wire_dtype = wire_dtype
        self.calls = 0
        self.bytes_sent = 0


def reduce_shards(shards: torch.Tensor, world: int, q: int, out: torch.Tensor) -> None:
    """The owner's pass of the direct exchange: out[i] = (shards[0][i] + shards[1][i] + ... in rank order, fp32) / world, written in
    out's dtype (fp32 or bf16).  `shards` is the flat [world * q] receive buffer.  Device tensors: ONE libsmokehip launch
    (smk_reduce_shards: every shard read once, the mean written once).  Host tensors (the gloo rehearsal): the same arithmetic in torch."""
    if shards.is_cuda:
        from .. import _lib
        code = {torch.float32: 0, torch.bfloat16: 1}
        _lib.check(_lib.load().smk_reduce_shards(shards.data_ptr(), code[shards.dtype], world, q, q, out.data_ptr(), code[out.dtype],
                                                 torch.cuda.current_stream(shards.device).cuda_stream))
        return
    parts = shards.view(world, q)
    acc = parts[0].to(torch.float32)
    for r in range(1, world):                               # rank order, one add per source: the same sum on every rank and run
        acc = acc + parts[r].to(torch.float32)
    if world > 1:
        acc = acc / world
    out.copy_(acc)


def direct_exchange_hook(state: DirectExchangeState, bucket: dist.GradBucket) -> torch.futures.Future[torch.Tensor]:
    """DDP communication hook: the gradient mean as a DIRECT reduce-scatter + all-gather instead of a ring all-reduce.

    xGMI is a full mesh of point-to-point links (7 per GPU), so a ring moves 2 (N-1)/N of the bucket over ONE link per hop in 2 (N-1)
    dependent hops, while the mesh can carry every rank's shard to its owner at once: (1) all-to-all -- rank r receives shard r of every
    rank's bucket, one hop, all links busy; (2) the owner adds its N copies in rank order (the same order on every run and for every
    element: deterministic, unlike a ring whose summation order depends on the element's position) and divides by N; (3) all-gather of the
    reduced shards, one hop.  Each rank sends and receives (N-1)/N of the bucket twice -- the ring's byte count, in 2 hops instead of 2 (N-1).

    No staging copies: with the gradients' own fp32 on the wire the all-to-all reads the bucket where it lies (the first N * (n // N)
    elements; the < 4 N left over go through one tiny all-reduce), the owner's pass is ONE kernel (`reduce_shards`) that writes the mean
    of its shard (1 / N of the bucket), and the all-gather delivers straight into the bucket.  `state.wire_dtype = torch.bfloat16`
    halves the bytes on both hops (the sum itself stays fp32) at the price of one conversion pass each way; off by default because the
    averaged gradient is then rounded to 8 bits.  One rank: the mean over one copy is the copy -- with fp32 on the wire nothing is moved
    and nothing is launched (RCCL's own one-rank all-reduce is a no-op too, 0.014 ms; its one-rank all-gather is NOT: it copies the 111 MB
    onto themselves in 0.34 ms, so it is not called); the bf16 wire mode still rounds through bf16, which is its observable effect.
    Collectives are the backend's all_to_all_single / all_gather_into_tensor (RCCL on ROCm; gloo on CPU for the tests); the all-gather of
    one bucket overlaps the rest of backward."""
    group = state.process_group if state.process_group is not None else dist.group.WORLD
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    buf = bucket.buffer()
    n = buf.numel()
    q = (n // world) // 4 * 4                               # shard length: a multiple of 4 elements (16-byte pieces for the owner's kernel)
    main, tail = buf[:world * q], buf[world * q:]
    wire = state.wire_dtype or buf.dtype
    # (gloo moves host tensors only: a 1-GPU rehearsal of the multi-rank path stages the bucket through the host)
    via_host = buf.is_cuda and dist.get_backend(group) == "gloo"
    state.calls += 1
    state.bytes_sent += 2 * (world - 1) * q * torch.empty(0, dtype=wire).element_size()
    if tail.numel():                                        # < 4 N elements that do not fill a shard: one tiny all-reduce
        t = tail.cpu() if via_host else tail
        dist.all_reduce(t, group=group)
        tail.copy_(t / world)
    if q == 0 or (world == 1 and wire == buf.dtype and not via_host):
        fut = torch.futures.Future()                        # nothing to exchange: the bucket is its own mean
        fut.set_result(buf)
        return fut
    if via_host:
        src = main.cpu().to(wire)
    elif wire != buf.dtype:
        src = torch.empty(world * q, dtype=wire, device=buf.device)
        reduce_shards(main, 1, world * q, src)               # fp32 -> bf16, one pass
    else:
        src = main
    # hop 1 is enqueued in call order (RCCL: stream-ordered, the host does not block; the bucket's own backward kernels have finished by
    # the time DDP calls the hook); only hop 2 is returned as the future -- no callback ever waits on another collective, so backends that
    # run callbacks on their worker threads (gloo) cannot deadlock with several buckets in flight
    if world > 1:
        recv = torch.empty_like(src)
        dist.all_to_all_single(recv, src, group=group)
    else:
        recv = src                                          # one rank: its own shard is all there is
    in_place = src is main                                  # fp32 on the wire, device tensors: gather straight into the bucket
    out = main if in_place else torch.empty(world * q, dtype=wire, device=src.device)
    # the owner's mean goes to a buffer of its own (q elements: 1 / N of the bucket) rather than into out[rank * q : (rank + 1) * q]:
    # an all-gather whose input aliases its output is legal for NCCL / RCCL but has never met this tree's hardware at N > 1
    mine = torch.empty(q, dtype=wire, device=src.device)
    reduce_shards(recv, world, q, mine)
    if world == 1:                                          # (bf16 wire on one rank: the round trip through bf16 only)
        out[:q].copy_(mine)
        fut = torch.futures.Future()
        fut.set_result(out)
    else:
        fut = dist.all_gather_into_tensor(out, mine, group=group, async_op=True).get_future()

    def finish(_f):
        if not in_place:
            if out.is_cuda:
                reduce_shards(out, 1, world * q, main)       # bf16 -> fp32, one pass
            else:
                main.copy_(out)
        return buf

    return fut.then(finish)


def wrap_ddp(model: torch.nn.Module, device, sync_bn: bool = False, grad_exchange: str = "rccl", force: bool = False) -> torch.nn.Module:
    """DistributedDataParallel around `model` when a process group of more than one rank exists (gradient all-reduce = mean over ranks).
    grad_exchange: "rccl" = the backend's bucketed all-reduce; "direct" / "direct_bf16" = `direct_exchange_hook`.
    force: wrap in a one-rank group too (every collective of the N-rank step then runs on the real backend with N = 1)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return model
    if grad_exchange not in ("rccl", "direct", "direct_bf16"):
        raise ValueError(f"grad_exchange must be 'rccl', 'direct' or 'direct_bf16', not {grad_exchange!r}")
    if sync_bn:      # reference semantics = one process, whole-batch statistics: exchange the per-channel sums (models/sync_bn.py)
        from ..models.sync_bn import convert_sync_batchnorm
        model = convert_sync_batchnorm(model)
    dev = torch.device(device)
    if dev.type == "cuda":
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], bucket_cap_mb=64,
                                                        gradient_as_bucket_view=True)
    else:
        ddp = torch.nn.parallel.DistributedDataParallel(model)
    if grad_exchange != "rccl":
        ddp._smk_exchange_state = DirectExchangeState(None, torch.bfloat16 if grad_exchange == "direct_bf16" else None)
        ddp.register_comm_hook(ddp._smk_exchange_state, direct_exchange_hook)
    return ddp


def ddp_bucket_report(ddp) -> dict:
    """What the gradient exchange of one step looks like: the reducer's bucket sizes (bytes, in all-reduce launch order)
    and the collective backend.  Empty for an unwrapped (single-process) model."""
    if not isinstance(ddp, torch.nn.parallel.DistributedDataParallel):
        return {}
    try:
        log = ddp._get_ddp_logging_data()
    except Exception:                                    # private API: report what is known without it
        log = {}
    sizes = [int(x) for x in str(log.get("bucket_sizes", "")).split(",") if x.strip()]
    total = sum(p.numel() * p.element_size() for p in ddp.parameters() if p.requires_grad)
    st = getattr(ddp, "_smk_exchange_state", None)
    exchange = "rccl all-reduce" if st is None else ("direct reduce-scatter + all-gather" + (" (bf16 wire)" if st.wire_dtype else ""))
    return {"backend": log.get("backend_name", dist.get_backend()), "world_size": dist.get_world_size(), "grad_exchange": exchange,
            "bucket_cap_bytes": int(log.get("bucket_cap_bytes", 0)) or None, "bucket_bytes": sizes,
            "num_buckets": len(sizes) or None, "grad_bytes": int(total),
            "gradient_as_bucket_view": bool(log.get("gradient_as_bucket_view", False))}


def all_reduce_mean_scalars(values, device):
    """Mean over ranks of a few logging scalars (one small all-reduce)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(v) for v in values]
    dev = torch.device(device)
    t = torch.tensor([float(v) for v in values], dtype=torch.float64,
                     device=dev if (dev.type == "cuda" and dist.get_backend() == "nccl") else "cpu")
    dist.all_reduce(t)
    return (t / dist.get_world_size()).tolist()


def all_reduce_weighted_mean(sums, count, device):
    """sum over ranks of `sums` / sum over ranks of `count`: a sample-weighted mean when the ranks hold blocks of different
    size (`sums` = per-rank sum of value x samples, `count` = per-rank samples)."""
    vals = [float(v) for v in sums] + [float(count)]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dev = torch.device(device)
        t = torch.tensor(vals, dtype=torch.float64, device=dev if (dev.type == "cuda" and dist.get_backend() == "nccl") else "cpu")
        dist.all_reduce(t)
        vals = t.tolist()
    n = max(vals[-1], 1.0)
    return [v / n for v in vals[:-1]]


def max_over_ranks(value: int, device) -> int:
    """The largest `value` any rank holds (one tiny all-reduce); `value` itself for a single process."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(value)
    dev = torch.device(device)
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev if (dev.type == "cuda" and dist.get_backend() == "nccl") else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())
