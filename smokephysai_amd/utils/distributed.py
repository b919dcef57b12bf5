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


def init_distributed(backend: str = None):
    """Initialise from the torchrun environment (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*); no-op for one process.
    Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def wrap_ddp(model: torch.nn.Module, device, sync_bn: bool = False) -> torch.nn.Module:
    """DistributedDataParallel around `model` when a process group exists (gradient all-reduce = mean over ranks)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    if sync_bn:      # reference semantics = one process, whole-batch statistics: exchange the per-channel sums (models/sync_bn.py)
        from ..models.sync_bn import convert_sync_batchnorm
        model = convert_sync_batchnorm(model)
    dev = torch.device(device)
    if dev.type == "cuda":
        return torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], bucket_cap_mb=64,
                                                         gradient_as_bucket_view=True)
    return torch.nn.parallel.DistributedDataParallel(model)


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
    return {"backend": log.get("backend_name", dist.get_backend()), "world_size": dist.get_world_size(),
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
