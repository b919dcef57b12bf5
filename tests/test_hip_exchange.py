"""smk_reduce_shards -- the owner's pass of the direct gradient exchange (utils/distributed.reduce_shards): the mean of `world`
shards summed in rank order in fp32 with a true divide, fp32 or bf16 on either side.  Checked against the same arithmetic written in torch
(the form the gloo rehearsal runs on host tensors): bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _torch_form(shards, world, q, out_dtype):
    """The host form (what utils.distributed.reduce_shards runs on gloo's CPU tensors): on the CPU torch's `/ world` is a true divide,
    as in the kernel (PyTorch-ROCm's device kernel multiplies by the reciprocal instead: one ulp apart for world = 3)."""
    dev = shards.device
    shards = shards.cpu()
    parts = shards.view(world, q)
    acc = parts[0].to(torch.float32)
    for r in range(1, world):
        acc = acc + parts[r].to(torch.float32)
    if world > 1:
        acc = acc / world
    return acc.to(out_dtype).to(dev)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("in_dtype,out_dtype", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                                (torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16)])
def test_reduce_shards_equals_the_rank_order_sum(world, in_dtype, out_dtype):
    from smokephysai_amd.utils.distributed import reduce_shards
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(world)
    for q in (4, 1028, 262144 + 8):
        shards = (torch.randn(world * q, device=dev, generator=g) * torch.logspace(-6, 3, world * q, device=dev)).to(in_dtype)
        out = torch.full((q,), float("nan"), device=dev, dtype=out_dtype)
        reduce_shards(shards, world, q, out)
        want = _torch_form(shards, world, q, out_dtype)
        assert torch.equal(out, want), (world, q, float((out.float() - want.float()).abs().max()))


def test_reduce_shards_in_place_on_the_owners_slice():
    """The exchange reduces INTO the bucket: the output is shard `rank` of the receive buffer's sibling, or -- one rank -- the input itself."""
    from smokephysai_amd.utils.distributed import reduce_shards
    dev = torch.device("cuda", 0)
    world, q = 4, 4096
    recv = torch.randn(world * q, device=dev)
    bucket = torch.zeros(world * q, device=dev)
    want = _torch_form(recv, world, q, torch.float32)
    reduce_shards(recv, world, q, bucket[2 * q:3 * q])
    assert torch.equal(bucket[2 * q:3 * q], want) and float(bucket[:2 * q].abs().sum()) == 0 and float(bucket[3 * q:].abs().sum()) == 0
    one = torch.randn(q, device=dev)
    keep = one.clone()
    reduce_shards(one, 1, q, one)
    assert torch.equal(one, keep)


def test_direct_hook_on_one_rank_moves_nothing(tmp_path):
    """World 1 on the real backend: the hook's future resolves to the untouched bucket (the mean over one copy), through the in-place
    all-gather only -- the local passes of the eager form (0.26 ms per 111 MB in round 3) are gone."""
    import subprocess, sys, os, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, json, time
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from smokephysai_amd.utils.distributed import DirectExchangeState, direct_exchange_hook, init_distributed
init_distributed("nccl", force=True)
dev = torch.device("cuda", 0)
class B:
    def __init__(s, t): s.t = t
    def buffer(s): return s.t
res = {{}}
for wire in (None, torch.bfloat16):
    for n in (27782890, 1027):
        t = torch.randn(n, device=dev)
        keep = t.clone()
        st = DirectExchangeState(None, wire)
        out = direct_exchange_hook(st, B(t)).wait()
        torch.cuda.synchronize()
        if wire is None:
            res[f"equal_{{n}}"] = bool(torch.equal(t, keep))
        else:
            q = n // 4 * 4
            res[f"bf16_{{n}}"] = bool(torch.equal(t[:q], keep[:q].to(torch.bfloat16).float())) and bool(torch.equal(t[q:], keep[q:]))
t = torch.randn(27782890, device=dev)
st = DirectExchangeState(None, None)
for _ in range(3):
    direct_exchange_hook(st, B(t)).wait()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    direct_exchange_hook(st, B(t)).wait()
torch.cuda.synchronize()
res["ms"] = (time.perf_counter() - t0) / 20 * 1e3
dist.destroy_process_group()
print("RESULT " + json.dumps(res))
"""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    assert d["equal_27782890"] and d["equal_1027"] and d["bf16_27782890"] and d["bf16_1027"], d
    assert d["ms"] <= 0.05, d                     # the whole 111 MB exchange at world 1 (was 0.262 ms)
