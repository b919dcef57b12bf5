"""BASELINE configs[3]'s code path on the REAL collective backend with one rank (the GPU box has one card): a one-rank RCCL process group,
DistributedDataParallel(device_ids=...) forced around the model (utils.distributed.wrap_ddp(force=True)), both gradient-exchange forms
(RCCL's bucketed all-reduce, the direct all-to-all + all-gather hook), train.py's train_epoch at the per-GPU shape (64 frames of 256 x 256
from the persistent-projection simulator, Jacobi-100) -- everything the 8-GPU job runs except the other seven ranks.  With one rank the
exchange is the identity, so parameters and gradients after two optimisation steps must equal the unwrapped model's bit for bit
(reference: train.py:41-127).  Each scenario runs in a child process (a process group lives for a process).

The step runs with torch.backends.cudnn.deterministic = True: tools/rccl_diag.py (profiles/r04/rccl_diag.txt) compared every parameter's
gradient right after ONE backward -- the only tensors that differ between two runs of the SAME unwrapped step are the weight gradients
of the reconstruction head's three convolutions (reconstruction_head.{0,3,6}.weight: MIOpen's default backward-weights solvers
accumulate with atomics); all other gradients, and every gradient under DDP (bucket views or not, RCCL all-reduce or the direct hook),
are bit-identical already.  With the flag set, MIOpen is restricted to its deterministic solvers and bare == bare == wrapped exactly,
so the test asserts equality and nothing weaker."""
import json
import os
import subprocess
import tempfile
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import hashlib, json, os, sys
sys.path.insert(0, {root!r})
import torch
import torch.distributed as dist
import train
from smokephysai_amd.models import SmokePhysNet
from smokephysai_amd.models.physics_regularizer import PhysicsRegularizer
from smokephysai_amd.physics import SmokeSimulator
from smokephysai_amd.utils.distributed import ddp_bucket_report, init_distributed, wrap_ddp

torch.backends.cudnn.deterministic = True      # MIOpen: deterministic backward-weights solvers for the reconstruction head (see the docstring)
torch.backends.cudnn.benchmark = False
rank, world, local_rank = init_distributed("nccl", force=True)
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", 0)
B, N = 64, 256
sim = SmokeSimulator((N, N), device=dev, batch_size=B, jacobi_iters=100)
sim.ns_solver.add_smoke_sources([(b, 40 + 2 * b, 200 - b, 8, 0.5 + 0.02 * b) for b in range(B)])
assert sim.ns_solver.jacobi_plan()["projection"]["persistent"] is True          # the deployed combination: persistent projection + nccl + DDP
seq = sim.simulate_sequence(20, add_fractal=True)                               # [B, 20, N, N]; raises if a hand-off timed out
batches = [{{"input": seq[:, f:f + 1].contiguous(), "target": seq[:, f + 1:f + 2].contiguous(),
            "chaos_features": torch.full((B, 3), 0.25 * (f - 8)), "sequence": seq}} for f in (9, 12)]


class Null:
    def add_scalar(self, *a, **k): pass


def run(mode):
    torch.manual_seed(1234)
    model = SmokePhysNet().to(dev)
    ddp = model if mode == "bare" else wrap_ddp(model, dev, grad_exchange=mode, force=True)
    if mode != "bare":
        assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    opt = torch.optim.AdamW(ddp.parameters(), lr=1e-3, weight_decay=0.01)
    torch.manual_seed(99)                                                       # dropout masks + chaos noise of the two steps
    metrics = train.train_epoch(ddp, batches, opt, PhysicsRegularizer(), dev, 0, Null())
    # one more step of the simulator in between: the persistent launch and RCCL's kernels share the device within one process
    sim.simulate_sequence(2, add_fractal=True)
    torch.cuda.synchronize()
    hp, hg = hashlib.sha256(), hashlib.sha256()
    for name, p in model.named_parameters():
        hp.update(p.detach().cpu().numpy().tobytes())
        hg.update(p.grad.detach().cpu().numpy().tobytes())
    flat = torch.cat([p.grad.detach().flatten() for p in model.parameters()])
    rep = ddp_bucket_report(ddp) if mode != "bare" else {{}}
    return {{"params": hp.hexdigest(), "grads": hg.hexdigest(), "metrics": metrics, "report": rep}}, flat

out = {{}}
ref, gref = run("bare")
again, gagain = run("bare")
out["bare"] = ref
out["bare_repeatable"] = ref["params"] == again["params"] and ref["grads"] == again["grads"]
out["bare_noise"] = float((gref - gagain).abs().max())
out["grad_max"] = float(gref.abs().max())
for mode in ("rccl", "direct"):
    got, g = run(mode)
    out[mode] = got
    out[mode + "_maxdiff"] = float((g - gref).abs().max())
    out[mode + "_finite"] = bool(torch.isfinite(g).all())
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def _env():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                "HSA_ENABLE_IPC_MODE_LEGACY": "0", "SMK_JACOBI_PERSIST": "1",
                # the child runs MIOpen restricted to deterministic solvers: what it finds must not land in the account's find-db, where an
                # ordinary run (bench.py's train-step leg) would reuse it -- 60 -> 485 ms per step (smokephysai_amd/utils/miopen_db.py)
                "MIOPEN_USER_DB_PATH": tempfile.mkdtemp(prefix="miopen_det_")})
    return env


def test_one_rank_rccl_ddp_step_equals_the_unwrapped_step_bit_for_bit():
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT)], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    for mode in ("rccl", "direct"):
        assert d[mode + "_finite"], mode
        rep = d[mode]["report"]
        assert rep["backend"] == "nccl" and rep["world_size"] == 1 and rep["grad_bytes"] == 111131560, rep
        assert ("direct" in rep["grad_exchange"]) == (mode == "direct"), rep
        # the bare step repeats bit for bit (deterministic MIOpen solvers), so the one-rank exchange -- a mean over one rank -- must
        # change nothing at all
        assert d["bare_repeatable"], ("two runs of the unwrapped step differ", d["bare_noise"])
        assert d[mode]["grads"] == d["bare"]["grads"], (mode, d[mode + "_maxdiff"])
        assert d[mode]["params"] == d["bare"]["params"], mode
        for k, v in d["bare"]["metrics"].items():
            assert abs(d[mode]["metrics"][k] - v) <= 1e-6 * max(1.0, abs(v)), (mode, k)


def test_bench_train_step_leg_runs_on_rccl_at_n1():
    """`python bench.py --gpus 1` (what the driver runs): the train-step leg is the DDP step on a one-rank RCCL group, and says so."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-config1", "--no-alt", "--no-dataset", "--no-config4",
           "--no-inference", "--cpu-frames", "0"]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[:600]      # ONE JSON line: RCCL's version banner must not reach stdout
    ts = json.loads(lines[0])["train_step"]
    assert "error" not in ts, ts
    assert ts["rccl_ranks"] == 1 and ts["ddp_wrapped"] is True and ts["collective_backend"].startswith("rccl"), ts
    assert "gradient all-reduce over 1 rank(s)" in ts["note"], ts["note"]
    assert ts["allreduce_flat"]["bytes"] == 111131560 and ts["ddp_buckets"]["backend"] == "nccl", ts
    assert ts["direct_exchange_step"]["ms_per_step"] > 0
