"""bench.py's N>1 control path: `python bench.py --gpus N` must start its own ranks.

CPU: the launch command (dry run), the refusal of a world-size mismatch, and that the parent process never needs a GPU.
GPU (-m gpu): the self-launched 2-rank run on one card with the gloo backend -- sharded stepper + encoder and the DDP
train-step leg -- prints exactly one JSON line.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_self_launch_dry_run_builds_the_torchrun_command():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "7", "--warmup", "2", "--dry-run", "--master-port", "29517"],
                       capture_output=True, text=True, env=_clean_env(), timeout=300)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = d["launch"]
    assert d["n_gpus"] == 4
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index(BENCH)
    child = cmd[i + 1:]
    assert child == ["--gpus", "4", "--steps", "7", "--warmup", "2", "--master-port", "29517"]      # forwarded verbatim, minus --dry-run


def test_self_launch_picks_a_free_port_and_relays_exit_code(tmp_path):
    import bench
    args = bench.parse_args(["--gpus", "2"])
    cmd = bench.launch_command(args, ["--gpus", "2"])
    port = int(cmd[cmd.index("--master-port") + 1])
    assert 1024 < port < 65536
    # the child of a self-launch is started with subprocess (never exec): a failing child's code comes back
    fake = tmp_path / "fake_bench.py"
    fake.write_text("import sys; print('{\"metric\": \"x\"}'); sys.exit(3)\n")
    orig = bench.launch_command
    try:
        bench.launch_command = lambda a, v: [sys.executable, str(fake)]
        assert bench.self_launch(args, ["--gpus", "2"]) == 3
        fake.write_text("print('no json here')\n")
        assert bench.self_launch(args, ["--gpus", "2"]) == 1           # rc 0 but no JSON line -> failure, not silence
    finally:
        bench.launch_command = orig


def test_world_size_mismatch_is_refused():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True,
                       env=_clean_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_self_launched_two_ranks_gloo_rehearsal():
    """`python bench.py --gpus 2` with no torchrun environment, on ONE card (gloo, both ranks on cuda:0): the parent spawns the
    ranks, rank 0 prints one JSON line with n_gpus 2 and the DDP train-step block (2 ranks exchanged gradients)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--grid", "128", "--batch", "4",
                        "--jacobi", "20", "--train-step-limit", "240"], capture_output=True, text=True, env=_clean_env(SMK_BENCH_BACKEND="gloo"),
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    # configs[4] (quoted on 8 GPUs) shards its 8 volumes over the ranks: 4 each here, no collective on the data path
    c4 = d["config4"]
    assert c4["n_gpus"] == 2 and c4["volumes_per_gpu"] == 4 and c4["value"] > 0
    # measured as a pipeline (step -> encode -> step ...): the whole step is the two parts plus the launch gaps between them
    # (every component is the MAX over the two ranks -- which share one card here -- so the whole step may undercut the sum of the maxima)
    parts = c4["ms_sim_per_step"] + 4 * c4["ms_encode_per_volume"]
    assert 0.6 * parts <= c4["ms_per_step"] <= 1.25 * parts, (c4["ms_per_step"], parts)
    assert c4["ms_encode_per_volume_dense"] > 0 and "pipelined" in c4["timing"]
    ts = d["train_step"]
    assert "error" not in ts, ts
    assert ts["ranks"] == 2 and ts["global_batch"] == 8 and ts["ms_per_step"] > 0
    assert ts["ddp_buckets"]["world_size"] == 2 and ts["ddp_buckets"]["grad_bytes"] == 27782890 * 4
    assert ts["allreduce_flat"]["bytes"] == 27782890 * 4
    # the direct reduce-scatter + all-gather alternative is timed beside it: flat and as the DDP hook inside the step
    assert ts["direct_exchange_flat"]["ms"] > 0
    assert ts["direct_exchange_step"]["ms_per_step"] > 0 and "direct" in ts["direct_exchange_step"]["ddp_buckets"]["grad_exchange"]
