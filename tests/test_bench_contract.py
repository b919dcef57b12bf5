"""The bench line the driver consumes: required keys, types and internal consistency, checked on the committed
round-1 line (profiles/r01/bench_default.json) and on bench.py's argument surface.  CPU only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_keys():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01", "bench_default.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    B, K = d["config"]["batch_per_gpu"], d["steps"]
    assert abs(d["value"] - d["n_gpus"] * B * K / (d["ms_per_step"] * 1e-3 * K)) / d["value"] < 1e-6
    for r in (d["roofline"], d["roofline_stencil"], d["roofline_encoder"]):
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert r["traffic"] is None or r["traffic"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]


def test_round3_bench_line_has_the_new_blocks():
    """The driver-like line of the round-3 build (profiles/r03/bench_steps20_driverlike.json): the contract keys plus this round's blocks --
    cpu_baseline on every granted core, the dataset and configs[4] legs, the train step on a one-rank RCCL group, the effective warm-up."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r03", "bench_steps20_driverlike.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "dataset", "config4", "train_step", "effective_warmup_steps", "inference_ms_per_frame"):
        assert k in d, k
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1
    assert abs(d["value"] - 64 * 20 / (d["ms_per_step"] * 1e-3 * 20)) / d["value"] < 1e-6
    cb = d["cpu_baseline"]
    assert cb["cores"] == cb["host_cores_available"] >= 1 and cb["single_thread"]["cores"] == 1 and cb["kind"] == "port"
    ts = d["train_step"]
    assert ts["rccl_ranks"] == 1 and ts["ddp_wrapped"] is True and "error" not in ts and "1 rank(s)" in ts["note"]
    c4 = d["config4"]
    assert c4["algorithmic_bytes_per_step"] == 8 * 64 * 512 * 512 * 4.0 * (37 + 3 * 20)
    assert c4["n_gpus"] == 1 and c4["volumes_per_gpu"] == 8
    assert abs(c4["ms_per_step"] - (c4["ms_sim_per_step"] + 8 * c4["ms_encode_per_volume"])) < 1e-6
    assert d["dataset"]["unit"] == "samples/s" and d["dataset"]["value"] > d["dataset"]["cpu_port"]["value"]
    assert d["effective_warmup_steps"]["headline_leg"] == 5


def test_round4_bench_line_has_the_new_blocks():
    """The driver-like line of the round-4 build (profiles/r04/bench_steps20_driverlike.json): configs[4] timed as a pipeline with measured
    traffic behind its stencil roofline, the encoder's dense-input figures beside the simulated-frame ones, the direct exchange's flat time."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r04", "bench_steps20_driverlike.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "dataset", "config4", "train_step", "inference_ms_per_frame", "encode_only_dense"):
        assert k in d, k
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1
    assert abs(d["value"] - 64 * 20 / (d["ms_per_step"] * 1e-3 * 20)) / d["value"] < 1e-6
    for r in (d["roofline"], d["roofline_stencil"], d["roofline_encoder"]):
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    dense = d["encode_only_dense"]
    assert dense["ms_per_launch"] > d["ms_encode_per_step"] and abs(dense["frac"] - d["roofline_encoder"]["frac_dense"]) < 1e-12
    c4 = d["config4"]
    parts = c4["ms_sim_per_step"] + 8 * c4["ms_encode_per_volume"]
    assert 0.98 * parts <= c4["ms_per_step"] <= 1.1 * parts and abs(c4["value"] - 8 / (c4["ms_per_step"] * 1e-3)) < 1e-6
    assert c4["ms_sim_per_step"] <= 7.0                                   # VERDICT r3 item 1 (was 9.03)
    rs = c4["roofline_stencil"]
    assert rs["traffic"] > 0 and rs["traffic"] < c4["algorithmic_bytes_per_step"] and rs["traffic"] >= rs["compulsory_bytes_per_step"]
    assert abs(rs["frac_measured"] - rs["traffic"] / (c4["ms_sim_per_step"] * 1e-3) / 1e9 / rs["peak"]) < 1e-9
    assert c4["ms_encode_per_volume_dense"] > c4["ms_encode_per_volume"]
    ts = d["train_step"]
    assert ts["rccl_ranks"] == 1 and "error" not in ts and ts["direct_exchange_flat"]["ms"] <= 0.05        # VERDICT r3 item 6 (was 0.262)
    assert len(ts["ms_per_step_blocks"]) == 2 and ts["ms_per_step"] == min(ts["ms_per_step_blocks"]) < 70.0  # (a poisoned MIOpen find-db read 485)
    assert d["ms_sim_per_step"] <= 0.18                                   # VERDICT r3 item 8 (was 0.191)
    assert d["roofline"]["traffic_source"]["stale"] is False and d["roofline"]["pmc"]["stale"] is False      # the counters are of THIS build
    inf = d["inference_ms_per_frame"]
    assert inf["batch4"] < 0.300 and inf["batch1"] < 0.631 and inf["batch64"] <= 0.195                     # round 3's driver record


def test_bench_cli_defaults_are_the_baseline_config():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, check=True).stdout
    for flag in ("--gpus", "--steps", "--warmup", "--grid", "--batch", "--jacobi", "--encoder-dtype"):
        assert flag in out
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'default=256' in src and 'default=64' in src and '"--jacobi", type=int, default=100' in src
