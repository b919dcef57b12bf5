"""train.py's main() end to end (reference train.py:182-280): yaml -> loaders (the batched HIP simulator generates the
samples) -> model -> epochs of train_epoch / validate_epoch -> scheduler -> best-val checkpoint, in a child process on the GPU; then the
checkpoint it wrote is read back the way the reference's consumers do (benchmark.load_model, reference benchmark.py:101-114)."""
import os
import subprocess
import sys

import pytest
import yaml

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_main_trains_validates_and_writes_a_reference_schema_checkpoint(tmp_path):
    import torch
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "config.yaml")))
    cfg["data"].update(grid_size=[128, 128], num_train=8, num_val=4, cache_dir=str(tmp_path / "cache"))
    cfg["training"].update(batch_size=4, num_epochs=2)
    cfg["mi355x"].update(sim_batch=8, deterministic=True)
    os.makedirs(cfg["data"]["cache_dir"], exist_ok=True)
    path = tmp_path / "config.yaml"
    path.write_text(yaml.dump(cfg))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.pop("MIOPEN_USER_DB_PATH", None)        # train.py itself must move a deterministic run's find-db out of the account's (checked below)
    env["HOME"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--config", str(path)], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = r.stdout
    assert "Using device: cuda:0" in out and "Training completed!" in out
    # mi355x.deterministic restricts MIOpen's solvers: its find results went to a find-db of their own (utils/miopen_db.py), not ~/.config/miopen
    assert os.path.isdir(tmp_path / ".config" / "miopen_deterministic") and not os.path.exists(tmp_path / ".config" / "miopen")
    for e in (1, 2):
        assert f"Epoch {e}/2" in out
    assert out.count("Train Loss:") == 2 and out.count("Val Loss:") == 2 and out.count("Learning Rate:") == 2   # train.py:262-265
    losses = [float(l.split(":")[1]) for l in out.splitlines() if l.startswith(("Train Loss:", "Val Loss:"))]
    # (the reference's mass-conservation term is an MSE of per-frame SUMS over 128 x 128 pixels: ~1e6 for an untrained network)
    assert all(x == x and 0.0 <= x < float('inf') for x in losses), losses
    exps = [d for d in os.listdir(tmp_path / "experiments") if d.startswith("smokephys_")]
    assert len(exps) == 1
    exp = tmp_path / "experiments" / exps[0]
    assert (exp / "config.yaml").exists()                                       # setup_experiment (train.py:25-39)
    # the dataset caches the loaders wrote are the reference's files (data_loader.py:145-147)
    assert sorted(os.listdir(cfg["data"]["cache_dir"])) == ["train_data.pkl", "val_data.pkl"]
    ckpt = torch.load(exp / "best_model.pth", map_location="cpu", weights_only=False)
    assert set(ckpt) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "val_loss", "config"}   # train.py:268-277
    assert ckpt["epoch"] in (0, 1) and ckpt["config"]["data"]["num_train"] == 8
    assert abs(ckpt["val_loss"] - min(losses[1::2])) <= 1e-4 * max(1.0, abs(ckpt["val_loss"]))   # the best validation epoch (printed to 4 decimals)
    assert ckpt["scheduler_state_dict"]["T_max"] == 2 and len(ckpt["optimizer_state_dict"]["param_groups"]) == 1
    sys.path.insert(0, ROOT)
    import benchmark
    model = benchmark.load_model(cfg, str(exp / "best_model.pth"), "cuda:0")     # the reference's reader of this file
    assert not model.training
    sd = model.state_dict()
    assert set(sd) == set(ckpt["model_state_dict"])
    for k in ("feature_proj.weight", "chaos_layers.5.ffn.3.bias", "reconstruction_head.6.weight"):
        assert torch.equal(sd[k].cpu(), ckpt["model_state_dict"][k]), k
    with torch.no_grad():
        y = model(torch.rand(2, 1, 128, 128, device="cuda:0"))
    assert y["reconstructed"].shape == (2, 1, 128, 128) and torch.isfinite(y["reconstructed"]).all()
