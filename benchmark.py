#!/usr/bin/env python3
"""SmokePhysAI model benchmark on MI355X -- CLI and function surface of the reference's benchmark.py
(load_config / load_model / evaluate_model / print_results / main, same flags and result-dict keys).

Differences, all deliberate: the timer is device-synchronised (the reference's time.time() around model(inputs) has
no sync, benchmark.py:131-133); the test set is simulated on the GPU; the OpenCV optical-flow comparison
(benchmark.py:21-94,161-211) is out of scope (third-party CV baselines; cv2 is not a dependency here).
"""
import argparse
import time

import numpy as np
import torch
import yaml
from tqdm import tqdm

from smokephysai_amd.models.smokephys_net import SmokePhysNet
from smokephysai_amd.utils.data_loader import SyntheticSmokeDataset


def load_config(config_path: str) -> dict:
    with open(config_path, "r") as f:
        return yaml.safe_load(f)


def load_model(config: dict, checkpoint_path: str, device: str) -> SmokePhysNet:
    """benchmark.py:101-114 (a reference best_model.pth loads unchanged: identical state_dict keys)."""
    hw = config.get("mi355x", {}) or {}
    model = SmokePhysNet(input_dim=config["model"]["input_dim"], hidden_dim=config["model"]["hidden_dim"],
                         num_layers=config["model"]["num_layers"], num_heads=config["model"]["num_heads"],
                         chaos_strength=config["model"]["chaos_strength"],
                         encoder_dtype=hw.get("encoder_dtype", "bf16x3")).to(device)
    if checkpoint_path:
        checkpoint = torch.load(checkpoint_path, map_location=device)
        model.load_state_dict(checkpoint["model_state_dict"])
    model.eval()
    return model


def _pearson(a: np.ndarray, b: np.ndarray) -> float:
    """scipy.stats.pearsonr(...)[0] (benchmark.py:142-146) without the scipy dependency."""
    a = a - a.mean(); b = b - b.mean()
    den = np.sqrt((a * a).sum() * (b * b).sum())
    return float((a * b).sum() / den) if den > 0 else float("nan")


def evaluate_model(model, test_loader, device, hip_graph: bool = True):
    """benchmark.py:116-159; inference_time = sum of per-batch forward wall time / dataset size, synchronised.
    hip_graph: the forward is replayed from a captured hipGraph (captured on the first batch of each shape, outside
    the timed region)."""
    model.eval()
    forward = model
    if hip_graph:
        from smokephysai_amd.models import GraphedSmokePhysNet
        forward = GraphedSmokePhysNet(model)
        shapes_seen = set()
    total_mse, total_ssim, total_time = 0.0, 0.0, 0.0
    physics_corr = []
    with torch.no_grad():
        for batch in tqdm(test_loader, desc="Evaluating SmokePhysAI"):
            inputs = batch["input"].to(device)
            targets = batch["target"].to(device)
            chaos_targets = batch["chaos_features"].to(device)
            if hip_graph and tuple(inputs.shape) not in shapes_seen:
                forward(inputs)                       # capture for this batch shape (untimed)
                shapes_seen.add(tuple(inputs.shape))
            torch.cuda.synchronize(device)
            start_time = time.time()
            outputs = forward(inputs)
            torch.cuda.synchronize(device)
            total_time += time.time() - start_time
            total_mse += torch.nn.functional.mse_loss(outputs["reconstructed"], targets).item()
            phys_pred = outputs["physics_features"].cpu().numpy()
            tgt = chaos_targets.cpu().numpy()
            for i in range(phys_pred.shape[0]):
                physics_corr.append(_pearson(phys_pred[i].astype(np.float64), tgt[i].astype(np.float64)))
    return {"mse": total_mse / len(test_loader), "ssim": total_ssim / len(test_loader),
            "physics_correlation": float(np.mean(physics_corr)),
            "inference_time": total_time / len(test_loader.dataset)}


def print_results(model_results, cv_results):
    """benchmark.py:213-234."""
    print("\n" + "=" * 60)
    print(f"{'Model':<20} | {'MSE':<15} | {'Physics Corr':<15} | {'Inference Time (ms)':<15}")
    print("-" * 60)
    print(f"{'SmokePhysAI':<20} | {model_results['mse']:.6f} | {model_results['physics_correlation']:.4f} | "
          f"{model_results['inference_time']*1000:.2f}")
    for method, results in cv_results.items():
        print(f"{method:<20} | {results['mse']:.6f} | {'N/A':<15} | {results['inference_time']*1000:.2f}")
    print("=" * 60)
    print("Note: Physics Correlation measures how well the model predicts chaos features")
    print("      (Lyapunov exponent, Fractal dimension, Entropy) compared to ground truth")


def main():
    parser = argparse.ArgumentParser(description="SmokePhysAI Benchmark")
    parser.add_argument("--config", type=str, default="config/config.yaml", help="Path to configuration file")
    parser.add_argument("--checkpoint", type=str, default=None,
                        help="Path to model checkpoint (omit for random-init weights: timing only)")
    parser.add_argument("--num_samples", type=int, default=50, help="Number of test samples to evaluate")
    args = parser.parse_args()
    config = load_config(args.config)
    if not torch.cuda.is_available():
        raise RuntimeError("benchmark.py needs a ROCm GPU: smokephysai_amd has no CPU fallback")
    device = torch.device("cuda")
    print(f"Using device: {device}")
    model = load_model(config, args.checkpoint, str(device))
    hw = config.get("mi355x", {}) or {}
    test_dataset = SyntheticSmokeDataset(num_samples=args.num_samples, grid_size=tuple(config["data"]["grid_size"]),
                                         device="cuda", sim_batch=hw.get("sim_batch", 64),
                                         jacobi_iters=hw.get("jacobi_iters", 20))
    test_loader = torch.utils.data.DataLoader(test_dataset, batch_size=4, shuffle=False)
    print("\nEvaluating SmokePhysAI model...")
    model_results = evaluate_model(model, test_loader, device,
                                   hip_graph=bool((config.get("mi355x", {}) or {}).get("hip_graph", True)))
    print_results(model_results, {})


if __name__ == "__main__":
    main()
