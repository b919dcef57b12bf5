"""Synthetic smoke dataset on MI355X -- drop-in for src/utils/data_loader.py:10-184.

The reference generates `num_samples` sequences by looping a single-grid simulator in Python
(data_loader.py:44-97).  Here the same samples come from the batched HIP stepper: the source lists are drawn from
the global np.random stream in the reference's exact order (so a given seed gives the same sources), `sim_batch`
grids are simulated per launch, and the chaos-statistics labels come from the HIP reductions -- including the
reference's quirk that the simulator history is never cleared between samples (data_loader.py:46 resets only the
solver), so sample i's Lyapunov window reaches back into sample i-1's frames.
"""
import os
import pickle
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .. import _lib
from ..physics.smoke_simulator import (SmokeSimulator, chaos_stats, entropy_from_hist, fractal_dimension_from_counts,
                                       frame_diff_norms, lyapunov_from_norms)
from .distributed import shard_range


def draw_source_configs(num_samples: int, grid_size: Tuple[int, int]) -> List[Dict]:
    """data_loader.py:49-58: per sample k=randint(1,4), then k x (x, y, intensity), from the global np.random."""
    cfgs = []
    for _ in range(num_samples):
        k = np.random.randint(1, 4)
        pos, inten = [], []
        for _ in range(k):
            x = np.random.randint(20, grid_size[1] - 20)
            y = np.random.randint(20, grid_size[0] - 20)
            intensity = np.random.uniform(0.5, 2.0)
            pos.append((x, y))
            inten.append(intensity)
        cfgs.append({"positions": pos, "intensities": inten})
    return cfgs


def labels_from_stats(diff_norms, box_counts, hists, T: int, off: int, start: int = 10) -> Tuple[dict, List[dict]]:
    """Host part of the labels (data_loader.py:71-88 + smoke_simulator.py:47-140's scalar formulas).
    diff_norms: distances between consecutive frames of the extended history (off frames of the previous sample, then
    this sample's T frames); box_counts / hists: per frame t = start..T-1."""
    d = np.asarray(diff_norms, dtype=np.float64)
    feats = []
    for t in range(start, T):
        e = off + t                                        # index of frame t in the history; history length = e + 1
        if e + 1 < 10:
            continue                                       # smoke_simulator.py:49-50 -> {} (not appended)
        lyap = 0.0
        if e + 1 >= 20:                                    # smoke_simulator.py:69-70
            lyap = lyapunov_from_norms(d[e - 19:e])        # 19 distances between the last 20 frames
        feats.append({"lyapunov_exponent": lyap,
                      "fractal_dimension": fractal_dimension_from_counts(box_counts[t - start]),
                      "entropy": entropy_from_hist(hists[t - start])})
    if feats:
        avg = {k: np.mean([f[k] for f in feats]) for k in ("lyapunov_exponent", "fractal_dimension", "entropy")}
    else:
        avg = {"lyapunov_exponent": 0.0, "fractal_dimension": 1.0, "entropy": 0.0}
    return avg, feats


def chaos_labels(seq: torch.Tensor, prev_tail: Optional[torch.Tensor], start: int = 10) -> Tuple[dict, List[dict]]:
    """Labels of one sample: average over t=start..T-1 of get_chaos_features() (data_loader.py:71-88).
    seq [T,H,W] on the device; prev_tail = the frames the simulator history held before this sample (19 are used)."""
    T = seq.shape[0]
    ext = seq if prev_tail is None or prev_tail.shape[0] == 0 else torch.cat([prev_tail[-19:], seq])
    off = ext.shape[0] - T
    d = frame_diff_norms(ext).cpu().numpy()
    _, box, hist = chaos_stats(seq[start:])
    return labels_from_stats(d, box.cpu().numpy(), hist.cpu().numpy(), T, off, start)


HIST_TAIL = 19      # frames of earlier samples a Lyapunov window can reach back into (smoke_simulator.py:69-79: the last 20 states)


def chunk_chaos_labels(buf: torch.Tensor, n_samples: int, T: int, valid_head: int, start: int = 10) -> List[Tuple[dict, List[dict]]]:
    """Labels of a whole chunk of samples from ONE pass of each reduction and one device-to-host copy per result.

    buf [HIST_TAIL + n_samples * T, H, W]: the last HIST_TAIL frames the simulator history held before this chunk (only the final
    `valid_head` of them are real), then the chunk's samples back to back -- the order in which the reference's never-cleared history
    sees them (data_loader.py:46 resets only the solver), so the distance between the last frame of sample i-1 and the first of sample i
    is simply one more pair of the flat stream.  Launches: smk_frame_diff_norms over the stream, smk_chaos_stats over the chunk's
    frames (all T per sample; t < start is 2x redundant work on a reduction that costs microseconds, and keeps one frame stride)."""
    frames = buf[HIST_TAIL:]
    d = frame_diff_norms(buf).cpu().numpy()                                   # d[k] = ||buf[k+1] - buf[k]||
    _, box, hist = chaos_stats(frames)
    box, hist = box.cpu().numpy(), hist.cpu().numpy()
    out = []
    for i in range(n_samples):
        off = min(HIST_TAIL, valid_head + i * T)                              # earlier frames the history holds for this sample
        lo = HIST_TAIL + i * T - off                                          # buf index of the first of them
        sl = slice(i * T + start, (i + 1) * T)
        out.append(labels_from_stats(d[lo:HIST_TAIL + (i + 1) * T - 1], box[sl], hist[sl], T, off, start))
    return out


class SyntheticSmokeDataset(Dataset):
    """Same constructor/items as the reference (data_loader.py:13-123).  Extra keyword-only knobs:
    sim_batch (grids per launch), jacobi_iters, storage_device (where sequences are kept; default = device),
    rank/world (generate and hold only this rank's contiguous block of the global sample list)."""

    def __init__(self, num_samples: int = 1000, grid_size: Tuple[int, int] = (128, 128), sequence_length: int = 20,
                 device: str = "cuda", cache_path: Optional[str] = None, *, sim_batch: int = 64, jacobi_iters: int = 20,
                 storage_device: Optional[str] = None, rank: int = 0, world: int = 1):
        self.num_samples = num_samples
        self.grid_size = tuple(grid_size)
        self.sequence_length = sequence_length
        self.device = device
        self.cache_path = cache_path
        self.sim_batch = sim_batch
        self.jacobi_iters = jacobi_iters
        self.storage_device = storage_device if storage_device is not None else device
        self.rank, self.world = rank, world
        if self.cache_path and os.path.exists(self.cache_path):
            with open(self.cache_path, "rb") as f:
                self.data = pickle.load(f)
            print(f"Loaded synthetic data from {self.cache_path}")
        else:
            self.data = self._generate_synthetic_data()
            if self.cache_path:
                os.makedirs(os.path.dirname(self.cache_path) or ".", exist_ok=True)
                with open(self.cache_path, "wb") as f:
                    pickle.dump([dict(d, sequence=d["sequence"].cpu()) for d in self.data], f)
                print(f"Saved synthetic data to {self.cache_path}")

    def _generate_synthetic_data(self) -> List[Dict]:
        dev = _lib.require_cuda(self.device, "SyntheticSmokeDataset")
        cfgs = draw_source_configs(self.num_samples, self.grid_size)       # every rank draws the full list
        lo, hi = shard_range(self.num_samples, self.rank, self.world)
        first = max(lo - 1, 0)                                              # one extra sample: its frames seed the history
        data = []
        T = self.sequence_length
        H, W = self.grid_size
        tail, valid_head = None, 0                                          # the history's last HIST_TAIL frames before the chunk
        for c0 in range(first, hi, self.sim_batch):
            c1 = min(c0 + self.sim_batch, hi)
            n = c1 - c0
            sim = SmokeSimulator(self.grid_size, device=dev, batch_size=n, jacobi_iters=self.jacobi_iters)
            srcs = [(i - c0, x, y, 8, inten) for i in range(c0, c1)
                    for (x, y), inten in zip(cfgs[i]["positions"], cfgs[i]["intensities"])]
            sim.ns_solver.add_smoke_sources(srcs)
            # one buffer per chunk: [history tail | sample 0's T frames | sample 1's ...]; the stepper writes the frames in place
            buf = torch.empty(HIST_TAIL + n * T, H, W, device=dev)
            if tail is not None:
                buf[HIST_TAIL - tail.shape[0]:HIST_TAIL] = tail
            seqs = buf[HIST_TAIL:].view(n, T, H, W)
            sim.simulate_sequence(T, add_fractal=True, out=seqs)            # raises here if a persistent projection timed out
            labels = chunk_chaos_labels(buf, n, T, valid_head)
            for i in range(c0, c1):
                if i >= lo:
                    data.append({"sequence": seqs[i - c0].to(self.storage_device).clone(), "chaos_features": labels[i - c0][0],
                                 "source_config": cfgs[i]})
            # the reference's history keeps the last 100 frames across samples (smoke_simulator.py:41-43); 19 can matter
            tail = buf[HIST_TAIL - valid_head:][-HIST_TAIL:].clone()        # (buf[HIST_TAIL - valid_head:] = every real frame in the buffer)
            valid_head = tail.shape[0]
            del sim
        return data

    def __len__(self) -> int:
        return len(self.data)

    def __getitem__(self, idx: int) -> Dict:
        sample = self.data[idx]
        frame_idx = np.random.randint(5, self.sequence_length - 5)         # data_loader.py:108
        return {"input": sample["sequence"][frame_idx].unsqueeze(0),
                "target": sample["sequence"][frame_idx + 1].unsqueeze(0),
                "chaos_features": torch.tensor([sample["chaos_features"]["lyapunov_exponent"],
                                                sample["chaos_features"]["fractal_dimension"],
                                                sample["chaos_features"]["entropy"]], dtype=torch.float32),
                "sequence": sample["sequence"]}


def create_data_loaders(batch_size: int = 16, num_train: int = 800, num_val: int = 200,
                        grid_size: Tuple[int, int] = (128, 128), device: str = "cuda", cache_dir: Optional[str] = None,
                        **dataset_kwargs) -> Tuple[DataLoader, DataLoader]:
    """data_loader.py:126-184.  Sequences live on the device, so the loaders run in-process (num_workers=0): the
    reference forks workers only after generation, and forking after HIP initialisation is not allowed here."""
    rank, world = dataset_kwargs.get("rank", 0), dataset_kwargs.get("world", 1)
    suffix = "" if world == 1 else f".rank{rank}of{world}"
    train_cache = os.path.join(cache_dir, f"train_data{suffix}.pkl") if cache_dir else None
    val_cache = os.path.join(cache_dir, f"val_data{suffix}.pkl") if cache_dir else None
    train_dataset = SyntheticSmokeDataset(num_samples=num_train, grid_size=grid_size, device=device,
                                          cache_path=train_cache, **dataset_kwargs)
    val_dataset = SyntheticSmokeDataset(num_samples=num_val, grid_size=grid_size, device=device,
                                        cache_path=val_cache, **dataset_kwargs)
    train_loader = DataLoader(train_dataset, batch_size=batch_size, shuffle=True, num_workers=0)
    val_loader = DataLoader(val_dataset, batch_size=batch_size, shuffle=False, num_workers=0)
    return train_loader, val_loader
