import sys, numpy as np, torch
sys.path.insert(0, '.')
from smokephysai_amd.models.encoder import HipEncoder
w = {k: torch.from_numpy(v) for k, v in np.load('tests/golden/encoder_weights.npz').items()}
enc = HipEncoder(w)
for N in (64, 128, 256):
    g = np.load(f'tests/golden/encoder_io_{N}.npz')
    x = torch.from_numpy(g['frames']).cuda(); ref = g['features']
    for dt in ('f32', 'bf16x3', 'i8x3'):
        out = enc(x, input_dim=128, dtype=dt).cpu().numpy()
        d = np.abs(out - ref); scale = np.abs(ref).max()
        big = np.abs(ref) > 1e-3 * scale
        rel = d[big] / np.abs(ref[big])
        print(f"{N:4d} {dt:7s} max-norm rel {d.max()/scale:.2e} | elementwise (|ref| > 1e-3 max): median {np.median(rel):.1e} p99 {np.percentile(rel,99):.1e} max {rel.max():.1e} | frac > 1e-4: {(rel>1e-4).mean():.4f}")
