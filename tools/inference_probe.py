"""Eager vs hipGraph-replayed SmokePhysNet forward (metric M2): equality with pinned noise, ms/frame at batch 1/4/64."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smokephysai_amd.models import SmokePhysNet, GraphedSmokePhysNet

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = SmokePhysNet().to(dev).eval()
g = GraphedSmokePhysNet(model)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for bs in tuple(int(b) for b in os.environ.get('SMK_PROBE_BATCHES', '1,4,64').split(',')):
    x = torch.rand(bs, 1, N, N, device=dev)
    noise = torch.randn(len(model.chaos_layers), 3, bs, 1, device=dev)
    with torch.no_grad():
        ref = model(x, chaos_noise=noise)
        out = g(x, chaos_noise=noise)
        for k in ref:
            d = (ref[k] - out[k]).abs().max().item()
            print(f"batch {bs} {k}: max |eager - graph| = {d:.3e}")
        x2 = torch.rand_like(x)
        ref2 = model(x2, chaos_noise=noise); out2 = g(x2, chaos_noise=noise)
        print("  second input:", max((ref2[k] - out2[k]).abs().max().item() for k in ref2))
        a = g(x)["latent_features"].clone(); b = g(x)["latent_features"].clone()
        print("  unpinned replays differ (fresh noise):", (a - b).abs().max().item() > 0)

        def timeit(f, reps=20):
            for _ in range(3): f()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): f()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
        te = timeit(lambda: model(x)); tg = timeit(lambda: g(x))
        print(f"  eager {te:.3f} ms  graph {tg:.3f} ms  -> {te / bs:.3f} / {tg / bs:.3f} ms per frame")
