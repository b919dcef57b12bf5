"""Where does the one-rank DDP step differ from the unwrapped step?  (tests/test_rccl_world1.py)

One process, a one-rank RCCL group.  For each mode -- bare (three times), DDP over RCCL with gradient_as_bucket_view True / False, the
direct exchange hook -- ONE forward + backward on the same batch with the same seeds; gradients are cloned right after backward (before
clip and AdamW) and compared per parameter with the first bare run.  Prints, per mode, the parameters whose gradients differ and by how
much, so that a non-reproducible op shows up by the module it feeds.  `SMK_DIAG_DETERMINISTIC=1` repeats everything under
torch.use_deterministic_algorithms(True, warn_only=True)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import train
from smokephysai_amd.models import SmokePhysNet
from smokephysai_amd.models.physics_regularizer import PhysicsRegularizer
from smokephysai_amd.physics import SmokeSimulator
from smokephysai_amd.utils.distributed import DirectExchangeState, direct_exchange_hook, init_distributed


def main():
    if os.environ.get("SMK_DIAG_DETERMINISTIC") == "1" or os.environ.get("SMK_DIAG_CUDNN_DET") == "1":
        from smokephysai_amd.utils.miopen_db import use_private_find_db
        use_private_find_db("deterministic")          # keep the restricted solvers' find results out of the ordinary find-db
    init_distributed("nccl", force=True)
    dev = torch.device("cuda", 0)
    if os.environ.get("SMK_DIAG_DETERMINISTIC") == "1":
        torch.use_deterministic_algorithms(True, warn_only=True)
    if os.environ.get("SMK_DIAG_CUDNN_DET") == "1":           # MIOpen: only solvers marked deterministic (PyTorch-ROCm maps the cudnn flag)
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False
    B, N = int(os.environ.get("SMK_DIAG_B", "64")), 256
    sim = SmokeSimulator((N, N), device=dev, batch_size=B, jacobi_iters=100)
    sim.ns_solver.add_smoke_sources([(b, 40 + 2 * b, 200 - b, 8, 0.5 + 0.02 * b) for b in range(B)])
    seq = sim.simulate_sequence(20, add_fractal=True)
    f = 9
    batch = {"input": seq[:, f:f + 1].contiguous(), "target": seq[:, f + 1:f + 2].contiguous(),
             "chaos_features": torch.full((B, 3), 0.25, device=dev), "sequence": seq}
    reg = PhysicsRegularizer()

    def run(mode):
        torch.manual_seed(1234)
        model = SmokePhysNet().to(dev)
        net = model
        if mode != "bare":
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], bucket_cap_mb=64,
                                                            gradient_as_bucket_view=(mode != "rccl_noview"))
            if mode == "direct":
                net.register_comm_hook(DirectExchangeState(None, None), direct_exchange_hook)
        net.train()
        torch.manual_seed(99)
        total, *_ = train.batch_losses(net, reg, batch, dev)
        total.backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters()}, float(total)

    ref, loss_ref = run("bare")
    out = {"loss": loss_ref, "modes": {}}
    for mode in ("bare", "bare", "rccl", "rccl_noview", "direct"):
        g, loss = run(mode)
        diffs = {}
        for n, t in g.items():
            d = float((t - ref[n]).abs().max())
            if d != 0.0:
                diffs[n] = {"maxdiff": d, "gmax": float(ref[n].abs().max()), "n_diff": int((t != ref[n]).sum()), "numel": t.numel()}
        key = mode
        while key in out["modes"]:
            key += "_again"
        out["modes"][key] = {"loss_equal": loss == loss_ref, "n_params_differ": len(diffs), "differ": diffs}
    dist.destroy_process_group()
    print("DIAG " + json.dumps(out))
    for k, v in out["modes"].items():
        print(k, "loss_equal", v["loss_equal"], "params differing:", v["n_params_differ"])
        for n, d in sorted(v["differ"].items(), key=lambda kv: -kv[1]["maxdiff"])[:12]:
            print("   ", n, d)


if __name__ == "__main__":
    main()
