"""Parameter gradients of one train.py loss on the small fixture model: libsmokehip linears vs PyTorch fp32 vs PyTorch fp64."""
import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
import train
from smokephysai_amd.models import SmokePhysNet
from smokephysai_amd.models.linear import TrainableHipLinear
g = np.load("tests/golden/train_batch.npz")
model = SmokePhysNet(input_dim=32, hidden_dim=64, num_layers=2, num_heads=4, output_channels=16)
model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w::")})
model = model.cuda().train()
for m in model.modules():
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
batch = {"input": torch.from_numpy(g["inputs"]), "target": torch.from_numpy(g["targets"]),
         "chaos_features": torch.from_numpy(g["chaos_targets"]), "sequence": torch.zeros(2, 20, 128, 128)}
noise = torch.randn(2, 3, 2, 1, generator=torch.Generator().manual_seed(5)).cuda()
def run(mod, b, nz):
    mod.zero_grad()
    total, *_ = train.batch_losses(mod, mod.physics_regularizer, b, "cuda", chaos_noise=nz)
    total.backward()
    return {k: p.grad.detach().double().cpu() for k, p in mod.named_parameters() if p.grad is not None}, float(total)
res = {}
for hip in (True, False):
    for m in model.modules():
        if isinstance(m, TrainableHipLinear): m.hip_train = hip
    res[hip] = run(model, batch, noise)
m64 = copy.deepcopy(model).double()
for m in m64.modules():
    if isinstance(m, TrainableHipLinear): m.hip_train = False
b64 = {k: v.double() for k, v in batch.items()}
res[64] = run(m64, b64, noise.double())
print("loss", res[True][1], res[False][1], res[64][1])
def rel(a, b): return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
worst = {}
for k in res[64][0]:
    eh, ef = rel(res[True][0][k], res[64][0][k]), rel(res[False][0][k], res[64][0][k])
    print(f"{k:55s} hip {eh:.2e}  f32 {ef:.2e}")
