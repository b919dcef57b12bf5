import sys, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle.encoder3d import encoder3d_features
from smokephysai_amd.models import HipEncoder3D
from test_hip_encoder3d import _weights
for seed, shape in ((1, (8, 32, 32)), (2, (6, 64, 64)), (3, (3, 128, 32))):
    w = _weights(seed)
    vol = np.random.RandomState(seed).rand(*shape).astype(np.float32)
    ref, _ = encoder3d_features(vol, w)
    for mode in ("march", "implicit", "im2col"):
        enc = HipEncoder3D({k: torch.from_numpy(v) for k, v in w.items()}, conv2_mode=mode)
        got = enc(torch.from_numpy(vol)[None].cuda())[0].cpu().numpy()
        print(shape, mode, float(np.abs(got - ref).max() / np.abs(ref).max()))
