"""encoder3d.py -- TEST INFRASTRUCTURE ONLY: numpy (fp64) restatement of SPEC_3D.md section 8, the 3-D generalisation of
SmokePhysNet.input_encoder + the two adaptive pools (/root/reference/src/models/smokephys_net.py:24-32,87-91: Conv2d(1,64,7,p3) + BN + ReLU ->
Conv2d(64,128,3,p1) + BN + ReLU -> AdaptiveAvgPool2d(D,D) -> adaptive_avg_pool2d(32,32)), every 2-D operator replaced by its 3-D
counterpart and the depth axis pooled to 1.  The reference has no 3-D encoder; this file is the definition the HIP path is held to, and
tests/test_oracle_golden.py checks it against an independent evaluation of the same operators by torch's CPU conv3d / batch_norm /
adaptive_avg_pool3d (the third-party arithmetic the reference's own encoder runs on)."""
import numpy as np


def conv3d(x, w, b, pad):
    """x [Cin, D, H, W], w [Cout, Cin, k, k, k] -> [Cout, D, H, W] (cross-correlation, zero padding), fp64."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    cin, D, H, W = x.shape
    k = w.shape[2]
    xp = np.pad(x, ((0, 0), (pad, pad), (pad, pad), (pad, pad)))
    out = np.zeros((w.shape[0], D, H, W), np.float64)
    for kz in range(k):
        for ky in range(k):
            for kx in range(k):
                patch = xp[:, kz:kz + D, ky:ky + H, kx:kx + W]                  # [Cin, D, H, W]
                out += np.einsum("oc,cdhw->odhw", w[:, :, kz, ky, kx], patch)
    return out + np.asarray(b, np.float64)[:, None, None, None]


def bn_relu(x, weight, bias, mean, var, eps=1e-5):
    s = np.asarray(weight, np.float64) / np.sqrt(np.asarray(var, np.float64) + eps)
    y = (x - np.asarray(mean, np.float64)[:, None, None, None]) * s[:, None, None, None] + np.asarray(bias, np.float64)[:, None, None, None]
    return np.maximum(y, 0.0)


def adaptive_avg_pool3d(x, out_size):
    """F.adaptive_avg_pool3d: window [floor(o I / O), ceil((o + 1) I / O)) per axis."""
    C = x.shape[0]
    res = x
    for ax, O in zip((1, 2, 3), out_size):
        I = res.shape[ax]
        parts = []
        for o in range(O):
            lo, hi = (o * I) // O, -((-(o + 1) * I) // O)
            parts.append(np.take(res, range(lo, hi), axis=ax).mean(axis=ax, keepdims=True))
        res = np.concatenate(parts, axis=ax)
    return res


def encoder3d_features(vol, w, input_dim=128):
    """vol [D, H, W] -> features [128, 32, 32] (fp64).  w: the 12 tensors (conv1_w [64,1,7,7,7] ... bn2_var [128])."""
    a1 = bn_relu(conv3d(np.asarray(vol, np.float64)[None], w["conv1_w"], w["conv1_b"], 3), w["bn1_w"], w["bn1_b"], w["bn1_mean"], w["bn1_var"])
    a2 = bn_relu(conv3d(a1, w["conv2_w"], w["conv2_b"], 1), w["bn2_w"], w["bn2_b"], w["bn2_mean"], w["bn2_var"])
    p1 = adaptive_avg_pool3d(a2, (1, input_dim, input_dim))
    p2 = adaptive_avg_pool3d(p1, (1, 32, 32))
    return p2[:, 0], a1
