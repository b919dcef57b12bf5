"""ctypes/numpy front-end of oracle/smoke_oracle.c (the CPU restatement of the reference hot path).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Every class/function cites the reference
file:line it follows (paths relative to /root/reference).  Pinned against tests/golden/*.npz by
tests/test_oracle_golden.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "libsmoke_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("smoke_oracle.c", "encoder_fast.c", "Makefile")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libsmoke_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.so_linspace.argtypes = [C.c_float, C.c_float, C.c_int64, f32p]
        L.so_add_source.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
        L.so_diffuse.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_double, C.c_double]
        L.so_buoyancy.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_double]
        L.so_divergence.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int, C.c_double]
        L.so_jacobi.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int, C.c_int]
        L.so_grad_subtract.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int, C.c_double]
        L.so_project.argtypes = [f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_double, C.c_int]
        L.so_bilinear.argtypes = [f32p, C.c_int, C.c_int, f32p, f32p, f32p, C.c_int64]
        L.so_advect.argtypes = [f32p, f32p, C.c_int, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_double,
                                C.c_void_p, C.c_void_p]
        L.so_step.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, f32p]
        L.so_perlin.argtypes = [C.c_int, C.c_int, f32p]
        L.so_mandelbrot_counts.argtypes = [C.c_int, C.c_int, C.c_int, u8p]
        L.so_fractal_field.argtypes = [C.c_int, C.c_int, f32p]
        L.so_apply_fractal.argtypes = [f32p, f32p, f32p, C.c_size_t, C.c_double]
        L.so_box_counts.argtypes = [f32p, C.c_int, C.c_int, C.c_float, i64p]
        L.so_hist256.argtypes = [f32p, C.c_size_t, i64p]
        L.so_encoder_frame.argtypes = [f32p, C.c_int, C.c_int, C.c_int] + [f32p] * 12 + [C.c_void_p, f32p]
        L.so_encoder_frame_fast.argtypes = [f32p, C.c_int, C.c_int, C.c_int] + [f32p] * 12 + [f32p]
        L.so_encoder_frame_fast.restype = None
        L.so_encoder_fast_isa.argtypes = []
        L.so_encoder_fast_isa.restype = C.c_char_p
        L.so_set_threads.argtypes = [C.c_int]
        L.so_set_threads.restype = None
        for fn in ("so_linspace so_add_source so_diffuse so_buoyancy so_divergence so_jacobi so_grad_subtract "
                   "so_project so_bilinear so_advect so_step so_perlin so_mandelbrot_counts so_fractal_field "
                   "so_apply_fractal so_box_counts so_hist256 so_encoder_frame").split():
            getattr(L, fn).restype = None
        _LIB = L
    return _LIB


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def linspace(start, end, steps):
    out = np.empty(steps, np.float32)
    lib().so_linspace(start, end, steps, out)
    return out


class OracleNS:
    """Un-batched stable-fluids stepper; mirrors NavierStokesSimulator (navier_stokes.py:6-173)."""

    def __init__(self, grid_size=(128, 128), dt=0.01, viscosity=0.001, jacobi_iters=20):
        self.h, self.w = grid_size
        self.dt, self.viscosity, self.jacobi_iters = dt, viscosity, jacobi_iters
        self.setup_grid()

    def setup_grid(self):                                  # navier_stokes.py:24-35
        h, w = self.h, self.w
        self.u = np.zeros((h + 1, w), np.float32)
        self.v = np.zeros((h, w + 1), np.float32)
        self.p = np.zeros((h, w), np.float32)
        self.density = np.zeros((h, w), np.float32)
        self._scratch = np.empty(3 * (h + 1) * (w + 1), np.float32)

    def add_smoke_source(self, x, y, radius=10, intensity=1.0):   # navier_stokes.py:37-48
        lib().so_add_source(self.density, self.h, self.w, int(x), int(y), int(radius), float(intensity))

    def diffusion_step(self, field, viscosity):            # navier_stokes.py:50-72
        f = _c(field)
        out = np.empty_like(f)
        lib().so_diffuse(f, out, f.shape[0], f.shape[1], self.dt, viscosity)
        return out

    def buoyancy(self):                                    # navier_stokes.py:154-155
        lib().so_buoyancy(self.v, self.density, self.h, self.w, self.dt)

    def divergence(self):                                  # navier_stokes.py:136
        div = np.empty((self.h, self.w), np.float32)
        lib().so_divergence(self.u, self.v, div, self.h, self.w, self.dt)
        return div

    def pressure_projection(self, iters=None):             # navier_stokes.py:133-149
        div = np.empty((self.h, self.w), np.float32)
        tmp = np.empty((self.h, self.w), np.float32)
        lib().so_project(self.u, self.v, self.p, div, tmp, self.h, self.w, self.dt,
                         self.jacobi_iters if iters is None else iters)
        return div

    def advection_step(self, field, u, v, want_indices=False):    # navier_stokes.py:74-95
        f = _c(field)
        out = np.empty_like(f)
        if want_indices:
            x0 = np.empty(f.shape, np.int64)
            y0 = np.empty(f.shape, np.int64)
            lib().so_advect(f, out, f.shape[0], f.shape[1], _c(u), _c(v), self.h, self.w, self.dt,
                            x0.ctypes.data, y0.ctypes.data)
            return out, x0, y0
        lib().so_advect(f, out, f.shape[0], f.shape[1], _c(u), _c(v), self.h, self.w, self.dt, None, None)
        return out

    def step(self):                                        # navier_stokes.py:151-173
        lib().so_step(self.u, self.v, self.p, self.density, self.h, self.w, self.dt, self.viscosity,
                      self.jacobi_iters, self._scratch)
        return self.density.copy()


def bilinear_interpolate(field, y, x):                     # navier_stokes.py:111-131
    f = _c(field)
    y, x = _c(y), _c(x)
    out = np.empty_like(x)
    lib().so_bilinear(f, f.shape[0], f.shape[1], y, x, out, x.size)
    return out


def interpolate_velocity_u(u, y, x):                       # navier_stokes.py:97-102: x + 0.5 clamped to [0, u.shape[1] - 1]
    xs = np.clip(_c(x) + np.float32(0.5), np.float32(0), np.float32(np.asarray(u).shape[1] - 1)).astype(np.float32)
    return bilinear_interpolate(u, y, xs)


def interpolate_velocity_v(v, y, x):                       # navier_stokes.py:104-109: y + 0.5 clamped to [0, v.shape[0] - 1]
    ys = np.clip(_c(y) + np.float32(0.5), np.float32(0), np.float32(np.asarray(v).shape[0] - 1)).astype(np.float32)
    return bilinear_interpolate(v, ys, x)


def perlin(h, w):                                          # fractal_generator.py:12-31
    out = np.empty((w, h), np.float32)
    lib().so_perlin(h, w, out)
    return out


def mandelbrot_counts(h, w, iterations=100):               # fractal_generator.py:33-51
    out = np.empty((w, h), np.uint8)
    lib().so_mandelbrot_counts(h, w, iterations, out)
    return out


def fractal_field(h, w):                                   # fractal_generator.py:58-59
    out = np.empty((w, h), np.float32)
    lib().so_fractal_field(h, w, out)
    return out


def apply_fractal_perturbation(field, intensity=0.1, fractal=None):   # fractal_generator.py:53-62
    f = _c(field)
    if fractal is None:
        fractal = fractal_field(*f.shape[-2:])             # recomputed per call, as the reference does
    out = np.empty_like(f)
    lib().so_apply_fractal(f, _c(fractal), out, f.size, intensity)
    return out


class OracleSmokeSimulator:
    """Mirrors SmokeSimulator (smoke_simulator.py:8-139): stepper + fractal frame perturbation + history."""

    def __init__(self, grid_size=(128, 128), dt=0.01, viscosity=0.001, jacobi_iters=20, cache_fractal=False):
        self.ns_solver = OracleNS(grid_size, dt, viscosity, jacobi_iters)
        self.history, self.max_history = [], 100
        self._fractal = fractal_field(*grid_size) if cache_fractal else None

    def add_incense_source(self, positions, intensities):  # smoke_simulator.py:26-29
        for (x, y), inten in zip(positions, intensities):
            self.ns_solver.add_smoke_source(x, y, radius=8, intensity=inten)

    def simulate_step(self, add_fractal=True):             # smoke_simulator.py:31-45
        density = self.ns_solver.step()
        if add_fractal:
            density = apply_fractal_perturbation(density, 0.05, self._fractal)
        self.history.append(density.copy())
        if len(self.history) > self.max_history:
            self.history.pop(0)
        return density

    # ---- chaos statistics (smoke_simulator.py:47-140) ----
    def get_chaos_features(self):
        if len(self.history) < 10:
            return {}
        return {"lyapunov_exponent": self.compute_lyapunov_exponent(),
                "fractal_dimension": self.compute_fractal_dimension(),
                "entropy": self.compute_entropy()}

    def compute_lyapunov_exponent(self):                   # smoke_simulator.py:67-87
        if len(self.history) < 20:
            return 0.0
        st = self.history[-20:]
        d = np.array([float(np.float32(np.sqrt(np.sum((st[i + 1] - st[i]).astype(np.float64) ** 2)))) for i in range(19)])
        return max(0, float(np.mean(np.diff(np.log(d + 1e-8)))))

    def box_counts(self, frame=None):                      # smoke_simulator.py:96-115
        cur = self.history[-1] if frame is None else frame
        mean = np.float32(cur.astype(np.float64).mean())
        counts = np.empty(5, np.int64)
        lib().so_box_counts(_c(cur), cur.shape[0], cur.shape[1], float(mean), counts)
        return counts

    def compute_fractal_dimension(self):                   # smoke_simulator.py:89-124
        counts = self.box_counts()
        slope = np.polyfit(np.log([2, 4, 8, 16, 32]), np.log(counts + 1), 1)[0]
        return abs(float(slope))

    def hist_counts(self, frame=None):                     # smoke_simulator.py:134-135
        cur = self.history[-1] if frame is None else frame
        hist = np.empty(256, np.int64)
        lib().so_hist256(_c(cur).ravel(), cur.size, hist)
        return hist

    def compute_entropy(self):                             # smoke_simulator.py:126-140
        hist = self.hist_counts().astype(np.float32)
        probs = hist / hist.sum()
        return float(-(probs * np.log2(probs + np.float32(1e-8))).sum())


def encoder_features(frames, weights, input_dim=128, want_conv1=False):
    """input_encoder + pools (smokephys_net.py:24-32,87-91) on frames [B,H,W] -> [B,128,32,32] (eval-mode BN)."""
    frames = _c(frames)
    B, H, W = frames.shape
    keys = ["conv1_w", "conv1_b", "bn1_w", "bn1_b", "bn1_mean", "bn1_var",
            "conv2_w", "conv2_b", "bn2_w", "bn2_b", "bn2_mean", "bn2_var"]
    ws = [_c(weights[k]) for k in keys]
    out = np.empty((B, 128, 32, 32), np.float32)
    c1 = np.empty((B, 64, H, W), np.float32) if want_conv1 else None
    for b in range(B):
        lib().so_encoder_frame(frames[b], H, W, input_dim, *ws,
                               c1[b].ctypes.data if want_conv1 else None, out[b])
    return (out, c1) if want_conv1 else out


def set_threads(n):
    """OpenMP threads used by encoder_features_fast."""
    lib().so_set_threads(int(n))


def encoder_fast_isa():
    """Which clone of the timing-grade encoder's micro-kernel this host runs ("avx512f", "avx2+fma" or "sse2")."""
    return lib().so_encoder_fast_isa().decode()


def encoder_features_fast(frames, weights, input_dim=128):
    """Timing-grade variant of encoder_features (fp32 accumulation, vectorised rows, OpenMP): bench.py's CPU baseline."""
    frames = _c(frames)
    B, H, W = frames.shape
    keys = ["conv1_w", "conv1_b", "bn1_w", "bn1_b", "bn1_mean", "bn1_var",
            "conv2_w", "conv2_b", "bn2_w", "bn2_b", "bn2_mean", "bn2_var"]
    ws = [_c(weights[k]) for k in keys]
    out = np.empty((B, 128, 32, 32), np.float32)
    for b in range(B):
        lib().so_encoder_frame_fast(frames[b], H, W, input_dim, *ws, out[b])
    return out


def draw_sources(grid_size, rng=np.random):
    """Source draw order of data_loader.py:49-58 (k, then per source x, y, intensity)."""
    k = rng.randint(1, 4)
    pos, inten = [], []
    for _ in range(k):
        x = rng.randint(20, grid_size[1] - 20)
        y = rng.randint(20, grid_size[0] - 20)
        i = rng.uniform(0.5, 2.0)
        pos.append((x, y))
        inten.append(i)
    return pos, inten


# ---------------------------------------------------------------------------------------------------------------------
# Transformer body: numpy restatements in the REFERENCE's own formulation (the chaos scores are a
# separate term here, not folded into Q), double precision unless the reference's fp32 order matters.  Checker only.
def linear(x, weight, bias=None):                             # nn.Linear (smokephys_net.py:38,50-54,153-158)
    y = np.asarray(x, np.float64) @ np.asarray(weight, np.float64).T
    return y if bias is None else y + np.asarray(bias, np.float64)


def gelu(v):                                                  # nn.GELU() default = erf form (smokephys_net.py:155)
    from scipy.special import erf
    v = np.asarray(v, np.float64)
    return 0.5 * v * (1.0 + erf(v / np.sqrt(2.0)))


def layernorm(x, weight, bias, eps=1e-5):                     # nn.LayerNorm (smokephys_net.py:149-150): biased variance
    x = np.asarray(x, np.float64)
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * np.asarray(weight, np.float64) + np.asarray(bias, np.float64)


def lorenz_states(noise, sigma=10.0, rho=28.0, beta=8.0 / 3.0, dt=0.01):
    """chaos_attention.py:39-59: x0,y0,z0 = randn(B,1)*0.1 (noise [3,B] = the three draws), five explicit-Euler steps, in
    fp32 with the reference's operation order.  Returns [B,5,3] float32."""
    f = np.float32
    n = np.asarray(noise, np.float32).reshape(3, -1)
    x, y, z = n[0] * f(0.1), n[1] * f(0.1), n[2] * f(0.1)
    sigma, rho, beta, dt = f(sigma), f(rho), f(beta), f(dt)
    out = []
    for _ in range(5):
        dx = sigma * (y - x)
        dy = x * (rho - z) - y
        dz = x * y - beta * z
        x, y, z = x + dt * dx, y + dt * dy, z + dt * dz
        out.append(np.stack([x, y, z], -1))
    return np.stack(out, 1).astype(np.float32)


def chaos_field(noise, seq_len):                              # chaos_attention.py:61-65: the 5 states tiled along the sequence
    s = lorenz_states(noise)
    reps = (seq_len + 4) // 5
    return np.tile(s, (1, reps, 1))[:, :seq_len]


def chaos_attention(x, w, noise, num_heads, chaos_strength=0.1, temperature=1.0, prefix=""):
    """ChaosAttention.forward (chaos_attention.py:68-114) as written there: scores + strength * gate * chaos_scores, softmax,
    @ V, merge heads, out_proj.  w: dict of the module's tensors (keys '<prefix>q_proj.weight', ...)."""
    g = lambda k: np.asarray(w[prefix + k], np.float64)
    x = np.asarray(x, np.float64)
    B, L, D = x.shape
    H, d = num_heads, D // num_heads
    heads = lambda t: t.reshape(B, L, H, d).transpose(0, 2, 1, 3)
    q = heads(linear(x, g("q_proj.weight"), g("q_proj.bias")))
    k = heads(linear(x, g("k_proj.weight"), g("k_proj.bias")))
    v = heads(linear(x, g("v_proj.weight"), g("v_proj.bias")))
    scores = q @ k.transpose(0, 1, 3, 2) / np.sqrt(d)
    cf = linear(chaos_field(noise, L), g("chaos_proj.weight"), g("chaos_proj.bias"))          # [B,L,D]
    gate = 1.0 / (1.0 + np.exp(-linear(cf, g("chaos_gate.weight"), g("chaos_gate.bias"))))   # [B,L,1]
    chaos_scores = heads(cf) @ k.transpose(0, 1, 3, 2) / np.sqrt(d)
    final = (scores + chaos_strength * chaos_scores * gate[:, None]) / temperature
    final = final - final.max(-1, keepdims=True)
    p = np.exp(final)
    p /= p.sum(-1, keepdims=True)
    out = (p @ v).transpose(0, 2, 1, 3).reshape(B, L, D)
    return linear(out, g("out_proj.weight"), g("out_proj.bias"))


def chaos_transformer_layer(x, w, noise, num_heads, chaos_strength=0.1):
    """ChaosTransformerLayer.forward in eval mode (smokephys_net.py:161-167): pre-LN block, dropout off."""
    x = np.asarray(x, np.float64)
    x = x + chaos_attention(layernorm(x, w["norm1.weight"], w["norm1.bias"]), w, noise, num_heads, chaos_strength,
                            prefix="chaos_attention.")
    h = gelu(linear(layernorm(x, w["norm2.weight"], w["norm2.bias"]), w["ffn.0.weight"], w["ffn.0.bias"]))
    return x + linear(h, w["ffn.3.weight"], w["ffn.3.bias"])
