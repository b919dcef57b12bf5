"""ns_nd.py -- TEST INFRASTRUCTURE ONLY: the executable form of SPEC_3D.md.

A numpy restatement of the reference stepper (/root/reference/src/physics/navier_stokes.py:24-173) written ONCE for any number of grid
axes: every rule is stated per axis / per velocity component, in the reference's operation order, with one fp32 rounding per reference
elementwise op (numpy array ops round once each and never contract a*b+c).  Run with a 2-tuple grid it IS the reference's algorithm --
tests/test_oracle_golden.py holds it bit-exact to the fixtures the reference itself produced (stage states, 50/100-step trajectories,
int64 back-trace indices) -- and run with a 3-tuple it is the definition of BASELINE configs[4]'s 3-D stepper, which has no reference
counterpart (every grid in the reference is a 2-tuple: navier_stokes.py:10,21).  So the 3-D oracle is pinned the only way it can be:
the same code path, one more axis, with the 2-D instance pinned by the reference.

Axis conventions.  A grid is indexed [y, x] (2-D) or [z, y, x] (3-D).  Velocity components (name, staggered axis, axis it acts along):
    2-D:  u (y, x)   v (x, y)                 -- the reference's own, crossed, pairing (navier_stokes.py:27-28, 97-109, 136, 148-149)
    3-D:  u (y, x)   v (x, y)   w (z, z)      -- u and v exactly as in 2-D on every depth slice; the new component is staggered along
                                                 depth and acts along depth
"Acts along a" means: sampled at the cell's own index shifted by +0.5 along a (clamped to the component's extent), it displaces the
back-trace along a.  Term orders (fp32 sums are order-sensitive) always list the 2-D axes first and depth last.
"""
import numpy as np

F32 = np.float32


def _sl(ndim, axis, s):
    idx = [slice(None)] * ndim
    idx[axis] = s
    return tuple(idx)


class OracleNSnd:
    """Un-batched stepper on an N-D grid (N = 2: the reference; N = 3: SPEC_3D.md).  Fields are numpy float32 arrays."""

    def __init__(self, grid_size, dt=0.01, viscosity=0.001, jacobi_iters=20):
        self.shape = tuple(int(s) for s in grid_size)
        self.nd = len(self.shape)
        assert self.nd in (2, 3)
        self.dt, self.viscosity, self.jacobi_iters = dt, viscosity, jacobi_iters
        if self.nd == 2:
            self.comps = [("u", 0, 1), ("v", 1, 0)]                # (name, staggered axis, acts along axis)
            self.axis_order = [0, 1]                               # neighbour order of the Laplacian / Jacobi sums: up, down, left, right
            self.jacobi_scale = 0.25                               # navier_stokes.py:141
        else:
            self.comps = [("u", 1, 2), ("v", 2, 1), ("w", 0, 0)]
            self.axis_order = [1, 2, 0]                            # up, down, left, right, front, back
            self.jacobi_scale = 1.0 / 6.0                          # a Python double, cast to fp32 when it meets the tensor (like 0.25)
        self.buoyant = "v"                                         # navier_stokes.py:154-155
        self.setup_grid()

    # ---- state (navier_stokes.py:24-35) ----
    def comp_shape(self, stag):
        return tuple(s + (1 if a == stag else 0) for a, s in enumerate(self.shape))

    def setup_grid(self):
        for name, stag, _ in self.comps:
            setattr(self, name, np.zeros(self.comp_shape(stag), F32))
        self.p = np.zeros(self.shape, F32)
        self.density = np.zeros(self.shape, F32)

    def velocity(self):
        return [getattr(self, n) for n, _, _ in self.comps]

    # ---- navier_stokes.py:37-48 ----
    def add_smoke_source(self, *center, radius=10, intensity=1.0):
        """center = (x, y) in 2-D (the reference's argument order: column, row), (x, y, z) in 3-D."""
        assert len(center) == self.nd
        grids = np.meshgrid(*[np.arange(s, dtype=np.int64) for s in self.shape], indexing="ij")
        d2 = np.zeros(self.shape, np.int64)
        for k, c in enumerate(center):                              # x pairs with the LAST axis
            d2 = d2 + (grids[self.nd - 1 - k] - int(c)) ** 2
        dist = np.sqrt(d2.astype(F32))                              # torch.sqrt of an int64 tensor is float32
        mask = dist <= radius
        dm = dist[mask]
        denom = 2 * (radius / 3) ** 2                               # Python double
        self.density[mask] += F32(intensity) * np.exp(-(dm * dm) / F32(denom))

    # ---- navier_stokes.py:50-72 ----
    def diffusion_step(self, field, viscosity):
        f = np.asarray(field, F32)
        pad = np.pad(f, 1, mode="edge")                             # edges and corners replicated (:57-66)
        nd = f.ndim
        core = tuple(slice(1, -1) for _ in range(nd))
        lap = None
        for a in self.axis_order:
            lo = list(core); lo[a] = slice(None, -2)
            hi = list(core); hi[a] = slice(2, None)
            lap = pad[tuple(lo)] if lap is None else lap + pad[tuple(lo)]
            lap = lap + pad[tuple(hi)]
        lap = lap - F32(2 * nd) * f                                 # "- 4 * field" (:70)
        return f + F32(self.dt * viscosity) * lap                   # Python doubles multiply first (:72)

    # ---- navier_stokes.py:111-131, generalised: floor -> clamp indices -> weights from the CLAMPED indices ----
    def interpolate(self, field, coords, want_indices=False):
        f = np.asarray(field, F32)
        nd = f.ndim
        lo, hi, wlo, whi = [], [], [], []
        for a in range(nd):
            c = coords[a]
            i0 = np.floor(c).astype(np.int64)
            i1 = i0 + 1
            i0 = np.clip(i0, 0, f.shape[a] - 1)
            i1 = np.clip(i1, 0, f.shape[a] - 1)
            lo.append(i0); hi.append(i1)
            wlo.append(i1.astype(F32) - c)                          # weight of the LOW tap: (x1 - x)
            whi.append(c - i0.astype(F32))                          # weight of the HIGH tap: (x - x0)
        out = None
        for tap in range(1 << nd):                                  # first axis slowest, last axis fastest, low before high (:130-131)
            bits = [(tap >> (nd - 1 - a)) & 1 for a in range(nd)]
            wgt = None
            for a in reversed(range(nd)):                           # (x-factor * y-factor) * z-factor (:125-128)
                fac = whi[a] if bits[a] else wlo[a]
                wgt = fac if wgt is None else wgt * fac
            idx = tuple(hi[a] if bits[a] else lo[a] for a in range(nd))
            term = wgt * f[idx]
            out = term if out is None else out + term
        return (out, lo) if want_indices else out

    def interpolate_velocity(self, k, coords, comp=None):
        """Component k (default: the state's own) sampled at `coords` shifted by +0.5 along the axis it acts on (:97-109)."""
        name, _, act = self.comps[k]
        return self._interp_comp(getattr(self, name) if comp is None else np.asarray(comp, F32), act, coords)

    def _interp_comp(self, comp, act, coords):
        """(:99-101, :106-108) the shifted coordinate is clamped to the COMPONENT's extent along that axis.  The other coordinates are
        clamped to the component's extent too: in 2-D that is a no-op -- with the reference's crossed pairing a field's index never
        leaves the sampled component's extent on the un-shifted axis -- but in 3-D it is not (the planes of w number D + 1, those of u
        and v only D; likewise rows of u vs w, columns of v vs w), and an index past the extent would give the clamped taps weights -1
        and +1.  Clamped, it sits exactly on the upper edge and the sample is 0: the quirk's own answer (SPEC_3D.md section 5)."""
        c = list(coords)
        for a in range(len(c)):
            if a == act:
                c[a] = np.clip(c[a] + F32(0.5), F32(0), F32(comp.shape[a] - 1))
            else:
                c[a] = np.clip(c[a], F32(0), F32(comp.shape[a] - 1))
        return self.interpolate(comp, c)

    # ---- navier_stokes.py:74-95 ----
    def advection_step(self, field, vel, want_indices=False):
        f = np.asarray(field, F32)
        coords = np.meshgrid(*[np.arange(s, dtype=F32) for s in f.shape], indexing="ij")
        prev = list(coords)
        for (name, _, act), comp in zip(self.comps, vel):
            vi = self._interp_comp(np.asarray(comp, F32), act, coords)
            prev[act] = coords[act] - F32(self.dt) * vi
        for a in range(f.ndim):
            prev[a] = np.clip(prev[a], F32(0), F32(f.shape[a] - 1))
        return self.interpolate(f, prev, want_indices)

    # ---- navier_stokes.py:133-149 ----
    def divergence(self):
        div = None
        for name, stag, _ in self.comps:
            c = getattr(self, name)
            hi = c[_sl(self.nd, stag, slice(1, None))]
            lo = c[_sl(self.nd, stag, slice(None, -1))]
            div = (hi - lo) if div is None else (div + hi) - lo     # u[1:]-u[:-1] + v[:,1:] - v[:,:-1] (+ w[1:] - w[:-1])
        return div / F32(self.dt)

    def jacobi(self, div, iters):
        nd = self.nd
        core = tuple(slice(1, -1) for _ in range(nd))
        for _ in range(iters):
            p_new = np.zeros_like(self.p)
            s = None
            for a in self.axis_order:
                lo = list(core); lo[a] = slice(None, -2)
                hi = list(core); hi[a] = slice(2, None)
                s = self.p[tuple(lo)] if s is None else s + self.p[tuple(lo)]
                s = s + self.p[tuple(hi)]
            p_new[core] = F32(self.jacobi_scale) * (s - div[core])
            self.p = p_new

    def grad_subtract(self):
        for name, stag, _ in self.comps:
            c = getattr(self, name)
            g = self.p[_sl(self.nd, stag, slice(1, None))] - self.p[_sl(self.nd, stag, slice(None, -1))]
            c[_sl(self.nd, stag, slice(1, -1))] -= F32(self.dt) * g

    def pressure_projection(self, iters=None):
        div = self.divergence()
        self.jacobi(div, self.jacobi_iters if iters is None else iters)
        self.grad_subtract()
        return div

    # ---- navier_stokes.py:151-173 ----
    def buoyancy(self):
        name, stag, _ = [c for c in self.comps if c[0] == self.buoyant][0]
        b = self.density * F32(0.1)
        getattr(self, name)[_sl(self.nd, stag, slice(None, -1))] += F32(self.dt) * b

    def diffuse_all(self):
        for name, _, _ in self.comps:
            setattr(self, name, self.diffusion_step(getattr(self, name), self.viscosity))
        self.density = self.diffusion_step(self.density, self.viscosity * 0.1)

    def advect_all(self):
        for name, _, _ in self.comps:                               # sequentially dependent: each uses the components advected before it
            setattr(self, name, self.advection_step(getattr(self, name), self.velocity()))
        self.density = self.advection_step(self.density, self.velocity())

    def step(self):
        self.buoyancy()
        self.diffuse_all()
        self.pressure_projection()
        self.advect_all()
        self.density = self.density * F32(0.995)
        return self.density.copy()
