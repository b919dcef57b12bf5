// Stencil kernels of the batched stable-fluids step (gfx950).
//
// Arithmetic contract: every kernel evaluates the reference's fp32 expression tree with ONE rounding per
// reference elementwise op (file is compiled with -ffp-contract=off; divides/sqrt are the correctly rounded
// forms), so velocity/pressure/density grids are bit-identical to the torch-CPU reference for identical inputs.
// Reference: /root/reference/src/physics/navier_stokes.py (lines cited per kernel).
#include <map>
#include <mutex>

#include "stencil.h"

#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace smk {

#define TX 64
#define TY 4

__device__ __forceinline__ float clampf(float x, float lo, float hi) {
    float t = x < lo ? lo : x;   // torch.clamp = min(max(x, lo), hi)
    return t > hi ? hi : t;
}
__device__ __forceinline__ int clampi(int x, int lo, int hi) {
    int t = x < lo ? lo : x;
    return t > hi ? hi : t;
}

// ---------------------------------------------------------------- reset (navier_stokes.py:24-35)
__global__ void k_zero_state(Geom g, StateView s, const uint8_t *mask) {
    int b = blockIdx.z;
    if (mask && !mask[b]) return;
    int j = blockIdx.x * TX + threadIdx.x, i = blockIdx.y * TY + threadIdx.y;
    if (i <= g.H && j < g.pc) s.u[b * g.su + (size_t)i * g.pc + j] = 0.f;
    if (i < g.H && j < g.pv) s.v[b * g.sv + (size_t)i * g.pv + j] = 0.f;
    if (i < g.H && j < g.pc) {
        s.p[b * g.sc + (size_t)i * g.pc + j] = 0.f;
        s.d[b * g.sc + (size_t)i * g.pc + j] = 0.f;
    }
}

hipError_t launch_zero_state(const Geom &g, StateView s, const uint8_t *dev_mask, hipStream_t st) {
    int maxp = g.pc > g.pv ? g.pc : g.pv;
    dim3 grid(cdiv(maxp, TX), cdiv(g.H + 1, TY), g.B), block(TX, TY);
    hipLaunchKernelGGL(k_zero_state, grid, block, 0, st, g, s, dev_mask);
    return hipGetLastError();
}

// ---------------------------------------------------------------- sources (navier_stokes.py:37-48)
// dist = sqrt(float(dx^2+dy^2)); mask dist<=radius; density += fint * exp(-(dist*dist)/denom)
__global__ void k_add_sources(Geom g, float *density, const SrcDev *src, const int *first) {
    int b = blockIdx.z;
    int s0 = first[b], s1 = first[b + 1];
    if (s0 == s1) return;
    int j = blockIdx.x * TX + threadIdx.x, i = blockIdx.y * TY + threadIdx.y;
    if (i >= g.H || j >= g.W) return;
    float *cell = density + b * g.sc + (size_t)i * g.pc + j;
    float d = *cell;
    for (int s = s0; s < s1; ++s) {
        SrcDev q = src[s];
        int dx = j - q.x, dy = i - q.y;
        float dist = __fsqrt_rn((float)(dx * dx + dy * dy));
        if (dist <= (float)q.radius) {
            float e = expf(-__fdiv_rn(dist * dist, q.denom));
            d = d + q.fint * e;
        }
    }
    *cell = d;
}

hipError_t launch_add_sources(const Geom &g, float *density, const SrcDev *dev_src, const int *dev_first, hipStream_t st) {
    dim3 grid(cdiv(g.W, TX), cdiv(g.H, TY), g.B), block(TX, TY);
    hipLaunchKernelGGL(k_add_sources, grid, block, 0, st, g, density, dev_src, dev_first);
    return hipGetLastError();
}

// ---------------------------------------------------------------- diffusion (navier_stokes.py:50-72)
// out = c + coef*((((up+down)+left)+right) - 4*c), replicate padding.
__device__ __forceinline__ float diffuse_at(const float *f, int R, int C, int pitch, int i, int j, float coef) {
    int iu = i > 0 ? i - 1 : 0, id = i < R - 1 ? i + 1 : R - 1;
    int jl = j > 0 ? j - 1 : 0, jr = j < C - 1 ? j + 1 : C - 1;
    float c = f[(size_t)i * pitch + j];
    float lap = f[(size_t)iu * pitch + j] + f[(size_t)id * pitch + j];
    lap = lap + f[(size_t)i * pitch + jl];
    lap = lap + f[(size_t)i * pitch + jr];
    lap = lap - 4.0f * c;
    return c + coef * lap;
}

__global__ void k_diffuse(const float *in, float *out, int R, int C, int pitch, size_t stride, float coef) {
    int j = blockIdx.x * TX + threadIdx.x, i = blockIdx.y * TY + threadIdx.y;
    if (i >= R || j >= C) return;
    const float *f = in + blockIdx.z * stride;
    out[blockIdx.z * stride + (size_t)i * pitch + j] = diffuse_at(f, R, C, pitch, i, j, coef);
}

hipError_t launch_diffuse(const float *in, float *out, int B, int R, int C, int pitch, float coef, hipStream_t st) {
    dim3 grid(cdiv(C, TX), cdiv(R, TY), B), block(TX, TY);
    hipLaunchKernelGGL(k_diffuse, grid, block, 0, st, in, out, R, C, pitch, (size_t)R * pitch, coef);
    return hipGetLastError();
}

static bool knobs_scalar_diffuse() {                         // SMK_DIFFUSE_SCALAR=1: the one-cell-per-thread form (diagnostic)
    static const bool v = [] { const char *e = getenv("SMK_DIFFUSE_SCALAR"); return e && e[0] == '1'; }();
    return v;
}

// buoyancy (navier_stokes.py:154-155) fused into the three diffusions (:158-160).
// v_b(i,j) = j < W ? v + dt*(density*0.1) : v    -- the buoyancy-updated v that diffusion_step(v) sees.
__device__ __forceinline__ float vbuoy(const float *v, const float *d, const Geom &g, int i, int j) {
    float x = v[(size_t)i * g.pv + j];
    if (j < g.W) {
        float b = d[(size_t)i * g.pc + j] * 0.1f;
        x = x + g.dt * b;
    }
    return x;
}

__global__ void k_buoy_diffuse(Geom g, StateView in, StateView out) {
    int b = blockIdx.z;
    int j = blockIdx.x * TX + threadIdx.x, i = blockIdx.y * TY + threadIdx.y;
    const float *u = in.u + b * g.su, *v = in.v + b * g.sv, *d = in.d + b * g.sc;
    if (i <= g.H && j < g.W)
        out.u[b * g.su + (size_t)i * g.pc + j] = diffuse_at(u, g.H + 1, g.W, g.pc, i, j, g.coef_uv);
    if (i < g.H && j < g.W)
        out.d[b * g.sc + (size_t)i * g.pc + j] = diffuse_at(d, g.H, g.W, g.pc, i, j, g.coef_d);
    if (i < g.H && j <= g.W) {
        int iu = i > 0 ? i - 1 : 0, id = i < g.H - 1 ? i + 1 : g.H - 1;
        int jl = j > 0 ? j - 1 : 0, jr = j < g.W ? j + 1 : g.W;
        float c = vbuoy(v, d, g, i, j);
        float lap = vbuoy(v, d, g, iu, j) + vbuoy(v, d, g, id, j);
        lap = lap + vbuoy(v, d, g, i, jl);
        lap = lap + vbuoy(v, d, g, i, jr);
        lap = lap - 4.0f * c;
        out.v[b * g.sv + (size_t)i * g.pv + j] = c + g.coef_uv * lap;
    }
}

// The same stage with four consecutive columns per thread (W % 4 == 0): the centre / up / down rows of u, v, density arrive as nine
// 16-byte loads plus the row's two outer neighbours instead of ~20 dword loads per cell (k_buoy_diffuse moved 173 MB for 117 MB of
// fields and spent 78 % of its wave cycles waiting on them).  Per cell the expression tree is diffuse_at's / vbuoy's.
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, const float (&v)[4]) { *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void diffuse4(const float4 &cc, const float4 &uu, const float4 &dd, float left, float right, float coef,
                                         float (&out)[4]) {
    const float c[4] = {cc.x, cc.y, cc.z, cc.w}, up[4] = {uu.x, uu.y, uu.z, uu.w}, dn[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float lap = up[k] + dn[k];
        lap = lap + (k > 0 ? c[k - 1] : left);
        lap = lap + (k < 3 ? c[k + 1] : right);
        lap = lap - 4.0f * c[k];
        out[k] = c[k] + coef * lap;
    }
}
__device__ __forceinline__ float4 buoy4(const float4 &v, const float4 &d, float dt) {   // v + dt * (d * 0.1): two roundings per product chain
    float4 r;
    float b;
    b = d.x * 0.1f; r.x = v.x + dt * b;
    b = d.y * 0.1f; r.y = v.y + dt * b;
    b = d.z * 0.1f; r.z = v.z + dt * b;
    b = d.w * 0.1f; r.w = v.w + dt * b;
    return r;
}

__global__ __launch_bounds__(256) void k_buoy_diffuse4(Geom g, StateView in, StateView out) {
    const int b = blockIdx.z, jq = blockIdx.x * TX + threadIdx.x, i = blockIdx.y * TY + threadIdx.y, j0 = 4 * jq;
    if (j0 >= g.W || i > g.H) return;
    const float *u = in.u + b * g.su, *v = in.v + b * g.sv, *d = in.d + b * g.sc;
    const int jl = j0 > 0 ? j0 - 1 : 0, jr = j0 + 4 < g.W ? j0 + 4 : g.W - 1;
    {   // u: H+1 rows
        const int iu = i > 0 ? i - 1 : 0, id = i < g.H ? i + 1 : g.H;
        const float *row = u + (size_t)i * g.pc;
        float o[4];
        diffuse4(ld4(row + j0), ld4(u + (size_t)iu * g.pc + j0), ld4(u + (size_t)id * g.pc + j0), row[jl], row[jr], g.coef_uv, o);
        st4(out.u + b * g.su + (size_t)i * g.pc + j0, o);
    }
    if (i >= g.H) return;
    const int iu = i > 0 ? i - 1 : 0, id = i < g.H - 1 ? i + 1 : g.H - 1;
    const float *drow = d + (size_t)i * g.pc, *vrow = v + (size_t)i * g.pv;
    const float4 dc = ld4(drow + j0), du = ld4(d + (size_t)iu * g.pc + j0), dd = ld4(d + (size_t)id * g.pc + j0);
    const float dl = drow[jl], dr = drow[jr];
    {   // density
        float o[4];
        diffuse4(dc, du, dd, dl, dr, g.coef_d, o);
        st4(out.d + b * g.sc + (size_t)i * g.pc + j0, o);
    }
    {   // v with the buoyancy of this step folded in (columns < W receive it; column W does not)
        const float4 vc = buoy4(ld4(vrow + j0), dc, g.dt), vu = buoy4(ld4(v + (size_t)iu * g.pv + j0), du, g.dt),
                     vd = buoy4(ld4(v + (size_t)id * g.pv + j0), dd, g.dt);
        float left = vc.x, right;
        if (j0 > 0) {
            const float bb = dl * 0.1f;
            left = vrow[j0 - 1] + g.dt * bb;
        }
        if (j0 + 4 < g.W) {
            const float bb = drow[j0 + 4] * 0.1f;
            right = vrow[j0 + 4] + g.dt * bb;
        } else {
            right = vrow[g.W];                                // v_b(i, W) = v(i, W)
        }
        float o[4];
        diffuse4(vc, vu, vd, left, right, g.coef_uv, o);
        st4(out.v + b * g.sv + (size_t)i * g.pv + j0, o);
        if (j0 + 4 == g.W) {                                  // the field's last column j = W: right neighbour = itself
            const float c = vrow[g.W];
            float lap = v[(size_t)iu * g.pv + g.W] + v[(size_t)id * g.pv + g.W];
            lap = lap + vc.w;
            lap = lap + c;
            lap = lap - 4.0f * c;
            out.v[b * g.sv + (size_t)i * g.pv + g.W] = c + g.coef_uv * lap;
        }
    }
}

hipError_t launch_buoy_diffuse(const Geom &g, StateView in, StateView out, hipStream_t st) {
    if (g.W % 4 == 0 && g.pc % 4 == 0 && g.pv % 4 == 0 && !knobs_scalar_diffuse()) {
        dim3 grid(cdiv(g.W / 4, TX), cdiv(g.H + 1, TY), g.B), block(TX, TY);
        hipLaunchKernelGGL(k_buoy_diffuse4, grid, block, 0, st, g, in, out);
        return hipGetLastError();
    }
    dim3 grid(cdiv(g.W + 1, TX), cdiv(g.H + 1, TY), g.B), block(TX, TY);
    hipLaunchKernelGGL(k_buoy_diffuse, grid, block, 0, st, g, in, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------- projection (navier_stokes.py:133-149)
// div = (((u[i+1,j]-u[i,j]) + v[i,j+1]) - v[i,j]) / dt       (true fp32 divide, as torch CPU)
__global__ void k_divergence(Geom g, const float *u, const float *v, float *div, int dp, size_t ds) {
    int b = blockIdx.z;
    int j = blockIdx.x * TX + threadIdx.x, i = blockIdx.y * TY + threadIdx.y;
    if (i >= g.H || j >= g.W) return;
    const float *ub = u + b * g.su, *vb = v + b * g.sv;
    float a = ub[(size_t)(i + 1) * g.pc + j] - ub[(size_t)i * g.pc + j];
    a = a + vb[(size_t)i * g.pv + j + 1];
    a = a - vb[(size_t)i * g.pv + j];
    div[b * ds + (size_t)i * dp + j] = __fdiv_rn(a, g.dt);
}

hipError_t launch_divergence(const Geom &g, const float *u, const float *v, float *div, int dp, size_t ds,
                             hipStream_t st) {
    dim3 grid(cdiv(g.W, TX), cdiv(g.H, TY), g.B), block(TX, TY);
    hipLaunchKernelGGL(k_divergence, grid, block, 0, st, g, u, v, div, dp, ds);
    return hipGetLastError();
}

// One Jacobi sweep: ring = 0, interior 0.25*((((up+down)+left)+right) - div)   (:140-145)
__global__ void k_jacobi_sweep(Geom g, const float *p, float *pn, const float *div) {
    int b = blockIdx.z;
    int j = blockIdx.x * TX + threadIdx.x, i = blockIdx.y * TY + threadIdx.y;
    if (i >= g.H || j >= g.W) return;
    const float *pb = p + b * g.sc;
    size_t o = (size_t)i * g.pc + j;
    float r = 0.f;
    if (i > 0 && i < g.H - 1 && j > 0 && j < g.W - 1) {
        float s = pb[o - g.pc] + pb[o + g.pc];
        s = s + pb[o - 1];
        s = s + pb[o + 1];
        s = s - div[b * g.sc + o];
        r = 0.25f * s;
    }
    pn[b * g.sc + o] = r;
}

__global__ void k_copy_cells(Geom g, const float *src, float *dst) {
    int j = blockIdx.x * TX + threadIdx.x, i = blockIdx.y * TY + threadIdx.y;
    if (i >= g.H || j >= g.W) return;
    size_t o = blockIdx.z * g.sc + (size_t)i * g.pc + j;
    dst[o] = src[o];
}

// ---- register-resident, temporally blocked Jacobi (one wave = one grid row, VEC = W/64 cells per lane) -------------
// A workgroup of 16 waves holds a band of TR = 16*RPW full rows of p and div in registers and runs `iters` sweeps
// without touching HBM: horizontal neighbours come from the adjacent lanes (DPP wave shifts), vertical neighbours
// across waves through a double-buffered LDS edge exchange (one barrier per sweep).  Each band carries `halo`
// redundant rows on every non-physical side, so `iters` <= that many sweeps are exact on the rows it owns
// (the garbage front moves one row per sweep).  Per cell the arithmetic is exactly k_jacobi_sweep's.
// bound_ctrl zero-fills the lane without a source (lane 0 / lane 63: grid-edge cells, whose neighbour value is never used), so no
// copy of `x` into the destination is needed before the DPP move: one instruction per shift instead of two.
__device__ __forceinline__ float wave_shr1(float x) {   // lane i <- lane i-1 (lane 0: 0, unused)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_shl1(float x) {   // lane i <- lane i+1 (lane 63: 0, unused)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x130, 0xf, 0xf, true));
}

constexpr int JB_NW = 16;

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<8> { using type = float4; };
// VEC consecutive floats of a lane as vector loads / stores (rows are 128-byte aligned and a lane's cells start at a multiple of VEC)
template <int VEC>
__device__ __forceinline__ void ldv(float (&dst)[VEC], const float *src) {
    using V = typename VecT<VEC>::type;
    constexpr int N = sizeof(V) / 4;
#pragma unroll
    for (int q = 0; q < VEC / N; ++q) {
        const V t = reinterpret_cast<const V *>(src)[q];
        const float *f = reinterpret_cast<const float *>(&t);
#pragma unroll
        for (int c = 0; c < N; ++c) dst[q * N + c] = f[c];
    }
}
template <int VEC>
__device__ __forceinline__ void stv(float *dst, const float (&src)[VEC]) {
    using V = typename VecT<VEC>::type;
    constexpr int N = sizeof(V) / 4;
#pragma unroll
    for (int q = 0; q < VEC / N; ++q) {
        V t;
        float *f = reinterpret_cast<float *>(&t);
#pragma unroll
        for (int c = 0; c < N; ++c) f[c] = src[q * N + c];
        reinterpret_cast<V *>(dst)[q] = t;
    }
}

// MODE bit 0 (FIRST launch of a projection): the divergence of (u, v) is computed here for all tile rows (navier_stokes.py:136,
//   same expression tree as k_divergence) and stored for the owned rows so that later launches can read it;
// MODE bit 1 (LAST launch): after the final sweep the gradient subtraction (navier_stokes.py:148-149, as k_grad_subtract)
//   is applied to the owned rows of u and v from the p held in registers (needs iters <= halo - 1: the row above the
//   owned range must still be exact).
// PERSIST (with MODE 3): the whole projection in ONE launch.  p and div stay in registers for all `iters` sweeps; after each chunk of
//   at most `halo` sweeps a band hands the `halo` owned rows next to each inner boundary to its neighbour band through HBM (exchange
//   buffers x0 / x1 alternate, so a band never overwrites rows its neighbour may still be reading) and refreshes its own halo rows
//   from the neighbours'.  Hand-off as MI355X_MICROARCH.md "inter-workgroup visibility" prescribes for one workgroup per CU: every
//   payload store and load is a 16-byte `sc1` access, every storing wave drains its stores (s_waitcnt vmcnt(0)) before the workgroup
//   barrier, one lane then stores the band's flag `sc1`, the consumer's lane 0 polls the two neighbour flags with `sc1` loads and joins
//   a workgroup barrier before any wave loads.  Every spin is bounded (JacobiSync::timeout_ticks of the 100 MHz wall clock): a band
//   whose neighbour never arrives sets *status and the abort word and runs on, so the grid always drains; the host reports it on the
//   next call.  All bands x grids of one launch must be co-resident (the launcher sizes the grid to the CU count).
struct JacobiSync {
    unsigned *flags;          // [B * nb + 1]: per band the number of hand-offs it has published (monotonic over the handle's life); last = abort
    unsigned *status;         // host-visible word, set non-zero when a wait timed out
    float *x0, *x1;           // exchange buffers, laid out like p (div and p2 of the handle)
    unsigned base;            // flag value before this projection's first hand-off
    int chunks, nb, grid0, ngrids, abort_slot, fault;
    long long timeout_ticks;
    // FOLD: this step's buoyancy + diffusion stage runs as the launch's prologue: (u_in, v_in, d_in) -> the kernel's u, v (= u2, v2) and d_out
    const float *u_in, *v_in, *d_in;
    float *d_out;
};

template <int VEC>
__device__ __forceinline__ void ldv_sc1(float (&dst)[VEC], __amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    if constexpr (VEC >= 4) {
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off + 16 * q, 0, 16);
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[4 * q + c] = __uint_as_float(t[c]);
        }
    } else if constexpr (VEC == 2) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 16);
        dst[0] = __uint_as_float(t[0]);
        dst[1] = __uint_as_float(t[1]);
    } else {
        dst[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 16));
    }
}
template <int VEC>
__device__ __forceinline__ void stv_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, const float (&src)[VEC]) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    if constexpr (VEC >= 4) {
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            u32x4 t;
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = __float_as_uint(src[4 * q + c]);
            __builtin_amdgcn_raw_buffer_store_b128(t, rs, byte_off + 16 * q, 0, 16);
        }
    } else if constexpr (VEC == 2) {
        u32x2 t;
        t[0] = __float_as_uint(src[0]);
        t[1] = __float_as_uint(src[1]);
        __builtin_amdgcn_raw_buffer_store_b64(t, rs, byte_off, 0, 16);
    } else {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(src[0]), rs, byte_off, 0, 16);
    }
}

template <int VEC, int RPW, int MODE, bool PERSIST = false, bool FOLD = false>
__global__ __launch_bounds__(JB_NW * 64) void k_jacobi_band(Geom g, const float *__restrict__ p_in,
                                                            float *__restrict__ p_out, float *__restrict__ div,
                                                            float *__restrict__ u, float *__restrict__ v,
                                                            int iters, int BR, JacobiSync sy) {
    constexpr int TR = JB_NW * RPW, ROWF = 64 * VEC;
    __shared__ float edge[2][JB_NW][2][ROWF];
    __shared__ int handoff_failed;                            // PERSIST: lane 0 saw a timed-out / aborted hand-off wait -> poison the band's results
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Bands own unequal row ranges: the first and last band of a grid need a halo only on their inner side (the other side is the
    // physical boundary), so they own TR - HALO rows and the middle bands TR - 2 HALO (HALO = BR here).  One band: the whole grid.
    int band_, grid_, nb_;
    if constexpr (PERSIST) {
        // 1-D launch.  Blocks i and i + 8 share an XCD (round-robin dispatch: observed, not promised -- only the hand-off's speed depends
        // on it): when the grids divide by 8, the bands of one grid take consecutive slots of one XCD.
        nb_ = sy.nb;
        const int id = blockIdx.x;
        if ((sy.ngrids & 7) == 0) {
            const int xcd = id & 7, slot = id >> 3;
            grid_ = (slot / nb_) * 8 + xcd;
            band_ = slot % nb_;
        } else {
            grid_ = id / nb_;
            band_ = id % nb_;
        }
        grid_ += sy.grid0;
    } else {
        band_ = blockIdx.x; grid_ = blockIdx.y; nb_ = gridDim.x;
    }
    const int band = band_, b = grid_, nb = nb_, halo = BR;
    const int e_rows = TR - halo, m_rows = TR - 2 * halo;
    const int own0 = band == 0 ? 0 : e_rows + (band - 1) * m_rows;
    int own1 = band == nb - 1 ? g.H : e_rows + band * m_rows;
    own1 = own1 < g.H ? own1 : g.H;
    int r0 = band == 0 ? 0 : (band == nb - 1 ? g.H - TR : own0 - halo);
    if (r0 > g.H - TR) r0 = g.H - TR;
    if (r0 < 0) r0 = 0;
    const int row0 = r0 + wave * RPW, j0 = lane * VEC;
    const size_t base = b * g.sc + (size_t)row0 * g.pc + j0;
    float pv[RPW][VEC], dv[RPW][VEC];
    const bool first_col = lane == 0, last_col = lane == 63;
    if constexpr (FOLD) {
        // ---- buoyancy + the three diffusions (navier_stokes.py:154-160; per cell the expression trees of k_buoy_diffuse4) for the
        // tile's rows, each wave for itself from the step's input state: u2 rows row0 .. row0+RPW, v2 / d2 rows row0 .. row0+RPW-1.
        // Halo rows are computed redundantly by the neighbour bands (no band reads another band's u2 / v2); owned rows are stored
        // (the advection reads them) and the divergence is formed from the registers.  Rows roll through a 3-row window.
        const float *ui = sy.u_in + b * g.su, *vi = sy.v_in + b * g.sv, *di = sy.d_in + b * g.sc;
        float *uo = u + b * g.su, *vo = v + b * g.sv, *dout = sy.d_out + b * g.sc;
        const int H = g.H, W = g.W;
        auto rowu = [&](int i, float (&x)[VEC]) {
            i = i < 0 ? 0 : (i > H ? H : i);
            ldv<VEC>(x, ui + (size_t)i * g.pc + j0);
        };
        auto rowdv = [&](int i, float (&dx)[VEC], float (&vx)[VEC], float &vW) {       // d row; v row with this step's buoyancy; raw v(i, W)
            i = i < 0 ? 0 : (i > H - 1 ? H - 1 : i);
            ldv<VEC>(dx, di + (size_t)i * g.pc + j0);
            ldv<VEC>(vx, vi + (size_t)i * g.pv + j0);
            vW = vi[(size_t)i * g.pv + W];
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const float bb = dx[c] * 0.1f;
                vx[c] = vx[c] + g.dt * bb;
            }
        };
        auto lap = [&](const float (&cc)[VEC], const float (&up)[VEC], const float (&dn)[VEC], float left, float right, float coef,
                       float (&o)[VEC]) {
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                float l = up[c] + dn[c];
                l = l + (c > 0 ? cc[c - 1] : left);
                l = l + (c < VEC - 1 ? cc[c + 1] : right);
                l = l - 4.0f * cc[c];
                o[c] = cc[c] + coef * l;
            }
        };
        float um[VEC], uc[VEC], un[VEC], dm[VEC], dc[VEC], dn[VEC], vm[VEC], vc[VEC], vn[VEC], vWm, vWc, vWn;
        rowu(row0 - 1, um);
        rowu(row0, uc);
        rowdv(row0 - 1, dm, vm, vWm);
        rowdv(row0, dc, vc, vWc);
        float u2prev[VEC], v2keep[VEC + 1];
#pragma unroll
        for (int k = 0; k <= RPW; ++k) {
            const int gi = row0 + k;
            rowu(gi + 1, un);
            float u2[VEC];
            {
                const float sl = wave_shr1(uc[VEC - 1]), sr = wave_shl1(uc[0]);
                lap(uc, um, un, first_col ? uc[0] : sl, last_col ? uc[VEC - 1] : sr, g.coef_uv, u2);
            }
            if ((gi >= own0 && gi < own1) || (gi == H && band == nb - 1)) stv<VEC>(uo + (size_t)gi * g.pc + j0, u2);
            if (k > 0) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    float a = u2[c] - u2prev[c];
                    a = a + v2keep[c + 1];
                    a = a - v2keep[c];
                    dv[k - 1][c] = __fdiv_rn(a, g.dt);
                }
            }
#pragma unroll
            for (int c = 0; c < VEC; ++c) { u2prev[c] = u2[c]; um[c] = uc[c]; uc[c] = un[c]; }
            if (k < RPW) {
                rowdv(gi + 1, dn, vn, vWn);
                float d2[VEC], v2[VEC];
                {
                    const float sl = wave_shr1(dc[VEC - 1]), sr = wave_shl1(dc[0]);
                    lap(dc, dm, dn, first_col ? dc[0] : sl, last_col ? dc[VEC - 1] : sr, g.coef_d, d2);
                }
                {
                    const float sl = wave_shr1(vc[VEC - 1]), sr = wave_shl1(vc[0]);
                    lap(vc, vm, vn, first_col ? vc[0] : sl, last_col ? vWc : sr, g.coef_uv, v2);
                }
                // the field's last column j = W (no buoyancy; right neighbour = itself): meaningful in the last lane only
                float lw = vWm + vWn;
                lw = lw + vc[VEC - 1];
                lw = lw + vWc;
                lw = lw - 4.0f * vWc;
                const float v2W = vWc + g.coef_uv * lw;
                if (gi >= own0 && gi < own1) {
                    stv<VEC>(dout + (size_t)gi * g.pc + j0, d2);
                    stv<VEC>(vo + (size_t)gi * g.pv + j0, v2);
                    if (last_col) vo[(size_t)gi * g.pv + W] = v2W;
                }
                const float nx = wave_shl1(v2[0]);
#pragma unroll
                for (int c = 0; c < VEC; ++c) v2keep[c] = v2[c];
                v2keep[VEC] = last_col ? v2W : nx;
#pragma unroll
                for (int c = 0; c < VEC; ++c) { dm[c] = dc[c]; dc[c] = dn[c]; vm[c] = vc[c]; vc[c] = vn[c]; }
                vWm = vWc; vWc = vWn;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k)
#pragma unroll
        for (int c = 0; c < VEC; ++c) pv[k][c] = p_in[base + (size_t)k * g.pc + c];
    if constexpr (FOLD) {
        // (divergence already in dv)
    } else if (MODE & 1) {
        const float *ub = u + b * g.su + (size_t)row0 * g.pc + j0, *vb = v + b * g.sv + (size_t)row0 * g.pv + j0;
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            float vr[VEC + 1];
#pragma unroll
            for (int c = 0; c <= VEC; ++c) vr[c] = vb[(size_t)k * g.pv + c];
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                float a = ub[(size_t)(k + 1) * g.pc + c] - ub[(size_t)k * g.pc + c];
                a = a + vr[c + 1];
                a = a - vr[c];
                dv[k][c] = __fdiv_rn(a, g.dt);
            }
            const int gi = row0 + k;
            if (!PERSIST && gi >= own0 && gi < own1) {        // (a persistent launch keeps div in registers to the end)
#pragma unroll
                for (int c = 0; c < VEC; ++c) div[base + (size_t)k * g.pc + c] = dv[k][c];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < RPW; ++k)
#pragma unroll
            for (int c = 0; c < VEC; ++c) dv[k][c] = div[base + (size_t)k * g.pc + c];
    }
    // ring rows (grid row 0 / H-1) exist only in the first wave of the first band and the last wave of the last band
    const int ring_k = __builtin_amdgcn_readfirstlane(row0 == 0 ? 0 : (row0 + RPW == g.H ? RPW - 1 : -1));   // wave-uniform
    // one sweep src -> dst (register ping-pong: no row copies); par selects the LDS edge buffer
    auto sweep = [&](const float (&src)[RPW][VEC], float (&dst)[RPW][VEC], int par) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            edge[par][wave][0][j0 + c] = src[0][c];
            edge[par][wave][1][j0 + c] = src[RPW - 1][c];
        }
        __builtin_amdgcn_sched_barrier(0);                    // publish first: the neighbours' edge rows wait on these stores
        // one row of the sweep; up / dn: the rows above and below (registers, or the neighbour wave's edge row)
        auto row = [&](int k, const float (&up)[VEC], const float (&dn)[VEC]) {
            const float lin = wave_shr1(src[k][VEC - 1]), rin = wave_shl1(src[k][0]);
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const float l = c > 0 ? src[k][c - 1] : lin;
                const float r = c < VEC - 1 ? src[k][c + 1] : rin;
                float sm = up[c] + dn[c];
                sm = sm + l;
                sm = sm + r;
                sm = sm - dv[k][c];
                dst[k][c] = 0.25f * sm;
            }
            dst[k][0] = first_col ? 0.f : dst[k][0];          // column ring: only the two edge cells need a select
            dst[k][VEC - 1] = last_col ? 0.f : dst[k][VEC - 1];
        };
        // The sweep is bound by publish -> barrier -> read -> compute, not by VALU throughput: the rows that need nothing from
        // the neighbour waves (1 .. RPW-2) are computed BEFORE the barrier, under the wait; only the two edge rows follow it.
#pragma unroll
        for (int k = 1; k < RPW - 1; ++k) row(k, src[k - 1], src[k + 1]);
        __builtin_amdgcn_sched_barrier(0);                    // (hipcc otherwise sinks these rows below the barrier)
        __syncthreads();
        const float *eu = &edge[par][wave > 0 ? wave - 1 : 0][1][j0];            // top wave: value unused (ring or halo row)
        const float *ed = &edge[par][wave < JB_NW - 1 ? wave + 1 : JB_NW - 1][0][j0];
        float above[VEC], below[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            above[c] = eu[c];
            below[c] = ed[c];
        }
        if (RPW == 1) {
            row(0, above, below);
        } else {
            row(0, above, src[1]);
            row(RPW - 1, src[RPW - 2], below);
        }
        // row ring (grid row 0 / H-1: 2 waves of a grid): wave-uniform selects -- as scalar branches they cost more in register
        // copies at the control-flow merges (16 v_mov per sweep) than the 2 * VEC v_cndmask they save
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            dst[0][c] = ring_k == 0 ? 0.f : dst[0][c];
            dst[RPW - 1][c] = ring_k == RPW - 1 ? 0.f : dst[RPW - 1][c];
        }
    };
    float pw[RPW][VEC];
    auto run = [&](int n) {                                   // n sweeps, result in pv
        int it = 0;
        for (; it + 2 <= n; it += 2) {
            sweep(pv, pw, 0);
            sweep(pw, pv, 1);
        }
        if (it < n) {
            sweep(pv, pw, 0);
#pragma unroll
            for (int k = 0; k < RPW; ++k)
#pragma unroll
                for (int c = 0; c < VEC; ++c) pv[k][c] = pw[k][c];
        }
    };
    if constexpr (!PERSIST) {
        run(iters);
    } else {
        const int me = b * nb + band;
        const unsigned row_off = (unsigned)(base * sizeof(float));          // byte offset of this lane's cells of tile row row0 in x0 / x1
        const unsigned pitch_b = (unsigned)(g.pc * sizeof(float));
        const unsigned xbytes = (unsigned)((size_t)(sy.grid0 + sy.ngrids) * g.sc * sizeof(float));
        const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(sy.x0, 0, xbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(sy.x1, 0, xbytes, 0x00020000);
        int done = 0;
        if (threadIdx.x == 0) handoff_failed = 0;             // (read only after a hand-off: at least two workgroup barriers later)
        for (int c = 0; c < sy.chunks; ++c) {
            const int n = (iters - done + (sy.chunks - c) - 1) / (sy.chunks - c);
            run(n);
            done += n;
            if (c == sy.chunks - 1) break;
            const __amdgpu_buffer_rsrc_t rx = (c & 1) ? rx1 : rx0;
            // publish: the `halo` owned rows next to each inner boundary
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
                const int gi = row0 + k;
                const bool pub = (band > 0 && gi >= own0 && gi < own0 + halo) || (band < nb - 1 && gi >= own1 - halo && gi < own1);
                if (pub) stv_sc1<VEC>(rx, row_off + k * pitch_b, pv[k]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores before the barrier
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned tgt = sy.base + (unsigned)c + 1u;
                if (!(sy.fault && me == sy.grid0 * nb))       // (fault injection for the time-out test: one band never publishes)
                    __hip_atomic_store(sy.flags + me, tgt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const long long t0 = wall_clock64();
                for (;;) {
                    const bool up = band == 0 ||
                        (int)(__hip_atomic_load(sy.flags + me - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - tgt) >= 0;
                    const bool dn = band == nb - 1 ||
                        (int)(__hip_atomic_load(sy.flags + me + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - tgt) >= 0;
                    if (up && dn) break;
                    if (__hip_atomic_load(sy.flags + sy.abort_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                        handoff_failed = 1;                   // another band gave up: this band's halo rows are stale too
                        break;
                    }
                    if (wall_clock64() - t0 > sy.timeout_ticks) {
                        __hip_atomic_store(sy.flags + sy.abort_slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(sy.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        handoff_failed = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
            // refresh: the halo rows (the neighbours' published rows; rows of the tile beyond them stay stale, which `halo` sweeps
            // cannot carry into the owned range)
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
                const int gi = row0 + k;
                const bool need = (gi >= own0 - halo && gi < own0) || (gi >= own1 && gi < own1 + halo);
                if (need) ldv_sc1<VEC>(pv[k], rx, row_off + k * pitch_b);
            }
        }
        // A hand-off that did not complete leaves stale halo rows, and p, u, v are updated in place: results that cannot be right must not
        // look like data.  The band's p becomes NaN (so do its u, v through the gradient below, and every frame of the grid from here on);
        // the host reads *status at its next synchronising call (smk_sim_status) and reports the time-out for THIS projection.
        if (sy.chunks > 1 && handoff_failed) {
#pragma unroll
            for (int k = 0; k < RPW; ++k)
#pragma unroll
                for (int c = 0; c < VEC; ++c) pv[k][c] = __builtin_nanf("");
        }
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int gi = row0 + k;
        if (gi >= own0 && gi < own1) {
#pragma unroll
            for (int c = 0; c < VEC; ++c) p_out[base + (size_t)k * g.pc + c] = pv[k][c];
        }
    }
    if (MODE & 2) {
        // u[i,:] -= dt*(p[i,:] - p[i-1,:]) for 1 <= i <= H-1;  v[:,j] -= dt*(p[:,j] - p[:,j-1]) for 1 <= j <= W-1
        __syncthreads();                                      // all reads of the last sweep's edges are done
#pragma unroll
        for (int c = 0; c < VEC; ++c) edge[0][wave][1][j0 + c] = pv[RPW - 1][c];
        __syncthreads();
        float above[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) above[c] = edge[0][wave > 0 ? wave - 1 : 0][1][j0 + c];
        float *ub = u + b * g.su + (size_t)row0 * g.pc + j0, *vb = v + b * g.sv + (size_t)row0 * g.pv + j0;
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int gi = row0 + k;
            const float lin = wave_shr1(pv[k][VEC - 1]);
            if (gi >= own0 && gi < own1) {
                // whole-row read-modify-write (VEC cells per lane as one load / one store; the untouched cells -- row 0 of u, column 0
                // of v -- are written back unchanged)
                float un[VEC], vn[VEC];
                ldv<VEC>(un, ub + (size_t)k * g.pc);
                ldv<VEC>(vn, vb + (size_t)k * g.pv);
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    if (gi >= 1) {                            // gi == row0 == 0 only in the first wave of band 0: skipped
                        const float pu = k > 0 ? pv[k - 1][c] : above[c];
                        const float gr = pv[k][c] - pu;
                        un[c] = un[c] - g.dt * gr;
                    }
                    const float pl = c > 0 ? pv[k][c - 1] : lin;
                    const float gr = pv[k][c] - pl;
                    const float nv = vn[c] - g.dt * gr;
                    vn[c] = (c == 0 && first_col) ? vn[c] : nv;   // j >= 1 (j <= W-1 always holds here)
                }
                if (gi >= 1) stv<VEC>(ub + (size_t)k * g.pc, un);
                stv<VEC>(vb + (size_t)k * g.pv, vn);
            }
        }
    }
}

// Diagnostic switches of the projection, read ONCE per process (never on the per-step path):
//   SMK_JACOBI_GENERIC=1   one launch per sweep (k_jacobi_sweep)        SMK_PROJECT_UNFUSED=1  divergence / gradient as own launches
//   SMK_JACOBI_RPW, SMK_JACOBI_BANDS   pin the band plan of k_jacobi_band   SMK_STENCIL_DEBUG=1    print the chosen plan
//   SMK_JACOBI_PERSIST=0   one launch per chunk of sweeps instead of the single persistent launch
//   SMK_JACOBI_FAULT=1     (test only) one band never publishes its hand-off: exercises the bounded wait and the error report
//   SMK_PROJECT_FOLD=0     buoyancy + diffusion as their own launch in front of the persistent projection
struct StencilKnobs {
    bool generic, unfused, debug, persist, fault, fold;
    int rpw, bands;
    StencilKnobs() {
        auto flag = [](const char *n, bool dflt) { const char *v = getenv(n); return v ? v[0] == '1' : dflt; };
        auto num = [](const char *n) { const char *v = getenv(n); return v ? atoi(v) : 0; };
        generic = flag("SMK_JACOBI_GENERIC", false);
        unfused = flag("SMK_PROJECT_UNFUSED", false);
        debug = flag("SMK_STENCIL_DEBUG", false);
        persist = flag("SMK_JACOBI_PERSIST", true);
        fault = flag("SMK_JACOBI_FAULT", false);
        fold = flag("SMK_PROJECT_FOLD", true);
        rpw = num("SMK_JACOBI_RPW");
        bands = num("SMK_JACOBI_BANDS");
    }
};
static const StencilKnobs &knobs() {
    static const StencilKnobs k;
    return k;
}

// Band plan for the register-resident kernel; false -> use the generic per-sweep kernel.
struct JacobiPlan { int vec, rpw, br, nb, halo; };
// persist: cost the plan for the single-launch form (a hand-off between chunks instead of a relaunch)
static bool plan_jacobi(const Geom &g, JacobiPlan &pl, int iters = 100, bool persist = false) {
    if (g.W % 64 != 0 || g.W / 64 > 8 || (g.W / 64 & (g.W / 64 - 1)) || g.pc % 4 != 0) return false;
    if (knobs().generic) return false;
    pl.vec = g.W / 64;
    // candidates: rows/wave; pick the plan with the least estimated time ~ launches*(t0 + iters*waves_of_work)
    const int rpws[] = {2, 3, 4, 6, 8};
    double best = 1e30;
    bool ok = false;
    const int env_rpw = knobs().rpw, env_nb = knobs().bands;
    for (int rpw : rpws) {
        if (env_rpw && env_rpw != rpw) continue;
        if (pl.vec * rpw > 32) continue;                      // register budget (p + div)
        const int TR = JB_NW * rpw;
        if (TR > g.H) continue;
        for (int nb = 1; nb <= g.H / 8; ++nb) {
            if (env_nb && env_nb != nb) continue;
            // nb bands of TR rows cover H owned rows with a halo on every inner side: 2 (TR - h) + (nb - 2)(TR - 2h) >= H
            int halo = nb == 1 ? 1 << 20 : (nb * TR - g.H) / (2 * nb - 2);
            if (nb == 1 && TR != g.H) continue;
            if (nb * TR < g.H || halo < 2) continue;
            if (nb > 1) {
                if (halo > 64) halo = 64;
                if (halo > (TR - 1) / 2) halo = (TR - 1) / 2;          // middle bands keep at least one owned row
                if ((TR - halo) + (nb - 2) * (TR - 2 * halo) >= g.H) continue;   // a smaller halo than the cover needs: the last band would own nothing
            }
            if (persist && nb > 1) {                          // every band must own more rows than the halo (persist_chunks)
                const int e_rows = TR - halo, m_rows = TR - 2 * halo, last = g.H - (e_rows + (nb - 2) * m_rows);
                if (e_rows <= halo || last <= halo || (nb > 2 && m_rows <= halo) || nb > 64) continue;
            }
            const int br = halo;                              // handed to the kernel (it derives the owned ranges from it)
            const double wgs = (double)nb * g.B, rounds = ceil(wgs / 256.0);
            // measured on MI355X (256^2 x 64, profiles/r01): ~6 us fixed per launch, ~1.33 us per sweep at 96 rows per
            // workgroup (barrier + LDS round trip bound, roughly linear in the rows a CU owns)
            const double cost_per_sweep = rounds * TR * (1.33 / 96.0), launch = 6.0;
            // a relaunch reloads the band's p and div (~6 us with the boundary); a hand-off inside the persistent launch moves 2 * halo rows
            // per band through sc1 stores / flag / sc1 loads (~3.5 us)
            const double cost = persist ? launch + (ceil((double)iters / halo) - 1) * 3.5 + iters * cost_per_sweep
                                        : 2 * ceil(0.5 * iters / halo) * launch + iters * cost_per_sweep;
            if (cost < best) { best = cost; pl.rpw = rpw; pl.br = br; pl.nb = nb; pl.halo = halo; ok = true; }
        }
    }
    return ok;
}


template <int VEC, int MODE>
static void launch_band(const Geom &g, const JacobiPlan &pl, const float *pin, float *pout, float *div, float *u, float *v,
                        int iters, hipStream_t st) {
    dim3 grid(pl.nb, g.B), block(JB_NW * 64);
    const JacobiSync none{};
    switch (pl.rpw) {
        case 2: hipLaunchKernelGGL((k_jacobi_band<VEC, 2, MODE>), grid, block, 0, st, g, pin, pout, div, u, v, iters, pl.br, none); break;
        case 3: hipLaunchKernelGGL((k_jacobi_band<VEC, 3, MODE>), grid, block, 0, st, g, pin, pout, div, u, v, iters, pl.br, none); break;
        case 4: hipLaunchKernelGGL((k_jacobi_band<VEC, 4, MODE>), grid, block, 0, st, g, pin, pout, div, u, v, iters, pl.br, none); break;
        case 6: if constexpr (VEC <= 4) hipLaunchKernelGGL((k_jacobi_band<VEC, 6, MODE>), grid, block, 0, st, g, pin, pout, div, u, v, iters, pl.br, none); break;
        case 8: if constexpr (VEC <= 4) hipLaunchKernelGGL((k_jacobi_band<VEC, 8, MODE>), grid, block, 0, st, g, pin, pout, div, u, v, iters, pl.br, none); break;
    }
}

// the whole projection (FOLD: with the step's buoyancy + diffusion stage as its prologue) as one persistent launch per group of
// co-resident grids
template <int VEC, bool FOLD>
static void launch_persist(const Geom &g, const JacobiPlan &pl, float *p, float *u, float *v, int iters, const JacobiSync &sy, hipStream_t st) {
    dim3 grid(pl.nb * sy.ngrids), block(JB_NW * 64);
    switch (pl.rpw) {
        case 2: hipLaunchKernelGGL((k_jacobi_band<VEC, 2, 3, true, FOLD>), grid, block, 0, st, g, p, p, nullptr, u, v, iters, pl.br, sy); break;
        case 3: hipLaunchKernelGGL((k_jacobi_band<VEC, 3, 3, true, FOLD>), grid, block, 0, st, g, p, p, nullptr, u, v, iters, pl.br, sy); break;
        case 4: hipLaunchKernelGGL((k_jacobi_band<VEC, 4, 3, true, FOLD>), grid, block, 0, st, g, p, p, nullptr, u, v, iters, pl.br, sy); break;
        case 6: if constexpr (VEC <= 4) hipLaunchKernelGGL((k_jacobi_band<VEC, 6, 3, true, FOLD>), grid, block, 0, st, g, p, p, nullptr, u, v, iters, pl.br, sy); break;
        case 8: if constexpr (VEC <= 4) hipLaunchKernelGGL((k_jacobi_band<VEC, 8, 3, true, FOLD>), grid, block, 0, st, g, p, p, nullptr, u, v, iters, pl.br, sy); break;
    }
}
template <bool FOLD>
static void launch_persist_vec(const Geom &g, const JacobiPlan &pl, float *p, float *u, float *v, int iters, const JacobiSync &sy, hipStream_t st) {
    switch (pl.vec) {
        case 1: launch_persist<1, FOLD>(g, pl, p, u, v, iters, sy, st); break;
        case 2: launch_persist<2, FOLD>(g, pl, p, u, v, iters, sy, st); break;
        case 4: launch_persist<4, FOLD>(g, pl, p, u, v, iters, sy, st); break;
        case 8: launch_persist<8, false>(g, pl, p, u, v, iters, sy, st); break;       // (8 cells per lane: the prologue's row windows do not fit)
    }
}

// Can this plan run as one persistent launch, and in how many chunks?  Every band must own more rows than the halo (a band's halo rows
// then lie in its direct neighbour's owned range, and the u / v rows a neighbour's divergence reads are not rewritten before it has
// published once); chunk sizes are ceil / floor of iters / chunks: the largest <= halo, the last <= halo - 1 (the fused gradient needs
// the row above the owned range exact); two or more bands need two or more chunks (the first hand-off orders the final stores
// behind the neighbours' initial loads).
static bool persist_chunks(const Geom &g, const JacobiPlan &pl, int iters, int &chunks) {
    const int TR = JB_NW * pl.rpw;
    if (pl.nb == 1) { chunks = 1; return true; }
    if (pl.nb > 64 || (size_t)g.B * g.sc * sizeof(float) >= (1ull << 31)) return false;
    const int e_rows = TR - pl.halo, m_rows = TR - 2 * pl.halo, last = g.H - (e_rows + (pl.nb - 2) * m_rows);
    int min_owned = e_rows < last ? e_rows : last;
    if (pl.nb > 2 && m_rows < min_owned) min_owned = m_rows;
    if (min_owned < pl.halo + 1) return false;
    chunks = (iters + pl.halo - 1) / pl.halo;
    if (chunks < 2) chunks = 2;
    while ((iters + chunks - 1) / chunks > pl.halo || iters / chunks > pl.halo - 1) ++chunks;
    return chunks <= iters;
}

template <int MODE>
static void launch_band_vec(const Geom &g, const JacobiPlan &pl, const float *pin, float *pout, float *div, float *u, float *v,
                            int iters, hipStream_t st) {
    switch (pl.vec) {
        case 1: launch_band<1, MODE>(g, pl, pin, pout, div, u, v, iters, st); break;
        case 2: launch_band<2, MODE>(g, pl, pin, pout, div, u, v, iters, st); break;
        case 4: launch_band<4, MODE>(g, pl, pin, pout, div, u, v, iters, st); break;
        case 8: launch_band<8, MODE>(g, pl, pin, pout, div, u, v, iters, st); break;
    }
}

// `iters` Jacobi sweeps on a given divergence field (result in p; p2 scratch).
hipError_t launch_jacobi(const Geom &g, float *p, float *p2, const float *div, int iters, hipStream_t st) {
    if (iters <= 0) return hipSuccess;
    dim3 grid(cdiv(g.W, TX), cdiv(g.H, TY), g.B), block(TX, TY);
    JacobiPlan pl;
    float *cur = p, *nxt = p2;
    if (plan_jacobi(g, pl, iters)) {
        // an even number of nearly equal chunks (global ping-pong ends back in p); each chunk <= halo sweeps
        int L = 2 * ((iters + 2 * pl.halo - 1) / (2 * pl.halo));
        if (iters == 1) L = 1;
        int done = 0;
        for (int c = 0; c < L; ++c) {
            const int n = (iters - done + (L - c) - 1) / (L - c);
            launch_band_vec<0>(g, pl, cur, nxt, const_cast<float *>(div), nullptr, nullptr, n, st);
            done += n;
            float *t = cur; cur = nxt; nxt = t;
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            hipLaunchKernelGGL(k_jacobi_sweep, grid, block, 0, st, g, cur, nxt, div);
            float *t = cur; cur = nxt; nxt = t;
        }
    }
    if (cur != p) hipLaunchKernelGGL(k_copy_cells, grid, block, 0, st, g, cur, p);
    return hipGetLastError();
}

// pressure_projection (navier_stokes.py:133-149) on (u, v, p): divergence, `iters` Jacobi sweeps, gradient subtraction.
// With a band plan the divergence is computed inside the first Jacobi launch and the gradient subtraction inside the last.
hipError_t project_sync_create(ProjectSync &ps, int B) {
    ps.flags_len = B * 64 + 1;
    hipError_t e = hipMalloc((void **)&ps.flags, ps.flags_len * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(ps.flags, 0, ps.flags_len * sizeof(unsigned));
    if (e == hipSuccess) e = hipHostMalloc((void **)&ps.status, sizeof(unsigned), hipHostMallocMapped);
    if (e == hipSuccess) *ps.status = 0u;
    return e;
}
void project_sync_destroy(ProjectSync &ps) {
    if (ps.flags) (void)hipFree(ps.flags);
    if (ps.status) (void)hipHostFree((void *)ps.status);
    ps.flags = nullptr;
    ps.status = nullptr;
}

static bool use_persist(const Geom &g, const ProjectSync *ps, int iters, JacobiPlan &pl, int &chunks) {
    if (!ps || !ps->flags || ps->disabled || !knobs().persist || knobs().unfused || iters < 2) return false;
    if (!plan_jacobi(g, pl, iters, true) || pl.halo < 3) return false;
    if (pl.nb > device_num_cu()) return false;
    return persist_chunks(g, pl, iters, chunks);
}

// Two persistent projections cannot share the device: each needs all of its workgroups resident at once, and two half-resident grids
// would wait for each other until their spins run out.  Launches from ONE stream are ordered anyway; when a launch comes on a different
// stream than the previous one (of this process, on this device), the new stream is made to wait for everything submitted to the previous
// one (an event recorded there now) -- no host stall, and nothing at all in the single-stream case.  Other PROCESSES on the device are
// beyond this: see the bounded waits.
static std::unique_lock<std::mutex> order_persistent_launch(hipStream_t st) {
    struct Last { hipStream_t stream = nullptr; hipEvent_t ev = nullptr; bool any = false; };
    static std::mutex mu;
    static std::map<int, Last> last;
    int dev = 0;
    (void)hipGetDevice(&dev);
    // The caller keeps the lock until its persistent kernels are enqueued: a second host thread must not record its event on this
    // stream before the launch it is meant to wait for has been submitted.
    std::unique_lock<std::mutex> lk(mu);
    Last &l = last[dev];
    if (l.any && l.stream != st) {
        if (!l.ev) (void)hipEventCreateWithFlags(&l.ev, hipEventDisableTiming);
        if (l.ev && hipEventRecord(l.ev, l.stream) == hipSuccess) (void)hipStreamWaitEvent(st, l.ev, 0);
        else (void)hipGetLastError();                         // (the previous stream may have been destroyed: nothing of it is left to wait for)
    }
    l.stream = st;
    l.any = true;
    return lk;
}

// A bounded wait of an earlier persistent launch of this handle ran out (the kernel set the host-visible word and turned the band's results
// into NaN).  Acknowledge it: the word is cleared, the handle uses the multi-launch form from here on, and the caller reports the error
// exactly once.  The word is meaningful after the stream has been synchronised; read earlier it may simply not be set yet.
bool project_sync_take_timeout(ProjectSync &ps) {
    if (!ps.status || *ps.status == 0u) return false;
    *ps.status = 0u;
    ps.disabled = true;
    return true;
}

hipError_t launch_project(const Geom &g, float *u, float *v, float *p, float *p2, float *div, int iters, hipStream_t st, ProjectSync *ps,
                          const StateView *fold_in, float *fold_d_out, bool *folded) {
    JacobiPlan pl;
    if (folded) *folded = false;
    if (ps && project_sync_take_timeout(*ps)) {
        // a wait inside an earlier persistent launch timed out (its workgroups were not co-resident within the limit): that projection's
        // result is invalid (NaN).  Say so once, loudly, and use the multi-launch form from here on.
        return hipErrorLaunchTimeOut;
    }
    int chunks = 0;
    // (a captured launch would replay with the hand-off count of capture time: under stream capture the multi-launch form is recorded)
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap_status) == hipSuccess && cap_status != hipStreamCaptureStatusNone;
    if (!capturing && use_persist(g, ps, iters, pl, chunks)) {
        int per = device_num_cu() / pl.nb;                    // grids whose bands are all co-resident (one 1024-thread workgroup per CU)
        if (per >= 8) per &= ~7;
        if (knobs().debug)
            fprintf(stderr, "[smk] project %dx%dx%d J=%d: persistent, vec=%d rpw=%d nb=%d halo=%d, %d chunks, %d grids per launch\n", g.B, g.H, g.W,
                    iters, pl.vec, pl.rpw, pl.nb, pl.halo, chunks, per);
        JacobiSync sy{};
        sy.flags = ps->flags; sy.status = const_cast<unsigned *>(ps->status); sy.x0 = div; sy.x1 = p2;
        sy.base = ps->seq; sy.chunks = chunks; sy.nb = pl.nb; sy.abort_slot = ps->flags_len - 1; sy.fault = knobs().fault ? 1 : 0;
        sy.timeout_ticks = knobs().fault ? 200000ll : 50000000ll;      // 100 MHz wall clock: 2 ms under fault injection, 0.5 s otherwise
        ps->seq += (unsigned)chunks;
        const std::unique_lock<std::mutex> launch_order = order_persistent_launch(st);
        // the buoyancy + diffusion stage as this launch's prologue (16-byte row accesses: pitches in multiples of 4; up to 4 cells per lane)
        const bool fold = fold_in && fold_d_out && knobs().fold && pl.vec <= 4 && g.pv % 4 == 0 && g.pc % 4 == 0;
        if (fold) {
            sy.u_in = fold_in->u; sy.v_in = fold_in->v; sy.d_in = fold_in->d; sy.d_out = fold_d_out;
            if (folded) *folded = true;
        }
        for (int g0 = 0; g0 < g.B; g0 += per) {
            sy.grid0 = g0;
            sy.ngrids = g.B - g0 < per ? g.B - g0 : per;
            if (fold) launch_persist_vec<true>(g, pl, p, u, v, iters, sy, st);
            else launch_persist_vec<false>(g, pl, p, u, v, iters, sy, st);
        }
        return hipGetLastError();
    }
    if (iters < 2 || !plan_jacobi(g, pl, iters) || pl.halo < 3 || knobs().unfused) {
        hipError_t e = launch_divergence(g, u, v, div, g.pc, g.sc, st);
        if (e != hipSuccess) return e;
        e = launch_jacobi(g, p, p2, div, iters, st);
        if (e != hipSuccess) return e;
        return launch_grad_subtract(g, u, v, p, st);
    }
    const int cap = pl.halo - 1;                              // the fused gradient needs the row above the owned range exact
    const int L = 2 * ((iters + 2 * cap - 1) / (2 * cap));
    if (knobs().debug)
        fprintf(stderr, "[smk] project %dx%dx%d J=%d: band plan vec=%d rpw=%d nb=%d halo=%d -> %d launches\n", g.B, g.H, g.W, iters, pl.vec,
                pl.rpw, pl.nb, pl.halo, L);
    float *cur = p, *nxt = p2;
    int done = 0;
    for (int c = 0; c < L; ++c) {
        const int n = (iters - done + (L - c) - 1) / (L - c);
        if (c == 0) launch_band_vec<1>(g, pl, cur, nxt, div, u, v, n, st);
        else if (c == L - 1) launch_band_vec<2>(g, pl, cur, nxt, div, u, v, n, st);
        else launch_band_vec<0>(g, pl, cur, nxt, div, u, v, n, st);
        done += n;
        float *t = cur; cur = nxt; nxt = t;
    }
    return hipGetLastError();                                 // L is even: the result is back in p
}

// Stages 1-3 of a time step (navier_stokes.py:154-163): buoyancy + diffusion (in -> out.u, out.v, out.d) and the projection of
// (out.u, out.v) with the pressure p.  One persistent launch where the plan allows, otherwise the two stages as before.
hipError_t launch_buoy_project(const Geom &g, StateView in, StateView out, float *p, float *div, int iters, hipStream_t st, ProjectSync *ps) {
    bool folded = false;
    JacobiPlan pl;
    int chunks = 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    const bool pending_error = ps && ps->status && *ps->status != 0u;
    if (!capturing && !pending_error && use_persist(g, ps, iters, pl, chunks) && knobs().fold && pl.vec <= 4 && g.pv % 4 == 0 && g.pc % 4 == 0) {
        const hipError_t e = launch_project(g, out.u, out.v, p, out.p, div, iters, st, ps, &in, out.d, &folded);
        if (e != hipSuccess || folded) return e;
        return hipErrorUnknown;                               // (unreachable: the conditions above are launch_project's own)
    }
    hipError_t e = launch_buoy_diffuse(g, in, out, st);
    if (e != hipSuccess) return e;
    return launch_project(g, out.u, out.v, p, out.p, div, iters, st, ps);
}

// What one projection launches for this geometry, as a JSON object (bench.py reports it as the stencil pass's on-chip bound: the
// Jacobi sweeps never touch HBM, so what limits them is sweeps x rows per workgroup x vector-issue time, not bytes).
std::string describe_projection(const Geom &g, int iters, const ProjectSync *ps) {
    JacobiPlan pl;
    char buf[1280];
    int chunks = 0;
    const bool persist = use_persist(g, ps, iters, pl, chunks);
    if (!persist && (iters < 2 || !plan_jacobi(g, pl, iters) || pl.halo < 3 || knobs().unfused)) {
        snprintf(buf, sizeof buf, "{\"kernel\": \"k_jacobi_sweep\", \"launches\": %d, \"sweeps\": %d, \"bound\": \"hbm (one pass over p and div per sweep)\"}",
                 iters + 2, iters);
        return buf;
    }
    const int cap = pl.halo - 1, TR = JB_NW * pl.rpw;
    const int per = persist ? (device_num_cu() / pl.nb >= 8 ? (device_num_cu() / pl.nb) & ~7 : device_num_cu() / pl.nb) : 0;
    const int L = persist ? (g.B + per - 1) / per : 2 * ((iters + 2 * cap - 1) / (2 * cap));
    const int parts = persist ? chunks : L;                   // runs of sweeps between two refreshes of the halo rows
    const double wgs = (double)pl.nb * g.B, rounds = ceil(wgs / device_num_cu());
    // measured (tools/probes/valu_probe, 4 waves per SIMD): a sweep row of 64 VEC-cell lanes = ~18 vector instructions of which 2 are DPP
    // wave shifts, ~2.6 cycles per instruction and SIMD -> TR rows on 4 SIMDs; plus the publish -> s_barrier -> read round trip per sweep
    const double valu_us_per_sweep = rounds * (TR / 4.0) * 18.0 * 2.6 / 2100.0;
    snprintf(buf, sizeof buf,
             "{\"kernel\": \"k_jacobi_band<%d,%d>\", \"persistent\": %s, \"bands_per_grid\": %d, \"rows_per_workgroup\": %d, \"halo_rows\": %d, "
             "\"workgroups\": %d, \"launches\": %d, \"halo_handoffs\": %d, \"sweeps\": %d, \"sweeps_per_chunk\": %d, \"redundant_row_factor\": %.3f, "
             "\"vector_issue_us_per_sweep_estimate\": %.3f, \"vector_issue_us_total_estimate\": %.1f, "
             "\"bound\": \"on-chip: sweeps x (vector issue of rows_per_workgroup rows + one LDS publish/barrier/read round trip); p and div are "
             "register-resident %s\"}",
             pl.vec, pl.rpw, persist ? "true" : "false", pl.nb, TR, pl.halo, (int)wgs, L, persist ? chunks - 1 : 0, iters, (iters + parts - 1) / parts,
             (double)pl.nb * TR / g.H, valu_us_per_sweep, valu_us_per_sweep * iters,
             persist ? "for the whole projection; bands hand halo rows to their neighbours through HBM between chunks" : "within a launch");
    return buf;
}

// u[1:-1,:] -= dt*(p[1:]-p[:-1]);  v[:,1:-1] -= dt*(p[:,1:]-p[:,:-1])   (:148-149)
__global__ void k_grad_subtract(Geom g, float *u, float *v, const float *p) {
    int b = blockIdx.z;
    int j = blockIdx.x * TX + threadIdx.x, i = blockIdx.y * TY + threadIdx.y;
    const float *pb = p + b * g.sc;
    if (i >= 1 && i < g.H && j < g.W) {
        float *c = u + b * g.su + (size_t)i * g.pc + j;
        float gr = pb[(size_t)i * g.pc + j] - pb[(size_t)(i - 1) * g.pc + j];
        *c = *c - g.dt * gr;
    }
    if (i < g.H && j >= 1 && j < g.W) {
        float *c = v + b * g.sv + (size_t)i * g.pv + j;
        float gr = pb[(size_t)i * g.pc + j] - pb[(size_t)i * g.pc + j - 1];
        *c = *c - g.dt * gr;
    }
}

hipError_t launch_grad_subtract(const Geom &g, float *u, float *v, const float *p, hipStream_t st) {
    dim3 grid(cdiv(g.W, TX), cdiv(g.H, TY), g.B), block(TX, TY);
    hipLaunchKernelGGL(k_grad_subtract, grid, block, 0, st, g, u, v, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------- advection (navier_stokes.py:74-131)
// bilinear_interpolate (:111-131): indices clamped before the weights are formed.
__device__ __forceinline__ float bilinear(const float *f, int h, int w, int pitch, float y, float x, int *ox0, int *oy0) {
    int x0 = (int)floorf(x), y0 = (int)floorf(y);
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = clampi(x0, 0, w - 1); x1 = clampi(x1, 0, w - 1);
    y0 = clampi(y0, 0, h - 1); y1 = clampi(y1, 0, h - 1);
    float fx0 = (float)x0, fx1 = (float)x1, fy0 = (float)y0, fy1 = (float)y1;
    float wa = (fx1 - x) * (fy1 - y);
    float wb = (x - fx0) * (fy1 - y);
    float wc = (fx1 - x) * (y - fy0);
    float wd = (x - fx0) * (y - fy0);
    float r = wa * f[(size_t)y0 * pitch + x0] + wb * f[(size_t)y0 * pitch + x1];
    r = r + wc * f[(size_t)y1 * pitch + x0];
    r = r + wd * f[(size_t)y1 * pitch + x1];
    if (ox0) { *ox0 = x0; *oy0 = y0; }
    return r;
}

template <int KIND>
__global__ void k_advect(Geom g, const float *field, float *out, const float *u, const float *v, float *frames,
                         int64_t fsb, const float *fractal, float fint, int32_t *x0o, int32_t *y0o) {
    constexpr bool IS_U = KIND == 0, IS_V = KIND == 1;   // KIND 2: density (+decay, frame emit); 3: plain cell field
    const int R = IS_U ? g.H + 1 : g.H, C = IS_V ? g.W + 1 : g.W;
    const int pitch = IS_V ? g.pv : g.pc;
    const size_t fs = IS_U ? g.su : (IS_V ? g.sv : g.sc);
    int b = blockIdx.z;
    int j = blockIdx.x * TX + threadIdx.x, i = blockIdx.y * TY + threadIdx.y;
    if (i >= R || j >= C) return;
    const float *ub = u + b * g.su, *vb = v + b * g.sv, *fb = field + b * fs;
    float Y = (float)i, X = (float)j;
    float xu = clampf(X + 0.5f, 0.f, (float)(g.W - 1));           // interpolate_velocity_u (:97-102)
    float ui = bilinear(ub, g.H + 1, g.W, g.pc, Y, xu, nullptr, nullptr);
    float yv = clampf(Y + 0.5f, 0.f, (float)(g.H - 1));           // interpolate_velocity_v (:104-109)
    float vi = bilinear(vb, g.H, g.W + 1, g.pv, yv, X, nullptr, nullptr);
    float px = clampf(X - g.dt * ui, 0.f, (float)(C - 1));        // :87-92
    float py = clampf(Y - g.dt * vi, 0.f, (float)(R - 1));
    int x0, y0;
    float r = bilinear(fb, R, C, pitch, py, px, &x0, &y0);
    if (x0o) {
        size_t o = ((size_t)b * R + i) * C + j;
        x0o[o] = x0; y0o[o] = y0;
    }
    if (KIND == 2) {
        r = r * 0.995f;                                           // :171
        if (frames) {
            float fr = r;
            if (fractal) {                                        // fractal_generator.py:62 (F is [W][H], square)
                float t = fint * fractal[(size_t)i * g.W + j];
                t = t * r;
                fr = r + t;
            }
            frames[(size_t)b * fsb + (size_t)i * g.W + j] = fr;
        }
    }
    if (out) out[b * fs + (size_t)i * pitch + j] = r;
}

hipError_t launch_advect(const Geom &g, int kind, const float *field, float *out, const float *u, const float *v,
                         float *frames, int64_t fsb, const float *fractal, float fint, int32_t *x0, int32_t *y0,
                         hipStream_t st) {
    dim3 block(TX, TY);
    if (kind == 0) {
        dim3 grid(cdiv(g.W, TX), cdiv(g.H + 1, TY), g.B);
        hipLaunchKernelGGL(k_advect<0>, grid, block, 0, st, g, field, out, u, v, frames, fsb, fractal, fint, x0, y0);
    } else if (kind == 1) {
        dim3 grid(cdiv(g.W + 1, TX), cdiv(g.H, TY), g.B);
        hipLaunchKernelGGL(k_advect<1>, grid, block, 0, st, g, field, out, u, v, frames, fsb, fractal, fint, x0, y0);
    } else if (kind == 3) {
        dim3 grid(cdiv(g.W, TX), cdiv(g.H, TY), g.B);
        hipLaunchKernelGGL(k_advect<3>, grid, block, 0, st, g, field, out, u, v, frames, fsb, fractal, fint, x0, y0);
    } else {
        dim3 grid(cdiv(g.W, TX), cdiv(g.H, TY), g.B);
        hipLaunchKernelGGL(k_advect<2>, grid, block, 0, st, g, field, out, u, v, frames, fsb, fractal, fint, x0, y0);
    }
    return hipGetLastError();
}


// ---- the three advections of a step as ONE launch (navier_stokes.py:166-171) --------------------------------------------------------
// u <- adv(u2; u2, v2), v <- adv(v2; u, v2), density <- adv(d2; u, v) * 0.995 are sequentially dependent, but only through the
// velocity SAMPLING, which happens at the integer cell coordinates (advection_step builds Y, X with meshgrid, :79-81), where
// interpolate_velocity_u / _v (:97-109) collapse exactly:
//   u at (Y = i, x + 0.5): y0 = i with weight (y1 - i) in {1, 0}, x-weights 0.5 / 0.5  ->  ui = 0.5*U[i][j] + 0.5*U[i][j+1]   (i <= H-1, j <= W-2)
//   v at (y + 0.5, X = j): x0 = j with weight (x1 - j) in {1, 0}, y-weights 0.5 / 0.5  ->  vi = 0.5*V[i][j] + 0.5*V[i+1][j]   (i <= H-2, j <= W-1)
//   and 0 otherwise (both clamped indices coincide: the zero-at-the-upper-edge quirk).  The dropped terms are products with an exact
//   zero weight: for finite fields they are +-0 and change at most the sign of a zero, which X - dt*ui cannot see (X >= 0).
// The displacement-dependent gathers read u2, v2, d2 -- inputs of the launch.  So a workgroup that owns a TH x TW tile of cells stages
// u2, v2, d2 (tile + 1 halo, + 2 on the high side for u2 / v2) in LDS, computes the advected u on the tile extended by one row and
// column and the advected v on the tile extended by one row (redundantly with its neighbours: ~5 % more points), keeps both in LDS
// for the next field's sampling, and writes its own cells of u, v, density and the emitted frame.  A back-trace that leaves the
// staged window (|dt * velocity| >= 1 cell: never in the reference's regime, |velocity| ~ 0.1) gathers from global memory instead
// -- one wave-level branch per cell, same values either way.  Per point the gather arithmetic is bilinear()'s (one rounding per
// reference op).  HBM traffic: u2, v2, d2, fractal in; u, v, density, frame out -- 8 floats per cell instead of the 12 of three
// launches; vector instructions per cell and field: ~60 instead of ~150 (k_advect evaluates all three bilinears in the general form).
constexpr int AF_TH = 32, AF_TW = 64, AF_THREADS = 256;

// the final gather of advection_step: bilinear_interpolate(field[R][C], py, px) with py, px already clamped into [0, R-1] x [0, C-1]
// (:91-92), so floor == truncation and the lower index clamps of :120-123 are no-ops.  `f` = window in LDS (top-left cell (oi, oj),
// nr x nc, pitch lp) or, when the four taps do not all lie inside it, the field in global memory (pitch gp).
// (the fallback is a real call: inlined, the compiler merges the two arms into ONE flat_load through a selected generic pointer, and
// flat loads of LDS addresses run at a fraction of ds_read's rate -- the first version of this kernel took 75 us instead of 3 x 19)
__device__ __attribute__((noinline)) void gather_far(const float *glob, int gp, int y0, int y1, int x0, int x1, float (&f)[4]) {
    f[0] = glob[(size_t)y0 * gp + x0]; f[1] = glob[(size_t)y0 * gp + x1];
    f[2] = glob[(size_t)y1 * gp + x0]; f[3] = glob[(size_t)y1 * gp + x1];
}

__device__ __forceinline__ float gather_cell(const float *lds, int oi, int oj, int nr, int nc, int lp, const float *glob, int gp, int R,
                                             int C, float py, float px) {
    const int x0 = (int)px, y0 = (int)py;
    int x1 = x0 + 1, y1 = y0 + 1;
    x1 = x1 > C - 1 ? C - 1 : x1;
    y1 = y1 > R - 1 ? R - 1 : y1;
    const float fx0 = (float)x0, fx1 = (float)x1, fy0 = (float)y0, fy1 = (float)y1;
    const float wa = (fx1 - px) * (fy1 - py);
    const float wb = (px - fx0) * (fy1 - py);
    const float wc = (fx1 - px) * (py - fy0);
    const float wd = (px - fx0) * (py - fy0);
    float f[4];
    const int r0 = y0 - oi, r1 = y1 - oi, c0 = x0 - oj, c1 = x1 - oj;
    if (__builtin_expect(r0 >= 0 && r1 < nr && c0 >= 0 && c1 < nc, 1)) {
        f[0] = lds[r0 * lp + c0]; f[1] = lds[r0 * lp + c1]; f[2] = lds[r1 * lp + c0]; f[3] = lds[r1 * lp + c1];
    } else {
        gather_far(glob, gp, y0, y1, x0, x1, f);
    }
    float r = wa * f[0] + wb * f[1];
    r = r + wc * f[2];
    r = r + wd * f[3];
    return r;
}

__global__ __launch_bounds__(AF_THREADS) void k_advect_fused(Geom g, StateView in, StateView out, float *frames, int64_t fsb,
                                                            const float *fractal, float fint) {
    constexpr int TH = AF_TH, TW = AF_TW;
    constexpr int UR = TH + 3, UC = TW + 3;      // u2 / v2 / d2 windows: rows i0-1 .. i0+TH+1, columns j0-1 .. j0+TW+1
    constexpr int NR = TH + 1, NC = TW + 1;      // advected u and v: rows i0 .. i0+TH, columns j0 .. j0+TW
    __shared__ float us[UR * UC], vs[UR * UC], ds[UR * UC], un[NR * NC], vn[NR * NC];
    const int b = blockIdx.z, i0 = blockIdx.y * TH, j0 = blockIdx.x * TW, tid = threadIdx.x;
    const int i1 = i0 + TH < g.H ? i0 + TH : g.H, j1 = j0 + TW < g.W ? j0 + TW : g.W;
    const int H = g.H, W = g.W;
    const float *u2 = in.u + b * g.su, *v2 = in.v + b * g.sv, *d2 = in.d + b * g.sc;
    // stage the inputs: all loads of a thread are issued before the first LDS write (addresses clamped into the field instead of
    // branches: a load under a branch waits for its data before the next one is issued -- 30 serial HBM latencies per thread)
    constexpr int NST = (UR * UC + AF_THREADS - 1) / AF_THREADS;
    float ru[NST], rv[NST], rd[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        int idx = tid + it * AF_THREADS;
        idx = idx < UR * UC ? idx : UR * UC - 1;
        const int r = idx / UC, c = idx - r * UC, gi = i0 - 1 + r, gj = j0 - 1 + c;
        const int ci = clampi(gi, 0, H - 1), cj = clampi(gj, 0, W - 1);
        ru[it] = u2[(size_t)clampi(gi, 0, H) * g.pc + cj];
        rv[it] = v2[(size_t)ci * g.pv + clampi(gj, 0, W)];
        rd[it] = d2[(size_t)ci * g.pc + cj];
    }
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int idx = tid + it * AF_THREADS;
        if (idx < UR * UC) { us[idx] = ru[it]; vs[idx] = rv[it]; ds[idx] = rd[it]; }
    }
    __syncthreads();
    // window-local index of cell (i, j) of the staged inputs
    auto wi = [&](int i, int j) { return (i - i0 + 1) * UC + (j - j0 + 1); };
    // 1. u <- adv(u2; u2, v2) on rows i0 .. i0+TH, columns j0 .. j0+TW (field shape [H+1][W])
    float *uo = out.u + b * g.su;
    // thread (ty, tx) walks rows ty, ty + 4, ... of column tx (no index division); the extra column TW is a pass of its own
    const int tx = tid & 63, ty = tid >> 6;
    auto adv_u = [&](int r, int c) {
        const int i = i0 + r, j = j0 + c, idx = r * NC + c;
        if (i > H || j >= W) return;
        float ui = 0.f, vi = 0.f;
        if (i <= H - 1 && j <= W - 2) ui = 0.5f * us[wi(i, j)] + 0.5f * us[wi(i, j + 1)];
        if (i <= H - 2) vi = 0.5f * vs[wi(i, j)] + 0.5f * vs[wi(i + 1, j)];
        const float px = clampf((float)j - g.dt * ui, 0.f, (float)(W - 1));
        const float py = clampf((float)i - g.dt * vi, 0.f, (float)H);
        const float val = gather_cell(us, i0 - 1, j0 - 1, UR, UC, UC, u2, g.pc, H + 1, W, py, px);
        un[idx] = val;
        if ((i < i1 || (i == H && i1 == H)) && j < j1) uo[(size_t)i * g.pc + j] = val;
    };
#pragma unroll 3
    for (int r = ty; r < NR; r += 4) adv_u(r, tx);
    if (tid < NR) adv_u(tid, TW);
    __syncthreads();
    // 2. v <- adv(v2; u, v2) on rows i0 .. i0+TH, columns j0 .. j0+TW (field shape [H][W+1]); u = the advected u in LDS
    float *vo = out.v + b * g.sv;
    auto adv_v = [&](int r, int c) {
        const int i = i0 + r, j = j0 + c, idx = r * NC + c;
        if (i >= H || j > W) return;
        float ui = 0.f, vi = 0.f;
        if (j <= W - 2) ui = 0.5f * un[idx] + 0.5f * un[idx + 1];
        if (i <= H - 2 && j <= W - 1) vi = 0.5f * vs[wi(i, j)] + 0.5f * vs[wi(i + 1, j)];
        const float px = clampf((float)j - g.dt * ui, 0.f, (float)W);
        const float py = clampf((float)i - g.dt * vi, 0.f, (float)(H - 1));
        const float val = gather_cell(vs, i0 - 1, j0 - 1, UR, UC, UC, v2, g.pv, H, W + 1, py, px);
        vn[idx] = val;
        if (i < i1 && (j < j1 || j == W)) vo[(size_t)i * g.pv + j] = val;
    };
#pragma unroll 3
    for (int r = ty; r < NR; r += 4) adv_v(r, tx);
    if (tid < NR && j0 + TW == W) adv_v(tid, TW);             // column j0+TW is needed only as the field's last column
    __syncthreads();
    // 3. density <- adv(d2; u, v) * 0.995 (+ frame emit) on the tile; u, v = the advected fields in LDS
#pragma unroll 4
    for (int r = ty; r < TH; r += 4) {
        const int c = tx, i = i0 + r, j = j0 + c;
        if (i >= i1 || j >= j1) continue;
        const int n = r * NC + c;
        float ui = 0.f, vi = 0.f;
        if (j <= W - 2) ui = 0.5f * un[n] + 0.5f * un[n + 1];
        if (i <= H - 2) vi = 0.5f * vn[n] + 0.5f * vn[n + NC];
        const float px = clampf((float)j - g.dt * ui, 0.f, (float)(W - 1));
        const float py = clampf((float)i - g.dt * vi, 0.f, (float)(H - 1));
        float val = gather_cell(ds, i0 - 1, j0 - 1, UR, UC, UC, d2, g.pc, H, W, py, px);
        val = val * 0.995f;                                           // :171
        if (frames) {
            float fr = val;
            if (fractal) {                                            // fractal_generator.py:62 (F is [W][H], square)
                float t = fint * fractal[(size_t)i * W + j];
                t = t * val;
                fr = val + t;
            }
            frames[(size_t)b * fsb + (size_t)i * W + j] = fr;
        }
        out.d[b * g.sc + (size_t)i * g.pc + j] = val;
    }
}

// ---- the same launch with wave-autonomous rows (round 4; the organisation of csrc/advect3d.hip's k3_advect_march, one plane) ------------
// One advected value in the general form, every tap from GLOBAL memory: what a wave falls back to when a back-trace of its units leaves the
// 2 x 2 LDS neighbourhood (never in the reference's regime).  A real call, so the hot code carries neither its registers nor its loads.
// Field [Rf][Cf] (pitch `pitch`); (y, x) = the cell; ui, vi = the velocity samples already formed by the caller.
__device__ __attribute__((noinline)) float far_value2(const float *f, int pitch, int Rf, int Cf, float dt, int y, int x, float ui, float vi) {
    const float px = clampf((float)x - dt * ui, 0.f, (float)(Cf - 1));
    const float py = clampf((float)y - dt * vi, 0.f, (float)(Rf - 1));
    const int x0 = (int)px, y0 = (int)py;
    int x1 = x0 + 1, y1 = y0 + 1;
    x1 = x1 > Cf - 1 ? Cf - 1 : x1;
    y1 = y1 > Rf - 1 ? Rf - 1 : y1;
    const float fx0 = (float)x0, fx1 = (float)x1, fy0 = (float)y0, fy1 = (float)y1;
    const float wa = (fx1 - px) * (fy1 - py);
    const float wb = (px - fx0) * (fy1 - py);
    const float wc = (fx1 - px) * (py - fy0);
    const float wd = (px - fx0) * (py - fy0);
    float r = wa * f[(size_t)y0 * pitch + x0] + wb * f[(size_t)y0 * pitch + x1];
    r = r + wc * f[(size_t)y1 * pitch + x0];
    r = r + wd * f[(size_t)y1 * pitch + x1];
    return r;
}


// Workgroup = NW waves = a (NW R) x 64 tile; u2, v2, d2 windows (tile + 1 low, + 2 high; every out-of-grid element holds the CLAMPED in-grid
// value, which is exactly what bilinear()'s clamped indices read) staged once into LDS, ONE barrier.  A wave then owns R consecutive rows:
// thread = column, the advected u and v of its rows (+ the next row) live in REGISTERS -- x + 1 comes from the next lane by DPP, lane 63's
// from the extra-column unit by v_readlane -- so nothing separates the three fields (k_advect_fused: two more barriers and two LDS round
// trips).  With |dt * velocity| < 1 cell a back-trace lands on floor(p) in {i-1, i}: four taps at immediate offsets from ONE LDS address;
// the test is one wave-level vote per batch of units, the fall-back the general form per lane (far_value2).  EDGE = false is the same
// arithmetic with the tests that cannot fail inside the grid removed (existence, the sampling rules' extents, the clamps: an unclamped
// back-trace that would have needed its clamp fails the neighbourhood test).  Bit-identical to k_advect_fused.
template <int R, int NW, int BU>
__global__ __launch_bounds__(NW * 64) void k_advect_rows(Geom g, StateView in, StateView out, float *frames, int64_t fsb,
                                                         const float *__restrict__ fractal, float fint) {
    constexpr int TYR = R * NW, TXC = 64, NT = NW * 64;
    constexpr int WR = TYR + 3, WC = TXC + 3, WP = 68, WPL = WR * WP;
    constexpr int NST = (WR * WC + NT - 1) / NT;
    __shared__ float U2s[WPL], V2s[WPL], D2s[WPL];
    const int H = g.H, W = g.W, pc = g.pc, pv = g.pv;
    const int ntx = (W + TXC - 1) / TXC, nty = (H + TYR - 1) / TYR;
    unsigned tile = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tix = tile % ntx; tile /= ntx;
    const int tiy = tile % nty;
    const int b = tile / nty;
    const int x0 = tix * TXC, y0 = tiy * TYR;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *u2 = in.u + b * g.su, *v2 = in.v + b * g.sv, *d2 = in.d + b * g.sc;
    float *uo = out.u + b * g.su, *vo = out.v + b * g.sv, *dn = out.d + b * g.sc;
    float *fr = frames ? frames + (size_t)b * fsb : nullptr;
    const float dt = g.dt;

    // ---- stage the windows: every address clamped into its field (no load under a branch), all loads before the first LDS write
    {
        float ru[NST], rv[NST], rd[NST];
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int e0 = tid + it * NT, e = e0 < WR * WC ? e0 : WR * WC - 1;
            const int r = e / WC, c = e - r * WC;
            const int gy = y0 - 1 + r, gx = x0 - 1 + c;
            const int yc = clampi(gy, 0, H - 1), xc = clampi(gx, 0, W - 1);
            ru[it] = u2[(unsigned)(clampi(gy, 0, H) * pc + xc)];
            rv[it] = v2[(unsigned)(yc * pv + clampi(gx, 0, W))];
            rd[it] = d2[(unsigned)(yc * pc + xc)];
        }
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int e0 = tid + it * NT;
            const int r = e0 / WC, c = e0 - r * WC;
            if (e0 < WR * WC) { U2s[r * WP + c] = ru[it]; V2s[r * WP + c] = rv[it]; D2s[r * WP + c] = rd[it]; }
        }
    }
    __syncthreads();

    const int rb = wv * R, yb = y0 + rb;                                  // this wave's rows: yb .. yb+R-1 (+ row yb+R for Un, Vn)
    const int x = x0 + lane;
    const float fxl = (float)x;
    const int x4 = 4 * x;
    auto stb = [&](float *rowbase, float v) {     // one value of this lane's column into a row (uniform base): descriptor base = the row
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), __builtin_amdgcn_make_buffer_rsrc(rowbase, 0, 0x7fffffff, 0x00020000), x4, 0, 0);
    };
    const int wbase = (rb + 1) * WP + lane + 1;                           // window index of (row yb, column x); + r WP for row yb + r
    const int wex = (rb + 1 + (lane <= R ? lane : R)) * WP + TXC + 1;     // extra-column unit: lane = row, column x0 + TXC
    const float fxe = (float)(x0 + TXC);
    // is this wave's whole neighbourhood inside the grid?  (rows yb-1 .. yb+R+1 and columns x0-1 .. x0+TXC+1 exist in every field, the
    // sampling rules hold on all its units, and no coordinate below 2: (int)NaN = 0 must fail the neighbourhood test)
    const bool inner_w = yb >= 2 && yb + R + 1 <= H - 2 && x0 >= 2 && x0 + TXC + 1 <= W - 2;

    struct Prep { float wx0, wx1, wy0, wy1; int a; };
    // back-trace -> weights + the LDS address; returns "lands in the neighbourhood".  Field [Rf][Cf].
    auto prep = [&](auto edge, int Rf, int Cf, int widx, int y, int xx, float fx, float ui, float vi, bool exists, Prep &P) -> bool {
        constexpr bool EDGE = decltype(edge)::value;
        float px = fx - dt * ui, py = (float)y - dt * vi;
        if (EDGE) { px = clampf(px, 0.f, (float)(Cf - 1)); py = clampf(py, 0.f, (float)(Rf - 1)); }
        const float fx0 = floorf(px), fy0 = floorf(py);
        float fx1 = fx0 + 1.f, fy1 = fy0 + 1.f;
        if (EDGE) { fx1 = fminf(fx1, (float)(Cf - 1)); fy1 = fminf(fy1, (float)(Rf - 1)); }
        P.wx0 = fx1 - px; P.wx1 = px - fx0; P.wy0 = fy1 - py; P.wy1 = py - fy0;
        int rx = (int)fx0 - xx, ry = (int)fy0 - y;
        bool fast = (unsigned)((rx + 1) | (ry + 1)) <= 1u;
        if (EDGE && !exists) { rx = 0; ry = 0; fast = true; }             // a lane without a cell reads its own (in-window) slot; result unused
        P.a = widx + ry * WP + rx;
        return fast;
    };
    auto finish = [&](const float *win, const Prep &P) -> float {
        const float t0 = win[P.a], t1 = win[P.a + 1], t2 = win[P.a + WP], t3 = win[P.a + WP + 1];
        float acc = (P.wx0 * P.wy0) * t0 + (P.wx1 * P.wy0) * t1;
        acc = acc + (P.wx0 * P.wy1) * t2;
        acc = acc + (P.wx1 * P.wy1) * t3;
        return acc;
    };

    float un[R + 1], unx[R + 1], vn[R + 1];
    auto body = [&](auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
        float ui[BU], vi[BU], val[BU];
        Prep P[BU];
        bool ex[BU];
        // ---------------- Un: units 0 .. R = rows yb .. yb+R at column x; unit R+1 = the tile's extra column x0 + TXC (lane = row).  u is [H+1][W]
        float uex = 0.f;
#pragma unroll
        for (int b0 = 0; b0 <= R + 1; b0 += BU) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < BU; ++j) {
                const int r = b0 + j;
                if (r > R + 1) continue;
                const bool xt = r == R + 1;
                const int y = xt ? yb + lane : yb + r, xx = xt ? x0 + TXC : x, w0 = xt ? wex : wbase + r * WP;
                const float a = 0.5f * U2s[w0] + 0.5f * U2s[w0 + 1];
                const float c = 0.5f * V2s[w0] + 0.5f * V2s[w0 + WP];
                ex[j] = (!EDGE && !xt) || ((!xt || lane <= R) && y <= H && xx <= W - 1);
                ui[j] = ((!EDGE && !xt) || (y <= H - 1 && xx <= W - 2)) ? a : 0.f;
                vi[j] = ((!EDGE && !xt) || y <= H - 2) ? c : 0.f;
                if (xt) ok &= prep(std::true_type{}, H + 1, W, w0, y, xx, fxe, ui[j], vi[j], ex[j], P[j]);
                else ok &= prep(edge, H + 1, W, w0, y, xx, fxl, ui[j], vi[j], ex[j], P[j]);
            }
            if (__builtin_expect(__all(ok), 1)) {
#pragma unroll
                for (int j = 0; j < BU; ++j)
                    if (b0 + j <= R + 1) val[j] = finish(U2s, P[j]);
            } else {
#pragma unroll
                for (int j = 0; j < BU; ++j) {
                    const int r = b0 + j;
                    if (r > R + 1) continue;
                    const bool xt = r == R + 1;
                    val[j] = ex[j] ? far_value2(u2, pc, H + 1, W, dt, xt ? yb + lane : yb + r, xt ? x0 + TXC : x, ui[j], vi[j]) : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < BU; ++j) {
                const int r = b0 + j;
                if (r > R + 1) continue;
                if (r == R + 1) { uex = val[j]; continue; }
                un[r] = val[j];
                const int y = yb + r;
                if ((r < R || (EDGE && y == H)) && ex[j]) stb(uo + (size_t)y * pc, val[j]);
            }
        }
#pragma unroll
        for (int r = 0; r <= R; ++r)                                       // Un at x + 1: the next lane's, lane 63 takes the extra column's row r
            unx[r] = shl1_with(un[r], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, uex), r)));
        // ---------------- Vn: rows 0 .. R.  v is [H][W+1] (x = W of a narrower last tile is one of these lanes)
#pragma unroll
        for (int b0 = 0; b0 <= R; b0 += BU) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < BU; ++j) {
                const int r = b0 + j;
                if (r > R) continue;
                const int y = yb + r, w0 = wbase + r * WP;
                const float a = 0.5f * un[r] + 0.5f * unx[r];
                const float c = 0.5f * V2s[w0] + 0.5f * V2s[w0 + WP];
                ex[j] = !EDGE || (y <= H - 1 && x <= W);
                ui[j] = (!EDGE || x <= W - 2) ? a : 0.f;
                vi[j] = (!EDGE || (y <= H - 2 && x <= W - 1)) ? c : 0.f;
                ok &= prep(edge, H, W + 1, w0, y, x, fxl, ui[j], vi[j], ex[j], P[j]);
            }
            if (__builtin_expect(__all(ok), 1)) {
#pragma unroll
                for (int j = 0; j < BU; ++j)
                    if (b0 + j <= R) val[j] = finish(V2s, P[j]);
            } else {
#pragma unroll
                for (int j = 0; j < BU; ++j)
                    if (b0 + j <= R) val[j] = ex[j] ? far_value2(v2, pv, H, W + 1, dt, yb + b0 + j, x, ui[j], vi[j]) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < BU; ++j) {
                const int r = b0 + j;
                if (r > R) continue;
                vn[r] = val[j];
                if (r < R && ex[j]) stb(vo + (size_t)(yb + r) * pv, val[j]);
            }
        }
        if (EDGE && x0 + TXC == W) {                                       // the field's own extra column x = W (lane = row, rows 0 .. R-1)
            const int y = yb + lane;
            if (lane < R && y <= H - 1)                                    // (ui needs x <= W-2, vi x <= W-1: both zero)
                vo[(unsigned)(y * pv + W)] = far_value2(v2, pv, H, W + 1, dt, y, W, 0.f, 0.f);
        }
        // ---------------- density: rows 0 .. R-1 (+ decay, frame)
#pragma unroll
        for (int b0 = 0; b0 < R; b0 += BU) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < BU; ++j) {
                const int r = b0 + j;
                if (r >= R) continue;
                const int y = yb + r, w0 = wbase + r * WP;
                const float a = 0.5f * un[r] + 0.5f * unx[r];
                const float c = 0.5f * vn[r] + 0.5f * vn[r + 1];
                ex[j] = !EDGE || (y <= H - 1 && x <= W - 1);
                ui[j] = (!EDGE || x <= W - 2) ? a : 0.f;
                vi[j] = (!EDGE || y <= H - 2) ? c : 0.f;
                ok &= prep(edge, H, W, w0, y, x, fxl, ui[j], vi[j], ex[j], P[j]);
            }
            if (__builtin_expect(__all(ok), 1)) {
#pragma unroll
                for (int j = 0; j < BU; ++j)
                    if (b0 + j < R) val[j] = finish(D2s, P[j]);
            } else {
#pragma unroll
                for (int j = 0; j < BU; ++j)
                    if (b0 + j < R) val[j] = ex[j] ? far_value2(d2, pc, H, W, dt, yb + b0 + j, x, ui[j], vi[j]) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < BU; ++j) {
                const int r = b0 + j;
                if (r >= R) continue;
                const int y = yb + r;
                if (!ex[j]) continue;
                const float v = val[j] * 0.995f;                              // :171
                if (fr) {
                    float f = v;
                    if (fractal) {                                            // fractal_generator.py:62 (F is [W][H], square)
                        float t = fint * fractal[(unsigned)(y * W + x)];
                        t = t * v;
                        f = v + t;
                    }
                    stb(fr + (size_t)y * W, f);
                }
                stb(dn + (size_t)y * pc, v);
            }
        }
    };
    if (inner_w) body(std::false_type{});
    else body(std::true_type{});
}

hipError_t launch_advect_fused(const Geom &g, StateView in, StateView out, float *frames, int64_t fsb, const float *fractal, float fint,
                               hipStream_t st) {
    static int rows_env = -1;            // SMK_ADVECT_ROWS=0: round 2's k_advect_fused (A/B runs)
    if (rows_env < 0) { const char *sv = getenv("SMK_ADVECT_ROWS"); rows_env = sv ? atoi(sv) : 1; }
    if (rows_env) {
        // R x NW = 8 x 4 rows per workgroup (measured at configs[2] against 4 x 4, 4 x 8, 8 x 8, 16 x 2 rows and batches of 5 units: equal within
        // noise except 16 x 2, +7 %)
        constexpr int R = 8, NW = 4;
        const long long n = (long long)cdiv(g.W, 64) * cdiv(g.H, R * NW) * g.B;
        hipLaunchKernelGGL((k_advect_rows<R, NW, 3>), dim3((unsigned)n), dim3(NW * 64), 0, st, g, in, out, frames, fsb, fractal, fint);
        return hipGetLastError();
    }
    dim3 grid(cdiv(g.W, AF_TW), cdiv(g.H, AF_TH), g.B), block(AF_THREADS);
    hipLaunchKernelGGL(k_advect_fused, grid, block, 0, st, g, in, out, frames, fsb, fractal, fint);
    return hipGetLastError();
}

// ---------------------------------------------------------------- the public interpolation helpers (navier_stokes.py:97-131)
// bilinear_interpolate / interpolate_velocity_u / interpolate_velocity_v as pure gathers on caller-given coordinates (any float:
// the reference floors to int64 and clamps, so coordinates far outside the field are legal).  floor() is saturated in float
// before the int conversion: every value below -2 / above dim + 1 clamps to the same indices as the int64 original.
__device__ __forceinline__ int floor_sat(float x, int dim) {
    float f = floorf(x);
    f = f < -2.0f ? -2.0f : f;
    f = f > (float)(dim + 1) ? (float)(dim + 1) : f;
    return (int)f;
}

// MODE 0: bilinear(field, y, x); 1: x <- clamp(x + 0.5, 0, w - 1) first (:97-102); 2: y <- clamp(y + 0.5, 0, h - 1) first (:104-109)
template <int MODE>
__global__ void k_interp(const float *field, int h, int w, int pitch, size_t fstride, const float *ys, const float *xs,
                         size_t cstride, size_t n, float *out) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int b = blockIdx.y;
    const float *f = field + b * fstride;
    float y = ys[b * cstride + k], x = xs[b * cstride + k];
    if (MODE == 1) x = clampf(x + 0.5f, 0.f, (float)(w - 1));
    if (MODE == 2) y = clampf(y + 0.5f, 0.f, (float)(h - 1));
    int x0 = floor_sat(x, w), y0 = floor_sat(y, h);
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = clampi(x0, 0, w - 1); x1 = clampi(x1, 0, w - 1);
    y0 = clampi(y0, 0, h - 1); y1 = clampi(y1, 0, h - 1);
    const float fx0 = (float)x0, fx1 = (float)x1, fy0 = (float)y0, fy1 = (float)y1;
    const float wa = (fx1 - x) * (fy1 - y);
    const float wb = (x - fx0) * (fy1 - y);
    const float wc = (fx1 - x) * (y - fy0);
    const float wd = (x - fx0) * (y - fy0);
    float r = wa * f[(size_t)y0 * pitch + x0] + wb * f[(size_t)y0 * pitch + x1];
    r = r + wc * f[(size_t)y1 * pitch + x0];
    r = r + wd * f[(size_t)y1 * pitch + x1];
    out[(size_t)b * n + k] = r;
}

hipError_t launch_interp(int mode, const float *field, int B, int h, int w, int pitch, size_t fstride, const float *y,
                         const float *x, size_t cstride, size_t n, float *out, hipStream_t st) {
    dim3 grid((unsigned)((n + 255) / 256), B), block(256);
    if (mode == 0) hipLaunchKernelGGL(k_interp<0>, grid, block, 0, st, field, h, w, pitch, fstride, y, x, cstride, n, out);
    else if (mode == 1) hipLaunchKernelGGL(k_interp<1>, grid, block, 0, st, field, h, w, pitch, fstride, y, x, cstride, n, out);
    else hipLaunchKernelGGL(k_interp<2>, grid, block, 0, st, field, h, w, pitch, fstride, y, x, cstride, n, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------- fractal constants (fractal_generator.py:12-51)
// torch.linspace fp32 CPU kernel: one rounding per element (FMA form), symmetric halves.
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
    float step = __fdiv_rn(end - start, (float)(steps - 1));
    return i < steps / 2 ? __fmaf_rn(step, (float)i, start) : __fmaf_rn(-step, (float)(steps - i - 1), end);
}

// out index [i][j] over [W][H] with X = x[i], Y = y[j] (meshgrid 'ij' of (x,y)); N = H = W.
__global__ void k_fractal_constants(int N, float *perlin, float *mandel, float *field) {
    int j = blockIdx.x * TX + threadIdx.x, i = blockIdx.y * TY + threadIdx.y;
    if (i >= N || j >= N) return;
    // perlin (:12-31): sum_{o<6} (amp*sin(f*x))*cos(f*y); (noise+1)/2
    float px = linspace_at(0.f, 10.f, N, i), py = linspace_at(0.f, 10.f, N, j);
    float noise = 0.f, amp = 1.f, freq = 1.f;
#pragma unroll
    for (int o = 0; o < 6; ++o) {
        float t = amp * sinf(freq * px);
        t = t * cosf(freq * py);
        noise = noise + t;
        amp *= 0.5f; freq *= 2.f;      // exact powers of two: same values as the python doubles
    }
    float per = __fdiv_rn(noise + 1.0f, 2.0f);
    // mandelbrot (:33-51): z <- z*z + c while |z| <= 2 (masked update == freeze after escape), count = last it
    float cr = linspace_at(-2.5f, 1.5f, N, i), ci = linspace_at(-1.5f, 1.5f, N, j);
    float zr = 0.f, zi = 0.f;
    int cnt = 0;
    for (int it = 0; it < 100; ++it) {
        float m = __fsqrt_rn(zr * zr + zi * zi);
        if (!(m <= 2.0f)) break;
        float rr = zr * zr - zi * zi;
        float ab = zr * zi;
        float ii = ab + ab;
        zr = rr + cr; zi = ii + ci;
        cnt = it;
    }
    float man = __fdiv_rn((float)cnt, 100.0f);
    size_t o = (size_t)i * N + j;
    perlin[o] = per;
    mandel[o] = man;
    field[o] = 0.7f * per + 0.3f * man;   // :58-59
}

hipError_t launch_fractal_constants(int N, float *perlin, float *mandel, float *field, hipStream_t st) {
    dim3 grid(cdiv(N, TX), cdiv(N, TY)), block(TX, TY);
    hipLaunchKernelGGL(k_fractal_constants, grid, block, 0, st, N, perlin, mandel, field);
    return hipGetLastError();
}

// field + (intensity*F)*field  (:62)
__global__ void k_apply_fractal(const float *in, float *out, const float *fractal, size_t per_field, size_t total, float fint) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t step = (size_t)gridDim.x * blockDim.x;
    for (; k < total; k += step) {
        float x = in[k];
        float t = fint * fractal[k % per_field];
        t = t * x;
        out[k] = x + t;
    }
}

hipError_t launch_apply_fractal(const float *in, float *out, const float *fractal, int n_fields, int N, float intensity,
                                hipStream_t st) {
    size_t per = (size_t)N * N, total = per * n_fields;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_apply_fractal, dim3(blocks), dim3(256), 0, st, in, out, fractal, per, total, intensity);
    return hipGetLastError();
}

}  // namespace smk
