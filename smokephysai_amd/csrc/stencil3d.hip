// 3-D stable-fluids stepper for gfx950 (BASELINE configs[4]: 512 x 512 x 64 grids, batch 8).
//
// Semantics: SPEC_3D.md -- the rule-by-rule generalisation of /root/reference/src/physics/navier_stokes.py:24-173 (the reference is 2-D
// only), executable form oracle/ns_nd.py.  Arithmetic contract as in stencil.hip: one fp32 rounding per listed operation, in the listed
// order (-ffp-contract=off), correctly rounded divide / sqrt, so every field is bit-identical to the oracle for identical inputs.
//
// Layout [B][planes][rows][pitch], x fastest, rows padded to 128-byte multiples.  Every kernel here is HBM-bound (0.2-0.3 flop per byte):
// one thread per cell (four cells per thread as 16-byte accesses in the Jacobi sweep, which is 60 of the step's 97 floats per cell at
// J = 20), 64 consecutive x per wavefront, z-neighbour planes served by the L2 / Infinity Cache (workgroups are issued plane-major, so
// the planes z-1, z, z+1 of a grid are in flight together).
#include "stencil3d.h"

#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace smk {

#define TX3 64
#define TY3 4


// ---------------------------------------------------------------- reset (navier_stokes.py:24-35)
__global__ void k3_zero(Geom3 g, State3 s, const uint8_t *mask) {
    const int b = blockIdx.z / (g.D + 1), z = blockIdx.z % (g.D + 1);
    if (mask && !mask[b]) return;
    const int x = blockIdx.x * TX3 + threadIdx.x, y = blockIdx.y * TY3 + threadIdx.y;
    if (z < g.D && y <= g.H && x < g.pc) s.u[b * g.su + ((size_t)z * (g.H + 1) + y) * g.pc + x] = 0.f;
    if (z < g.D && y < g.H && x < g.pv) s.v[b * g.sv + ((size_t)z * g.H + y) * g.pv + x] = 0.f;
    if (y < g.H && x < g.pc) s.w[b * g.sw + ((size_t)z * g.H + y) * g.pc + x] = 0.f;
    if (z < g.D && y < g.H && x < g.pc) {
        const size_t o = b * g.sc + ((size_t)z * g.H + y) * g.pc + x;
        s.p[o] = 0.f;
        s.d[o] = 0.f;
    }
}

hipError_t launch3_zero(const Geom3 &g, State3 s, const uint8_t *dev_mask, hipStream_t st) {
    const int maxp = g.pc > g.pv ? g.pc : g.pv;
    dim3 grid(cdiv(maxp, TX3), cdiv(g.H + 1, TY3), g.B * (g.D + 1)), block(TX3, TY3);
    hipLaunchKernelGGL(k3_zero, grid, block, 0, st, g, s, dev_mask);
    return hipGetLastError();
}

// ---------------------------------------------------------------- sources (SPEC_3D.md section 2; navier_stokes.py:37-48)
__global__ void k3_add_sources(Geom3 g, float *density, const Src3Dev *src, const int *first) {
    const int b = blockIdx.z / g.D, z = blockIdx.z % g.D;
    const int s0 = first[b], s1 = first[b + 1];
    if (s0 == s1) return;
    const int x = blockIdx.x * TX3 + threadIdx.x, y = blockIdx.y * TY3 + threadIdx.y;
    if (y >= g.H || x >= g.W) return;
    float *cell = density + b * g.sc + ((size_t)z * g.H + y) * g.pc + x;
    float d = *cell;
    for (int s = s0; s < s1; ++s) {
        const Src3Dev q = src[s];
        const int dx = x - q.x, dy = y - q.y, dz = z - q.z;
        const float dist = __fsqrt_rn((float)(dx * dx + dy * dy + dz * dz));
        if (dist <= (float)q.radius) {
            const float e = expf(-__fdiv_rn(dist * dist, q.denom));
            d = d + q.fint * e;
        }
    }
    *cell = d;
}

hipError_t launch3_add_sources(const Geom3 &g, float *density, const Src3Dev *src, const int *first, hipStream_t st) {
    dim3 grid(cdiv(g.W, TX3), cdiv(g.H, TY3), g.B * g.D), block(TX3, TY3);
    hipLaunchKernelGGL(k3_add_sources, grid, block, 0, st, g, density, src, first);
    return hipGetLastError();
}

// ---------------------------------------------------------------- buoyancy + diffusion (SPEC_3D.md sections 3, 6.1-6.2)
// out = c + coef * ((((((up + down) + left) + right) + front) + back) - 6 c), replicate padding; val(z, y, x) reads the field
template <class F>
__device__ __forceinline__ float diffuse3_at(F val, int Dd, int R, int C, int z, int y, int x, float coef) {
    const int yu = y > 0 ? y - 1 : 0, yd = y < R - 1 ? y + 1 : R - 1;
    const int xl = x > 0 ? x - 1 : 0, xr = x < C - 1 ? x + 1 : C - 1;
    const int zf = z > 0 ? z - 1 : 0, zb = z < Dd - 1 ? z + 1 : Dd - 1;
    const float c = val(z, y, x);
    float lap = val(z, yu, x) + val(z, yd, x);
    lap = lap + val(z, y, xl);
    lap = lap + val(z, y, xr);
    lap = lap + val(zf, y, x);
    lap = lap + val(zb, y, x);
    lap = lap - 6.0f * c;
    return c + coef * lap;
}

__global__ void k3_buoy_diffuse(Geom3 g, State3 in, State3 out) {
    const int b = blockIdx.z / (g.D + 1), z = blockIdx.z % (g.D + 1);
    const int x = blockIdx.x * TX3 + threadIdx.x, y = blockIdx.y * TY3 + threadIdx.y;
    const int D = g.D, H = g.H, W = g.W;
    const float *u = in.u + b * g.su, *v = in.v + b * g.sv, *w = in.w + b * g.sw, *d = in.d + b * g.sc;
    if (z < D && y <= H && x < W) {
        auto val = [&](int k, int i, int j) { return u[(k * (H + 1) + i) * g.pc + j]; };
        out.u[b * g.su + (z * (H + 1) + y) * g.pc + x] = diffuse3_at(val, D, H + 1, W, z, y, x, g.coef_uv);
    }
    if (z < D && y < H && x <= W) {
        // the buoyancy-updated v that diffusion_step(v) sees: v[:, :, :-1] += dt * (density * 0.1)
        auto val = [&](int k, int i, int j) {
            float t = v[(k * H + i) * g.pv + j];
            if (j < W) {
                const float bb = d[(k * H + i) * g.pc + j] * 0.1f;
                t = t + g.dt * bb;
            }
            return t;
        };
        out.v[b * g.sv + (z * H + y) * g.pv + x] = diffuse3_at(val, D, H, W + 1, z, y, x, g.coef_uv);
    }
    if (y < H && x < W) {
        auto val = [&](int k, int i, int j) { return w[(k * H + i) * g.pc + j]; };
        out.w[b * g.sw + (z * H + y) * g.pc + x] = diffuse3_at(val, D + 1, H, W, z, y, x, g.coef_uv);
    }
    if (z < D && y < H && x < W) {
        auto val = [&](int k, int i, int j) { return d[(k * H + i) * g.pc + j]; };
        out.d[b * g.sc + (z * H + y) * g.pc + x] = diffuse3_at(val, D, H, W, z, y, x, g.coef_d);
    }
}

// The same stage marching along z: a thread owns the column (y, x) of all four fields and keeps each field's values at planes z-1, z,
// z+1 in registers, so the front / back neighbours and the centre cost one new load per field and plane; the four lateral neighbours are
// same-plane loads that the neighbouring threads' own centre loads bring into the L1 (the one-cell-per-thread form above fetches every
// z-neighbour through the L2: ~5 x the compulsory bytes).  v's extra column x = W rides on the thread of x = W-1; u's extra row y = H and
// w's extra plane z = D are ordinary iterations.  Per cell the expression trees of diffuse3_at -- bit-identical.
template <int TXB, int TYB>
__global__ __launch_bounds__(TXB * TYB) void k3_buoy_diffuse_march(Geom3 g, State3 in, State3 out) {
    const int D = g.D, H = g.H, W = g.W;
    // tiles in (x fastest, y, grid) order, a contiguous range per XCD: the x / y neighbours whose edge lines this tile reads share its L2
    const unsigned ntx = (W + TXB - 1) / TXB, nty = (H + TYB) / TYB;
    unsigned t = xcd_contiguous(blockIdx.x, gridDim.x);
    const int bx = t % ntx; t /= ntx;
    const int by = t % nty;
    const int b = t / nty;
    const int x = bx * TXB + threadIdx.x, y = by * TYB + threadIdx.y;
    if (x >= W || y > H) return;
    const float *u = in.u + b * g.su, *v = in.v + b * g.sv, *w = in.w + b * g.sw, *d = in.d + b * g.sc;
    float *uo = out.u + b * g.su, *vo = out.v + b * g.sv, *wo = out.w + b * g.sw, *dd = out.d + b * g.sc;
    const bool row = y < H;                                   // v, w, density exist on rows 0 .. H-1; u also on row H
    const bool extra = row && x == W - 1;                     // this thread also forms v(:, y, W)
    const int pu = (H + 1) * g.pc, pvs = H * g.pv, pw = H * g.pc;      // plane strides
    // lateral index sets (replicate padding)
    const int yu_u = y > 0 ? y - 1 : 0, yd_u = y < H ? y + 1 : H;                       // u: rows 0 .. H
    const int yu = y > 0 ? y - 1 : 0, yd = y < H - 1 ? y + 1 : H - 1;                   // v, w, d: rows 0 .. H-1
    const int xl = x > 0 ? x - 1 : 0, xr = x < W - 1 ? x + 1 : W - 1;                   // u, w, d: columns 0 .. W-1
    const int xr_v = x + 1;                                                             // v: columns 0 .. W (x <= W-1 here)
    auto vb = [&](int k, int i, int j) {                      // the buoyancy-updated v that diffusion_step(v) sees
        float t = v[k * pvs + i * g.pv + j];
        if (j < W) {
            const float bb = d[k * pw + i * g.pc + j] * 0.1f;
            t = t + g.dt * bb;
        }
        return t;
    };
    auto finish = [](float c, float up, float dn, float lf, float rt, float fr, float bk, float coef) {
        float lap = up + dn;
        lap = lap + lf;
        lap = lap + rt;
        lap = lap + fr;
        lap = lap + bk;
        lap = lap - 6.0f * c;
        return c + coef * lap;
    };
    // planes z-1 (replicated at z = 0), z, z+1
    float u_c = u[y * g.pc + x], u_m = u_c, u_p;
    float v_c = row ? vb(0, y, x) : 0.f, v_m = v_c, v_p;
    float e_c = extra ? v[y * g.pv + W] : 0.f, e_m = e_c, e_p;               // v at column W: no buoyancy there
    float w_c = row ? w[y * g.pc + x] : 0.f, w_m = w_c, w_p;
    float d_c = row ? d[y * g.pc + x] : 0.f, d_m = d_c, d_p;
    for (int z = 0; z <= D; ++z) {
        const int zu = z + 1 < D ? z + 1 : D - 1;             // next plane of the D-plane fields (replicated at the end)
        if (z < D) {
            u_p = u[zu * pu + y * g.pc + x];
            if (row) {
                v_p = vb(zu, y, x);
                d_p = d[zu * pw + y * g.pc + x];
                if (extra) e_p = v[zu * pvs + y * g.pv + W];
            }
        }
        if (row) w_p = w[(z + 1 <= D ? z + 1 : D) * pw + y * g.pc + x];
        if (z < D) {
            const float *uz = u + z * pu;
            uo[z * pu + y * g.pc + x] = finish(u_c, uz[yu_u * g.pc + x], uz[yd_u * g.pc + x], uz[y * g.pc + xl], uz[y * g.pc + xr], u_m, u_p, g.coef_uv);
            if (row) {
                vo[z * pvs + y * g.pv + x] = finish(v_c, vb(z, yu, x), vb(z, yd, x), vb(z, y, xl), vb(z, y, xr_v), v_m, v_p, g.coef_uv);
                if (extra)          // column W: right neighbour = itself (replicate), left neighbour = this thread's own v_c
                    vo[z * pvs + y * g.pv + W] = finish(e_c, v[z * pvs + yu * g.pv + W], v[z * pvs + yd * g.pv + W], v_c, e_c, e_m, e_p, g.coef_uv);
                const float *dz = d + z * pw;
                dd[z * pw + y * g.pc + x] = finish(d_c, dz[yu * g.pc + x], dz[yd * g.pc + x], dz[y * g.pc + xl], dz[y * g.pc + xr], d_m, d_p, g.coef_d);
            }
        }
        if (row) {
            const float *wz = w + z * pw;
            wo[z * pw + y * g.pc + x] = finish(w_c, wz[yu * g.pc + x], wz[yd * g.pc + x], wz[y * g.pc + xl], wz[y * g.pc + xr], w_m, w_p, g.coef_uv);
        }
        u_m = u_c; u_c = u_p;
        v_m = v_c; v_c = v_p; e_m = e_c; e_c = e_p;
        d_m = d_c; d_c = d_p;
        w_m = w_c; w_c = w_p;
    }
}

hipError_t launch3_buoy_diffuse(const Geom3 &g, State3 in, State3 out, hipStream_t st) {
    static const bool cellwise = getenv("SMK_DIFFUSE3_CELLWISE") != nullptr;      // diagnostic: the one-cell-per-thread form
    if (!cellwise && g.B <= 65535) {
        constexpr int TXB = 64, TYB = 4;
        hipLaunchKernelGGL((k3_buoy_diffuse_march<TXB, TYB>), dim3((unsigned)(cdiv(g.W, TXB) * cdiv(g.H + 1, TYB) * g.B)), dim3(TXB, TYB), 0, st, g, in, out);
        return hipGetLastError();
    }
    dim3 grid(cdiv(g.W + 1, TX3), cdiv(g.H + 1, TY3), g.B * (g.D + 1)), block(TX3, TY3);
    hipLaunchKernelGGL(k3_buoy_diffuse, grid, block, 0, st, g, in, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------- divergence (SPEC_3D.md section 4)
// div = (((((u[y+1] - u[y]) + v[x+1]) - v[x]) + w[z+1]) - w[z]) / dt
__global__ void k3_divergence(Geom3 g, State3 s, float *div) {
    const int b = blockIdx.z / g.D, z = blockIdx.z % g.D;
    const int x = blockIdx.x * TX3 + threadIdx.x, y = blockIdx.y * TY3 + threadIdx.y;
    if (y >= g.H || x >= g.W) return;
    const float *u = s.u + b * g.su + ((size_t)z * (g.H + 1) + y) * g.pc + x;
    const float *v = s.v + b * g.sv + ((size_t)z * g.H + y) * g.pv + x;
    const float *w = s.w + b * g.sw + ((size_t)z * g.H + y) * g.pc + x;
    float a = u[g.pc] - u[0];
    a = a + v[1];
    a = a - v[0];
    a = a + w[(size_t)g.H * g.pc];
    a = a - w[0];
    div[b * g.sc + ((size_t)z * g.H + y) * g.pc + x] = __fdiv_rn(a, g.dt);
}

hipError_t launch3_divergence(const Geom3 &g, State3 s, float *div, hipStream_t st) {
    dim3 grid(cdiv(g.W, TX3), cdiv(g.H, TY3), g.B * g.D), block(TX3, TY3);
    hipLaunchKernelGGL(k3_divergence, grid, block, 0, st, g, s, div);
    return hipGetLastError();
}

// ---------------------------------------------------------------- Jacobi sweep (SPEC_3D.md section 4)
// p_new = 0 on the boundary shell; interior: (1/6) * ((((((up + down) + left) + right) + front) + back) - div)
__global__ void k3_jacobi(Geom3 g, const float *__restrict__ p, float *__restrict__ pn, const float *__restrict__ div) {
    const int b = blockIdx.z / g.D, z = blockIdx.z % g.D;
    const int x = blockIdx.x * TX3 + threadIdx.x, y = blockIdx.y * TY3 + threadIdx.y;
    if (y >= g.H || x >= g.W) return;
    const size_t o = b * g.sc + ((size_t)z * g.H + y) * g.pc + x, ps = (size_t)g.H * g.pc;
    float r = 0.f;
    if (z > 0 && z < g.D - 1 && y > 0 && y < g.H - 1 && x > 0 && x < g.W - 1) {
        float s = p[o - g.pc] + p[o + g.pc];
        s = s + p[o - 1];
        s = s + p[o + 1];
        s = s + p[o - ps];
        s = s + p[o + ps];
        s = s - div[o];
        r = g.sixth * s;
    }
    pn[o] = r;
}

// Four cells per thread (W % 4 == 0): rows as 16-byte accesses, the two x-neighbours outside the quad as scalar loads (same lines)
__global__ void k3_jacobi4(Geom3 g, const float *__restrict__ p, float *__restrict__ pn, const float *__restrict__ div) {
    const int b = blockIdx.z / g.D, z = blockIdx.z % g.D;
    const int x = (blockIdx.x * TX3 + threadIdx.x) * 4, y = blockIdx.y * TY3 + threadIdx.y;
    if (y >= g.H || x >= g.W) return;
    const size_t o = b * g.sc + ((size_t)z * g.H + y) * g.pc + x, ps = (size_t)g.H * g.pc;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (z > 0 && z < g.D - 1 && y > 0 && y < g.H - 1) {
        const float4 up = *reinterpret_cast<const float4 *>(p + o - g.pc), dn = *reinterpret_cast<const float4 *>(p + o + g.pc);
        const float4 fr = *reinterpret_cast<const float4 *>(p + o - ps), bk = *reinterpret_cast<const float4 *>(p + o + ps);
        const float4 c = *reinterpret_cast<const float4 *>(p + o), dv = *reinterpret_cast<const float4 *>(div + o);
        const float lft = x > 0 ? p[o - 1] : 0.f, rgt = x + 4 < g.W ? p[o + 4] : 0.f;
        auto cell = [&](float u_, float d_, float l_, float r_, float f_, float b_, float dv_) {
            float s = u_ + d_;
            s = s + l_;
            s = s + r_;
            s = s + f_;
            s = s + b_;
            s = s - dv_;
            return g.sixth * s;
        };
        r.x = x > 0 ? cell(up.x, dn.x, lft, c.y, fr.x, bk.x, dv.x) : 0.f;
        r.y = cell(up.y, dn.y, c.x, c.z, fr.y, bk.y, dv.y);
        r.z = cell(up.z, dn.z, c.y, c.w, fr.z, bk.z, dv.z);
        r.w = x + 4 < g.W ? cell(up.w, dn.w, c.z, rgt, fr.w, bk.w, dv.w) : 0.f;
    }
    *reinterpret_cast<float4 *>(pn + o) = r;
}

// T sweeps per launch (temporal blocking, 2.5-D): a workgroup owns a (y, x) tile of the grid plus a T-cell halo on each side, CY columns
// per thread, and marches along z.  Level l (0 = the input p, l = the result of sweep l) is kept per column as the values at three
// consecutive planes in registers; in iteration z sweep s = 1 .. T forms plane z - s + 1 from level s-1 (planes z-s and z-s+2 of its own
// column from registers -- the latter is what sweep s-1 produced a moment ago, or the prefetched input plane -- and the four lateral
// neighbours of plane z-s+1 from LDS, where every thread put its value at the top of the iteration).  Only sweep T's result goes to
// HBM.  The LDS planes are double-buffered: one workgroup barrier per plane.  Per output cell and per T sweeps the launch moves
// (2 reads x tile / inner + 1 write) floats instead of 3 T; every cell's value is k3_jacobi's expression applied T times --
// bit-identical.  Halo threads compute values nobody uses (their lateral reads are clamped into the block); cells on the grid's shell
// are 0 after every sweep, cells outside the grid are never stored.
template <int T, int TXB, int TYB, int CY>
__global__ __launch_bounds__(TXB * TYB) void k3_jacobi_xt(Geom3 g, const float *__restrict__ p, float *__restrict__ pn, const float *__restrict__ div) {
    constexpr int ROWS = TYB * CY;
    __shared__ float L[T][2][ROWS][TXB];
    const int tx = threadIdx.x;
    // tiles in (x fastest, y, grid) order, a contiguous range per XCD: overlapping halos are fetched into ONE L2
    const unsigned ntx = (g.W + (TXB - 2 * T) - 1) / (TXB - 2 * T), nty = (g.H + (ROWS - 2 * T) - 1) / (ROWS - 2 * T);
    unsigned tile = xcd_contiguous(blockIdx.x, gridDim.x);
    const int bx = tile % ntx; tile /= ntx;
    const int by = tile % nty;
    const int b = tile / nty;
    const int x = bx * (TXB - 2 * T) - T + tx;
    const bool x_in = x >= 0 && x < g.W;
    const int txl = tx > 0 ? tx - 1 : 0, txr = tx < TXB - 1 ? tx + 1 : TXB - 1;
    const size_t ps = (size_t)g.H * g.pc;
    int ty[CY], tyu[CY], tyd[CY];
    bool inside[CY], lshell[CY], outc[CY];
    size_t o[CY];
    float lm[T][CY], lc[T][CY];       // level l at planes z-l-1 and z-l
    float dv[T][CY];                  // div at planes z .. z-T+1
    float a_p[CY];                    // input at plane z+1
#pragma unroll
    for (int c = 0; c < CY; ++c) {
        ty[c] = threadIdx.y + c * TYB;
        tyu[c] = ty[c] > 0 ? ty[c] - 1 : 0;
        tyd[c] = ty[c] < ROWS - 1 ? ty[c] + 1 : ROWS - 1;
        const int y = by * (ROWS - 2 * T) - T + ty[c];
        inside[c] = x_in && y >= 0 && y < g.H;
        lshell[c] = x <= 0 || x >= g.W - 1 || y <= 0 || y >= g.H - 1;      // (also true outside the grid)
        outc[c] = inside[c] && tx >= T && tx < TXB - T && ty[c] >= T && ty[c] < ROWS - T;
        o[c] = b * g.sc + (size_t)(inside[c] ? y : 0) * g.pc + (inside[c] ? x : 0);
#pragma unroll
        for (int l = 0; l < T; ++l) { lm[l][c] = 0.f; lc[l][c] = 0.f; dv[l][c] = 0.f; }
        lc[0][c] = inside[c] ? p[o[c]] : 0.f;
        a_p[c] = inside[c] && g.D > 1 ? p[o[c] + ps] : 0.f;
        dv[0][c] = inside[c] ? div[o[c]] : 0.f;
    }
    for (int z = 0; z < g.D + T - 1; ++z) {
        float a_next[CY], d_next[CY];
#pragma unroll
        for (int c = 0; c < CY; ++c) {      // prefetch plane z+2 of the input and z+1 of div while this plane is worked on
            a_next[c] = (inside[c] && z + 2 < g.D) ? p[o[c] + (size_t)(z + 2) * ps] : 0.f;
            d_next[c] = (inside[c] && z + 1 < g.D) ? div[o[c] + (size_t)(z + 1) * ps] : 0.f;
#pragma unroll
            for (int l = 0; l < T; ++l) L[l][z & 1][ty[c]][tx] = lc[l][c];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CY; ++c) {
            float up = a_p[c];                                // level s-1 at plane z-s+2, starting with the input at z+1
#pragma unroll
            for (int s = 1; s <= T; ++s) {
                const int zp = z - s + 1;                     // the plane sweep s forms now
                float r = 0.f;
                if (zp >= 1 && zp < g.D - 1 && !lshell[c]) {
                    const float (*Lp)[TXB] = L[s - 1][z & 1];
                    float sm = Lp[tyu[c]][tx] + Lp[tyd[c]][tx];
                    sm = sm + Lp[ty[c]][txl];
                    sm = sm + Lp[ty[c]][txr];
                    sm = sm + lm[s - 1][c];
                    sm = sm + up;
                    sm = sm - dv[s - 1][c];
                    r = g.sixth * sm;
                }
                // level s-1 moves one plane on; `up` for the next sweep is this sweep's fresh value (level s at plane z-s+1)
                lm[s - 1][c] = lc[s - 1][c];
                lc[s - 1][c] = up;
                up = r;
            }
            if (z >= T - 1 && outc[c]) pn[o[c] + (size_t)(z - T + 1) * ps] = up;
#pragma unroll
            for (int l = T - 1; l > 0; --l) dv[l][c] = dv[l - 1][c];
            dv[0][c] = d_next[c];
            a_p[c] = a_next[c];
        }
    }
}

// The same temporal blocking with FOUR consecutive x per thread (W % 4 == 0): a 512-thread workgroup owns a 64 x 32 tile (16 lanes x 4 cells
// per row; halo 4 = one lane / four rows on every side, whatever T <= 4), levels as float4 registers.  A cell's x-neighbours are the
// thread's own registers except at the quad's two ends, its y-neighbours two 16-byte LDS reads per quad: per quad and sweep 1 ds_write_b128
// + 2 ds_read_b128 + 2 ds_read_b32 instead of 4 x (1 write + 4 reads), a third of the vector instructions per cell (the scalar form spends
// more on addresses and selects than on the seven flops), 16-byte global loads and stores.  Expression per cell = k3_jacobi's.
// PD: planes of p and div requested ahead of their use (2: a third plane in flight costs more in registers than it hides).
// hipcc notes: a 16-byte load or LDS read under a condition (`ok ? *q : zero`, `shell ? 0 : cell(...)`) is scalarised into per-component
// loads under exec-mask branches (64 dword loads, 92 branches, 4-way bank conflicts on the dword LDS reads) -- every load here is
// unconditional on a clamped address and the selects follow.
template <int T, int PD = 3>
__global__ __launch_bounds__(512, 2) void k3_jacobi_v4(Geom3 g, const float *__restrict__ p, float *__restrict__ pn, const float *__restrict__ div) {
    constexpr int LX = 16, ROWS = 32, HALO = 4, UX = 4 * LX - 2 * HALO, UY = ROWS - 2 * HALO;      // useful 56 x 24 of the 64 x 32 tile
    static_assert(T >= 1 && T <= HALO, "the halo covers T sweeps");
    // (unpadded 256-byte rows: conflict-free for the lane groups of ds_read_b128 / ds_write_b128 -- a one-quad pad made them 2-way conflicts)
    __shared__ float4 L[T][2][ROWS][LX];                                                          // T = 4: 64 KB
    const int lx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const unsigned ntx = (g.W + UX - 1) / UX, nty = (g.H + UY - 1) / UY;
    unsigned tile = xcd_contiguous(blockIdx.x, gridDim.x);
    const int bx = tile % ntx; tile /= ntx;
    const int by = tile % nty;
    const int b = tile / nty;
    const int x0 = bx * UX - HALO + 4 * lx, y = by * UY - HALO + ty;                              // this thread's cells: (y, x0 .. x0 + 3)
    const bool inside = y >= 0 && y < g.H && x0 >= 0 && x0 + 3 < g.W;                             // (W % 4 == 0: a quad is inside or outside as a whole)
    const bool rowshell = y <= 0 || y >= g.H - 1;
    const bool sh0 = rowshell || x0 <= 0 || !inside, sh3 = rowshell || x0 + 3 >= g.W - 1 || !inside, sh12 = rowshell || !inside;
    const bool outc = inside && lx >= 1 && lx < LX - 1 && ty >= HALO && ty < ROWS - HALO;
    const int tyu = ty > 0 ? ty - 1 : 0, tyd = ty < ROWS - 1 ? ty + 1 : ROWS - 1;
    const size_t ps = (size_t)g.H * g.pc;
    const size_t o = b * g.sc + (size_t)(inside ? y : 0) * g.pc + (inside ? x0 : 0);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // every load is unconditional (the address of a quad outside the grid is the tile's first in-grid quad, a plane past the end is the
    // last plane) and the select follows it: a load under a condition is scalarised into per-component branches by hipcc
    auto ld4 = [&](const float *base, int plane, bool ok) {
        const float4 v = *reinterpret_cast<const float4 *>(base + (size_t)(plane < g.D ? plane : g.D - 1) * ps);
        return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    };
    float4 lm[T], lc[T], dv[T];                          // level l at planes z-l-1 and z-l; div at planes z .. z-T+1
#pragma unroll
    for (int l = 0; l < T; ++l) { lm[l] = zero4; lc[l] = zero4; dv[l] = zero4; }
    lc[0] = ld4(p + o, 0, inside);
    dv[0] = ld4(div + o, 0, inside);
    float4 aq[PD], dq[PD];                               // requested: p and div at planes z+1 .. z+PD
#pragma unroll
    for (int i = 0; i < PD; ++i) {
        aq[i] = ld4(p + o, 1 + i, inside && 1 + i < g.D);
        dq[i] = ld4(div + o, 1 + i, inside && 1 + i < g.D);
    }
    const float sixth = g.sixth;
    for (int z = 0; z < g.D + T - 1; ++z) {
        const float4 a_new = ld4(p + o, z + 1 + PD, inside && z + 1 + PD < g.D);
        const float4 d_new = ld4(div + o, z + 1 + PD, inside && z + 1 + PD < g.D);
        const float4 a_p = aq[0];                        // input at plane z+1
#pragma unroll
        for (int l = 0; l < T; ++l) L[l][z & 1][ty][lx] = lc[l];
        __syncthreads();
        float4 up = a_p;                                 // level s-1 at plane z-s+2, starting with the input at z+1
#pragma unroll
        for (int s = 1; s <= T; ++s) {
            const int zp = z - s + 1;                    // the plane sweep s forms now
            float4 r = zero4;
            {
                const bool zin = zp >= 1 && zp < g.D - 1;        // (wave-uniform; as a select below: the LDS reads stay unconditional 16-byte reads)
                const float4 (*Lp)[LX] = L[s - 1][z & 1];
                const float4 u4 = Lp[tyu][lx], d4 = Lp[tyd][lx], c4 = lc[s - 1];
                // the quads to the left / right are the neighbouring lanes of this row of 16: DPP row shifts of the registers (the 4-byte LDS
                // reads at a 16-byte stride they replace were 4-way bank conflicts: 2/3 of the kernel's LDS cycles).  Lanes 0 / 15 of a row
                // receive 0: tile-halo cells, whose values nobody uses.
                const float lf = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c4.w), 0x111, 0xf, 0xf, true));
                const float rt = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c4.x), 0x101, 0xf, 0xf, true));
                auto cell = [&](float u_, float d_, float l_, float r_, float f_, float b_, float dv_) {
                    float sm = u_ + d_;
                    sm = sm + l_;
                    sm = sm + r_;
                    sm = sm + f_;
                    sm = sm + b_;
                    sm = sm - dv_;
                    return sixth * sm;
                };
                const float c0 = cell(u4.x, d4.x, lf, c4.y, lm[s - 1].x, up.x, dv[s - 1].x);
                const float c1 = cell(u4.y, d4.y, c4.x, c4.z, lm[s - 1].y, up.y, dv[s - 1].y);
                const float c2 = cell(u4.z, d4.z, c4.y, c4.w, lm[s - 1].z, up.z, dv[s - 1].z);
                const float c3 = cell(u4.w, d4.w, c4.z, rt, lm[s - 1].w, up.w, dv[s - 1].w);
                r.x = (sh0 || !zin) ? 0.f : c0;
                r.y = (sh12 || !zin) ? 0.f : c1;
                r.z = (sh12 || !zin) ? 0.f : c2;
                r.w = (sh3 || !zin) ? 0.f : c3;
            }
            lm[s - 1] = lc[s - 1];
            lc[s - 1] = up;
            up = r;
        }
        if (z >= T - 1 && outc) *reinterpret_cast<float4 *>(pn + o + (size_t)(z - T + 1) * ps) = up;
#pragma unroll
        for (int l = T - 1; l > 0; --l) dv[l] = dv[l - 1];
        dv[0] = dq[0];
#pragma unroll
        for (int i = 0; i + 1 < PD; ++i) { aq[i] = aq[i + 1]; dq[i] = dq[i + 1]; }
        aq[PD - 1] = a_new;
        dq[PD - 1] = d_new;
    }
}

template <int T>
static void launch3_jacobi_v4(const Geom3 &g, const float *cur, float *nxt, const float *div, hipStream_t st) {
    const unsigned nb = (unsigned)(cdiv(g.W, 56) * cdiv(g.H, 24) * g.B);
    static const int pd = [] { const char *e = getenv("SMK_JACOBI3_PD"); return e ? atoi(e) : 2; }();     // planes requested ahead (measured per configs[4] step: 1 -> 6.00 ms, 2 -> 5.98, 3 -> 6.70; the one-cell-per-thread kernel 6.50)
    if (pd == 1) hipLaunchKernelGGL((k3_jacobi_v4<T, 1>), dim3(nb), dim3(512), 0, st, g, cur, nxt, div);
    else if (pd == 2) hipLaunchKernelGGL((k3_jacobi_v4<T, 2>), dim3(nb), dim3(512), 0, st, g, cur, nxt, div);
    else hipLaunchKernelGGL((k3_jacobi_v4<T, 3>), dim3(nb), dim3(512), 0, st, g, cur, nxt, div);
}

template <int T, int TXB, int TYB, int CY>
static void launch3_jacobi_xt(const Geom3 &g, const float *cur, float *nxt, const float *div, hipStream_t st) {
    dim3 block(TXB, TYB), grid((unsigned)(cdiv(g.W, TXB - 2 * T) * cdiv(g.H, TYB * CY - 2 * T) * g.B));
    hipLaunchKernelGGL((k3_jacobi_xt<T, TXB, TYB, CY>), grid, block, 0, st, g, cur, nxt, div);
}

hipError_t launch3_jacobi(const Geom3 &g, float *p, float *p2, float *p3, const float *div, int iters, hipStream_t st) {
    const bool vec = g.W % 4 == 0 && g.pc % 4 == 0 && getenv("SMK_JACOBI3_SCALAR") == nullptr;
    dim3 block(TX3, TY3), grid(cdiv(vec ? g.W / 4 : g.W, TX3), cdiv(g.H, TY3), g.B * g.D);
    // temporally blocked launches first (SMK_JACOBI3_T = 1, 2, 4 caps the sweeps per launch; default 4), single sweeps for the rest
    static const int tmax = [] { const char *e = getenv("SMK_JACOBI3_T"); return e ? atoi(e) : 4; }();
    // four cells per thread (k3_jacobi_v4) where rows are whole quads; SMK_JACOBI3_QUAD=0 keeps the one-cell-per-thread blocked kernel
    static const int quad_env = [] { const char *e = getenv("SMK_JACOBI3_QUAD"); return e ? atoi(e) : 1; }();      // 0 never, 1 always, 2 for the 2-sweep launches only
    const bool quad_ok = vec && (((uintptr_t)p | (uintptr_t)p2 | (uintptr_t)p3 | (uintptr_t)div) & 15) == 0 && g.sc % 4 == 0;
    const bool quad = quad_ok && quad_env == 1, quad2 = quad_ok && quad_env >= 1;
    // the launch plan: sweeps per launch.  The result must end in p without a copy.  Two buffers ping-pong, so an even launch count does;
    // an odd count of three or more goes p -> p2 -> p3 -> p2 -> ... -> p through the third buffer (J = 20: five 4-sweep launches); without a
    // third buffer one 4-sweep launch is traded for two 2-sweep ones (round 3's plan: 4 x 4 + 2 x 2)
    int plan[64], n = 0, left = iters;
    const bool blocked = g.B <= 65535;
    int n4 = (blocked && tmax >= 4) ? left / 4 : 0;
    if (n4 > 60) n4 = 60;
    const bool third = p3 != nullptr && getenv("SMK_JACOBI3_THIRD") == nullptr;
    if (!third && left % 4 == 0 && (n4 & 1) && n4 * 4 == left) --n4;
    for (int k = 0; k < n4; ++k) plan[n++] = 4;
    left -= 4 * n4;
    if (blocked && tmax >= 2)
        for (; left >= 2 && n < 62; left -= 2) plan[n++] = 2;
    float *cur = p;
    auto target = [&](int i, int total) -> float * {            // where launch i of `total` writes
        if (i == total - 1 && cur != p) return p;
        if (third && (total & 1) && total >= 3) return cur == p2 ? p3 : p2;
        return cur == p ? p2 : p;
    };
    const int total = n + left;                                     // (the remaining `left` sweeps run one per launch)
    for (int i = 0; i < n; ++i) {
        float *nxt = target(i, total);
        if (plan[i] == 4) {
            if (quad) launch3_jacobi_v4<4>(g, cur, nxt, div, st);
            else launch3_jacobi_xt<4, 64, 16, 2>(g, cur, nxt, div, st);
        } else {
            if (quad2) launch3_jacobi_v4<2>(g, cur, nxt, div, st);
            else launch3_jacobi_xt<2, 64, 16, 1>(g, cur, nxt, div, st);
        }
        cur = nxt;
    }
    for (int i = n; i < total; ++i) {
        float *nxt = target(i, total);
        if (vec) hipLaunchKernelGGL(k3_jacobi4, grid, block, 0, st, g, cur, nxt, div);
        else hipLaunchKernelGGL(k3_jacobi, grid, block, 0, st, g, cur, nxt, div);
        cur = nxt;
    }
    if (cur != p) {                                                 // (one launch in all, or an odd count without a third buffer)
        const hipError_t e = hipMemcpyAsync(p, cur, (size_t)g.B * g.sc * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------- gradient subtraction (SPEC_3D.md section 4)
// u[:,1:-1,:] -= dt (p[:,1:,:] - p[:,:-1,:]);  v[:,:,1:-1] -= dt (p[:,:,1:] - p[:,:,:-1]);  w[1:-1] -= dt (p[1:] - p[:-1])
__global__ void k3_grad_subtract(Geom3 g, State3 s, const float *p) {
    const int b = blockIdx.z / g.D, z = blockIdx.z % g.D;
    const int x = blockIdx.x * TX3 + threadIdx.x, y = blockIdx.y * TY3 + threadIdx.y;
    if (y >= g.H || x >= g.W) return;
    const float *pb = p + b * g.sc + ((size_t)z * g.H + y) * g.pc + x;
    const float pc = pb[0];
    if (y >= 1) {
        float *c = s.u + b * g.su + ((size_t)z * (g.H + 1) + y) * g.pc + x;
        const float gr = pc - pb[-g.pc];
        *c = *c - g.dt * gr;
    }
    if (x >= 1) {
        float *c = s.v + b * g.sv + ((size_t)z * g.H + y) * g.pv + x;
        const float gr = pc - pb[-1];
        *c = *c - g.dt * gr;
    }
    if (z >= 1) {
        float *c = s.w + b * g.sw + ((size_t)z * g.H + y) * g.pc + x;
        const float gr = pc - pb[-(ptrdiff_t)((size_t)g.H * g.pc)];
        *c = *c - g.dt * gr;
    }
}

hipError_t launch3_grad_subtract(const Geom3 &g, State3 s, const float *p, hipStream_t st) {
    dim3 grid(cdiv(g.W, TX3), cdiv(g.H, TY3), g.B * g.D), block(TX3, TY3);
    hipLaunchKernelGGL(k3_grad_subtract, grid, block, 0, st, g, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------- advection (SPEC_3D.md section 5)
// A velocity component sampled at a field's INTEGER index (z, y, x), shifted +0.5 along the axis it acts on, every coordinate clamped to
// the component's extent: the un-shifted axes put weight (i1 - c) = 1 on the low tap while the index is below the upper edge and 0 on it,
// the shifted axis gives 0.5 / 0.5 below the edge and lands exactly on the edge otherwise (both weights 0).  So the eight-term
// interpolation collapses EXACTLY to 0.5 c[lo] + 0.5 c[hi] when every index is <= extent - 2, and to 0 otherwise (the dropped terms are
// products with an exact zero weight: at most the sign of a zero, which index - dt * velocity cannot see).
template <int ACT>
__device__ __forceinline__ float vel3_at(const float *c, int Dc, int Hc, int Wc, int pitch, int z, int y, int x) {
    if (z > Dc - 2 || y > Hc - 2 || x > Wc - 2) return 0.f;
    const int o = (z * Hc + y) * pitch + x;                  // in-grid offsets fit 32 bits (smk_sim3d_create checks the extents)
    const int step = ACT == 2 ? 1 : (ACT == 1 ? pitch : Hc * pitch);
    return 0.5f * c[o] + 0.5f * c[o + step];
}

// interpolate(field, pz, py, px): floor -> indices clamped -> weights from the CLAMPED indices, (wx * wy) * wz, eight terms z-low plane
// first, x fastest, low before high
__device__ __forceinline__ float interp3(const float *f, int Df, int Hf, int Wf, int pitch, float pz, float py, float px) {
    int x0 = (int)floorf(px), y0 = (int)floorf(py), z0 = (int)floorf(pz);
    int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    x0 = clampi3(x0, 0, Wf - 1); x1 = clampi3(x1, 0, Wf - 1);
    y0 = clampi3(y0, 0, Hf - 1); y1 = clampi3(y1, 0, Hf - 1);
    z0 = clampi3(z0, 0, Df - 1); z1 = clampi3(z1, 0, Df - 1);
    const float wx0 = (float)x1 - px, wx1 = px - (float)x0;
    const float wy0 = (float)y1 - py, wy1 = py - (float)y0;
    const float wz0 = (float)z1 - pz, wz1 = pz - (float)z0;
    const int r00 = (z0 * Hf + y0) * pitch, r01 = (z0 * Hf + y1) * pitch;
    const int r10 = (z1 * Hf + y0) * pitch, r11 = (z1 * Hf + y1) * pitch;
    float acc = ((wx0 * wy0) * wz0) * f[r00 + x0];
    acc = acc + ((wx1 * wy0) * wz0) * f[r00 + x1];
    acc = acc + ((wx0 * wy1) * wz0) * f[r01 + x0];
    acc = acc + ((wx1 * wy1) * wz0) * f[r01 + x1];
    acc = acc + ((wx0 * wy0) * wz1) * f[r10 + x0];
    acc = acc + ((wx1 * wy0) * wz1) * f[r10 + x1];
    acc = acc + ((wx0 * wy1) * wz1) * f[r11 + x0];
    acc = acc + ((wx1 * wy1) * wz1) * f[r11 + x1];
    return acc;
}

// WHICH 0..3 = u, v, w, density: the field's own extents and pitch
template <int WHICH>
__global__ void k3_advect(Geom3 g, const float *__restrict__ field, float *__restrict__ out, const float *__restrict__ u,
                          const float *__restrict__ v, const float *__restrict__ w, float *__restrict__ frames, int64_t fsb) {
    const int Df = g.D + (WHICH == 2), Hf = g.H + (WHICH == 0), Wf = g.W + (WHICH == 1);
    const int pitch = WHICH == 1 ? g.pv : g.pc;
    const size_t fs = WHICH == 0 ? g.su : (WHICH == 1 ? g.sv : (WHICH == 2 ? g.sw : g.sc));
    const int b = blockIdx.z / Df, z = blockIdx.z % Df;
    const int x = blockIdx.x * TX3 + threadIdx.x, y = blockIdx.y * TY3 + threadIdx.y;
    if (y >= Hf || x >= Wf) return;
    const float ui = vel3_at<2>(u + b * g.su, g.D, g.H + 1, g.W, g.pc, z, y, x);
    const float vi = vel3_at<1>(v + b * g.sv, g.D, g.H, g.W + 1, g.pv, z, y, x);
    const float wi = vel3_at<0>(w + b * g.sw, g.D + 1, g.H, g.W, g.pc, z, y, x);
    const float tx = g.dt * ui, ty = g.dt * vi, tz = g.dt * wi;
    const float px = clampf3((float)x - tx, 0.f, (float)(Wf - 1));
    const float py = clampf3((float)y - ty, 0.f, (float)(Hf - 1));
    const float pz = clampf3((float)z - tz, 0.f, (float)(Df - 1));
    float r = interp3(field + b * fs, Df, Hf, Wf, pitch, pz, py, px);
    if (WHICH == 3) {
        r = r * 0.995f;                                       // navier_stokes.py:171
        if (frames) frames[(size_t)b * fsb + ((size_t)z * g.H + y) * g.W + x] = r;
    }
    out[b * fs + ((size_t)z * Hf + y) * pitch + x] = r;
}

hipError_t launch3_advect(const Geom3 &g, int which, const float *field, float *out, const float *u, const float *v, const float *w,
                          float *frames, int64_t fsb, hipStream_t st) {
    const int Df = g.D + (which == 2), Hf = g.H + (which == 0), Wf = g.W + (which == 1);
    dim3 grid(cdiv(Wf, TX3), cdiv(Hf, TY3), g.B * Df), block(TX3, TY3);
    switch (which) {
        case 0: hipLaunchKernelGGL(k3_advect<0>, grid, block, 0, st, g, field, out, u, v, w, frames, fsb); break;
        case 1: hipLaunchKernelGGL(k3_advect<1>, grid, block, 0, st, g, field, out, u, v, w, frames, fsb); break;
        case 2: hipLaunchKernelGGL(k3_advect<2>, grid, block, 0, st, g, field, out, u, v, w, frames, fsb); break;
        case 3: hipLaunchKernelGGL(k3_advect<3>, grid, block, 0, st, g, field, out, u, v, w, frames, fsb); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------- the four advections of a step as ONE launch
// u <- adv(u2; u2, v2, w2), v <- adv(v2; u, v2, w2), w <- adv(w2; u, v, w2), density <- adv(d2; u, v, w) * 0.995 depend on each other only
// through the velocity SAMPLING, which happens at integer cell indices where it collapses to 0.5 c[lo] + 0.5 c[hi] (vel3_at).  A
// workgroup owns a TZ x TY x TX tile of cells; it forms the advected u, v, w on the tile + 1 in z / y / x (what the later fields' samples
// reach: about 30 % redundant points at 8 x 8 x 32) into LDS, with a workgroup barrier between the fields, then the density and the frame.
// The displacement-dependent eight-tap gathers read u2, v2, w2, d2 -- inputs of the launch -- straight from global memory (L1 / L2: the
// taps of neighbouring cells coincide).  HBM traffic: four fields in (tile + halo), four fields + the frame out, instead of seventeen
// field passes; bit-identical to the four-launch form (same expressions per cell).  The extra row of u (y = H), column of v (x = W) and
// plane of w (z = D) belong to the last tile along that axis.
template <int TZ, int TY, int TX>
__global__ __launch_bounds__(512) void k3_advect_fused(Geom3 g, State3 in, State3 out, float *__restrict__ frames, int64_t fsb) {
    constexpr int EZ = TZ + 1, EY = TY + 1, EX = TX + 1, NE = EZ * EY * EX;
    __shared__ float Us[NE], Vs[NE], Ws[NE];
    const int D = g.D, H = g.H, W = g.W;
    const int ntx = (W + TX - 1) / TX, nty = (H + TY - 1) / TY, ntz = (D + TZ - 1) / TZ;
    int bid = blockIdx.x;
    const int tix = bid % ntx; bid /= ntx;
    const int tiy = bid % nty; bid /= nty;
    const int tiz = bid % ntz;
    const int b = bid / ntz;
    const int x0 = tix * TX, y0 = tiy * TY, z0 = tiz * TZ;
    const bool lastx = tix == ntx - 1, lasty = tiy == nty - 1, lastz = tiz == ntz - 1;
    const float *u2 = in.u + b * g.su, *v2 = in.v + b * g.sv, *w2 = in.w + b * g.sw, *d2 = in.d + b * g.sc;
    // velocity sample from an LDS tile holding the ADVECTED component: the rule of vel3_at in global indices, the values from the tile
    auto lds_vel = [&](const float *tile, int Dc, int Hc, int Wc, int step, int z, int y, int x) -> float {
        if (z > Dc - 2 || y > Hc - 2 || x > Wc - 2) return 0.f;
        const int o = ((z - z0) * EY + (y - y0)) * EX + (x - x0);
        return 0.5f * tile[o] + 0.5f * tile[o + step];
    };
    // one advected value: field f (extents Df, Hf, Wf, pitch) at (z, y, x) with the three velocity samples given
    auto advect_at = [&](const float *f, int Df, int Hf, int Wf, int pitch, int z, int y, int x, float ui, float vi, float wi) -> float {
        const float tx = g.dt * ui, ty = g.dt * vi, tz = g.dt * wi;
        const float px = clampf3((float)x - tx, 0.f, (float)(Wf - 1));
        const float py = clampf3((float)y - ty, 0.f, (float)(Hf - 1));
        const float pz = clampf3((float)z - tz, 0.f, (float)(Df - 1));
        return interp3(f, Df, Hf, Wf, pitch, pz, py, px);
    };
    // ---- u on the tile + 1 (extents D, H+1, W)
    for (int e = threadIdx.x; e < NE; e += 512) {
        const int ex = e % EX, ey = (e / EX) % EY, ez = e / (EX * EY);
        const int x = x0 + ex, y = y0 + ey, z = z0 + ez;
        float r = 0.f;
        if (z < D && y <= H && x < W) {
            const float ui = vel3_at<2>(u2, D, H + 1, W, g.pc, z, y, x), vi = vel3_at<1>(v2, D, H, W + 1, g.pv, z, y, x);
            const float wi = vel3_at<0>(w2, D + 1, H, W, g.pc, z, y, x);
            r = advect_at(u2, D, H + 1, W, g.pc, z, y, x, ui, vi, wi);
            if (ez < TZ && ex < TX && (ey < TY || lasty)) out.u[b * g.su + (z * (H + 1) + y) * g.pc + x] = r;
        }
        Us[e] = r;
    }
    __syncthreads();
    // ---- v on the tile + 1 in z, y (and the extra column x = W in the last x tile): extents D, H, W+1
    for (int e = threadIdx.x; e < NE; e += 512) {
        const int ex = e % EX, ey = (e / EX) % EY, ez = e / (EX * EY);
        const int x = x0 + ex, y = y0 + ey, z = z0 + ez;
        float r = 0.f;
        if (z < D && y < H && x <= W && (ex < TX || lastx)) {
            const float ui = lds_vel(Us, D, H + 1, W, 1, z, y, x), vi = vel3_at<1>(v2, D, H, W + 1, g.pv, z, y, x);
            const float wi = vel3_at<0>(w2, D + 1, H, W, g.pc, z, y, x);
            r = advect_at(v2, D, H, W + 1, g.pv, z, y, x, ui, vi, wi);
            if (ez < TZ && ey < TY) out.v[b * g.sv + (z * H + y) * g.pv + x] = r;
        }
        Vs[e] = r;
    }
    __syncthreads();
    // ---- w on the tile + 1 in z (the extra plane z = D in the last z tile): extents D+1, H, W
    for (int e = threadIdx.x; e < NE; e += 512) {
        const int ex = e % EX, ey = (e / EX) % EY, ez = e / (EX * EY);
        const int x = x0 + ex, y = y0 + ey, z = z0 + ez;
        float r = 0.f;
        if (ex < TX && ey < TY && z <= D && y < H && x < W) {
            const float ui = lds_vel(Us, D, H + 1, W, 1, z, y, x), vi = lds_vel(Vs, D, H, W + 1, EX, z, y, x);
            const float wi = vel3_at<0>(w2, D + 1, H, W, g.pc, z, y, x);
            r = advect_at(w2, D + 1, H, W, g.pc, z, y, x, ui, vi, wi);
            if (ez < TZ || lastz) out.w[b * g.sw + (z * H + y) * g.pc + x] = r;
        }
        Ws[e] = r;
    }
    __syncthreads();
    // ---- density on the tile (+ 0.995 decay, frame)
    for (int e = threadIdx.x; e < TZ * TY * TX; e += 512) {
        const int ex = e % TX, ey = (e / TX) % TY, ez = e / (TX * TY);
        const int x = x0 + ex, y = y0 + ey, z = z0 + ez;
        if (z >= D || y >= H || x >= W) continue;
        const float ui = lds_vel(Us, D, H + 1, W, 1, z, y, x), vi = lds_vel(Vs, D, H, W + 1, EX, z, y, x);
        const float wi = lds_vel(Ws, D + 1, H, W, EX * EY, z, y, x);
        float r = advect_at(d2, D, H, W, g.pc, z, y, x, ui, vi, wi);
        r = r * 0.995f;                                       // navier_stokes.py:171
        if (frames) frames[(size_t)b * fsb + (z * H + y) * W + x] = r;
        out.d[b * g.sc + (z * H + y) * g.pc + x] = r;
    }
}

hipError_t launch3_advect_fused(const Geom3 &g, State3 in, State3 out, float *frames, int64_t fsb, hipStream_t st) {
    constexpr int TZ = 8, TY = 8, TX = 32;
    const long long nb = (long long)cdiv(g.W, TX) * cdiv(g.H, TY) * cdiv(g.D, TZ) * g.B;
    if (nb > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k3_advect_fused<TZ, TY, TX>), dim3((unsigned)nb), dim3(512), 0, st, g, in, out, frames, fsb);
    return hipGetLastError();
}

}  // namespace smk
