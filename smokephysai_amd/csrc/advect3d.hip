// The 3-D stepper's advection launch (BASELINE configs[4]; SPEC_3D.md sections 4-6; generalises navier_stokes.py:74-131,148-149,166-171).
// A file of its own: built with -fno-slp-vectorize (packed fp32 pairs cost register moves here and issue at the scalar rate).
#include "stencil3d.h"

#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace smk {

// ---------------------------------------------------------------- the four advections (+ the gradient subtraction) marching along z
// Same arithmetic per cell as k3_advect<0..3> (and k3_grad_subtract when GRAD), organised for the memory system:
//   * a workgroup owns a TY x 64 column of cells and marches it through the planes k = 0 .. D.  The inputs u2, v2, w2, d2 live in LDS as
//     rings of three planes of the (TY+3) x 67 window (tile + 1 low, + 2 high; every out-of-grid element holds the CLAMPED in-grid value,
//     which is exactly what interp3's clamped indices read), the advected u, v, w as rings of two planes of the tile + 1.  Every input
//     element is fetched from HBM / L2 once per workgroup (1.44 x with the halo at TY = 8) instead of once per tap, 16 input bytes + 20
//     output bytes per cell;
//   * iteration k: Un(k) | barrier | Vn(k) | barrier | Wn(k) | barrier | Dn(k-1) + the LDS writes of the planes requested at the top of the
//     iteration (u2, v2, w2 plane k+2; d2 plane k+1 is written one phase later, behind the barrier that retires d2 plane k-2);
//   * with |dt * velocity| < 1 cell every back-trace lands on floor(p) in {i-1, i}: the eight taps are four `ds_read2_b32` at immediate
//     offsets from two addresses.  A back-trace that leaves that neighbourhood (never in the reference's regime) sends its wave's units of
//     that field through the general form on global memory, a real call (far_value3) -- same values, same blend;
//   * GRAD: the launch's inputs are the velocities BEFORE the projection's gradient subtraction plus p, and the subtraction
//     (u[:,1:-1,:] -= dt (p[:,1:,:] - p[:,:-1,:]) etc., SPEC_3D.md section 4) is applied while a plane is staged: k3_grad_subtract's
//     seven field passes (3.8 GB at configs[4]) disappear, the staging reads p three times (centre, y-1, x-1; z-1 is the thread's own
//     centre value of the previous plane) out of the same L1 lines.  The post-projection velocities are then never materialised.
// Tiles go to the XCDs in contiguous ranges (xcd_contiguous), so the window overlaps of neighbouring tiles are L2 hits.
// One advected value in the general form, every tap from GLOBAL memory (the projection's gradient subtraction re-applied per tap if
// GRAD): what a wave falls back to when a back-trace of its units leaves the 2 x 2 x 2 LDS neighbourhood.  A real call, so the hot
// code carries neither its registers nor its flat loads.  Field WHICH 0..3 = u, v, w, density; (z, y, x) = the cell; ui, vi, wi = the
// velocity samples already formed by the caller.
template <int WHICH, bool GRAD>
__device__ __attribute__((noinline)) float far_value3(const float *f, const float *p, int D, int H, int W, int pc, int pv, float dt,
                                                      int z, int y, int x, float ui, float vi, float wi) {
    const int Df = D + (WHICH == 2), Hf = H + (WHICH == 0), Wf = W + (WHICH == 1), pitch = WHICH == 1 ? pv : pc;
    auto val = [&](int zz, int yy, int xx) -> float {
        float r = f[(size_t)(zz * Hf + yy) * pitch + xx];
        if (GRAD && WHICH < 3) {
            bool on;
            int back;
            if (WHICH == 0) { on = yy >= 1 && yy <= H - 1; back = pc; }
            else if (WHICH == 1) { on = xx >= 1 && xx <= W - 1; back = 1; }
            else { on = zz >= 1 && zz <= D - 1; back = H * pc; }
            if (on) {
                const float *q = p + (size_t)(zz * H + yy) * pc + xx;
                const float gr = q[0] - q[-back];
                r = r - dt * gr;
            }
        }
        return r;
    };
    const float tx = dt * ui, ty = dt * vi, tz = dt * wi;
    const float px = clampf3((float)x - tx, 0.f, (float)(Wf - 1));
    const float py = clampf3((float)y - ty, 0.f, (float)(Hf - 1));
    const float pz = clampf3((float)z - tz, 0.f, (float)(Df - 1));
    int x0 = (int)floorf(px), y0 = (int)floorf(py), z0 = (int)floorf(pz);
    int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    x0 = clampi3(x0, 0, Wf - 1); x1 = clampi3(x1, 0, Wf - 1);
    y0 = clampi3(y0, 0, Hf - 1); y1 = clampi3(y1, 0, Hf - 1);
    z0 = clampi3(z0, 0, Df - 1); z1 = clampi3(z1, 0, Df - 1);
    const float wx0 = (float)x1 - px, wx1 = px - (float)x0;
    const float wy0 = (float)y1 - py, wy1 = py - (float)y0;
    const float wz0 = (float)z1 - pz, wz1 = pz - (float)z0;
    float acc = ((wx0 * wy0) * wz0) * val(z0, y0, x0);
    acc = acc + ((wx1 * wy0) * wz0) * val(z0, y0, x1);
    acc = acc + ((wx0 * wy1) * wz0) * val(z0, y1, x0);
    acc = acc + ((wx1 * wy1) * wz0) * val(z0, y1, x1);
    acc = acc + ((wx0 * wy0) * wz1) * val(z1, y0, x0);
    acc = acc + ((wx1 * wy0) * wz1) * val(z1, y0, x1);
    acc = acc + ((wx0 * wy1) * wz1) * val(z1, y1, x0);
    acc = acc + ((wx1 * wy1) * wz1) * val(z1, y1, x1);
    return acc;
}

// One wave = R consecutive rows of a 64-column tile; the workgroup's NW waves stack their rows (tile = NW R x 64) and share the input rings.
// Within a plane a wave runs Un (its rows + the next one + the tile's extra column), Vn (rows + 1), Wn, then Dn of the previous plane;
// what the later fields sample of the earlier ones stays in REGISTERS (thread = column: rows are register arrays, x + 1 comes from the
// next lane by DPP, lane 63's from the extra-column unit by v_readlane), so no barrier separates the fields: two workgroup barriers per
// plane, around the LDS writes of the staged input planes.  Units of one field are straight-line code up to ONE wave-level test
// "every back-trace of these units lands in its 2 x 2 x 2 LDS neighbourhood" (always, in the reference's regime); the fall-back runs the
// general form per lane.  EDGE = false is the same arithmetic with the tests that cannot fail inside the grid removed (existence of the
// cell, vel3_at's extent rules, the x / y clamps: an unclamped back-trace that would have needed its clamp fails the neighbourhood test).
template <int R, int NW, int MINW, int BU, bool GRAD>
__global__ __launch_bounds__(NW * 64, MINW) void k3_advect_march(Geom3 g, State3 in, const float *__restrict__ pf, State3 out,
                                                          float *__restrict__ frames, int64_t fsb) {
    constexpr int TY = R * NW, TX = 64, NT = NW * 64;
    constexpr int WR = TY + 3, WC = TX + 3, WP = 68, WPL = WR * WP;       // input windows: rows y0-1 .. y0+TY+1, columns x0-1 .. x0+TX+1
    constexpr int NST = (WR * WC + NT - 1) / NT;
    static_assert(R + 1 <= 64 && NST <= 8, "the extra column is one wave's lanes; staging condition bits");
    __shared__ float U2s[3 * WPL], V2s[3 * WPL], W2s[3 * WPL], D2s[3 * WPL];
    const int D = g.D, H = g.H, W = g.W, pc = g.pc, pv = g.pv;
    const int ntx = (W + TX - 1) / TX, nty = (H + TY - 1) / TY;
    unsigned tile = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tix = tile % ntx; tile /= ntx;
    const int tiy = tile % nty;
    const int b = tile / nty;
    const int x0 = tix * TX, y0 = tiy * TY;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *u2 = in.u + b * g.su, *v2 = in.v + b * g.sv, *w2 = in.w + b * g.sw, *d2 = in.d + b * g.sc;
    const float *pp = GRAD ? pf + b * g.sc : nullptr;
    float *uo = out.u + b * g.su, *vo = out.v + b * g.sv, *wo = out.w + b * g.sw, *dn = out.d + b * g.sc;
    float *fr = frames ? frames + (size_t)b * fsb : nullptr;
    const int pus = (H + 1) * pc, pvs = H * pv, pcs = H * pc;             // plane strides of u, v and of w / density / p
    const float dt = g.dt;

    // ---- staging plan of this thread: window elements e = tid + it * NT (the same elements in every plane)
    int so[NST], gc[NST], gu[NST], gv[NST];
    unsigned cond = 0;                                                    // bit it: element exists; 8 + it: u gets its gradient; 16 + it: v does
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int e0 = tid + it * NT, e = e0 < WR * WC ? e0 : WR * WC - 1;
        const int r = e / WC, c = e - r * WC;
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        const int yc = clampi3(gy, 0, H - 1), xc = clampi3(gx, 0, W - 1), yu = clampi3(gy, 0, H), xv = clampi3(gx, 0, W);
        so[it] = r * WP + c;
        gc[it] = 4 * (yc * pc + xc);                                       // byte offsets inside a plane of w2 / d2 / p, of u2, of v2
        gu[it] = 4 * (yu * pc + xc);
        gv[it] = 4 * (yc * pv + xv);
        if (e0 < WR * WC) cond |= 1u << it;
        if (yu >= 1 && yu <= H - 1) cond |= 1u << (8 + it);
        if (xv >= 1 && xv <= W - 1) cond |= 1u << (16 + it);
    }
    float ru[NST], rv[NST], rw[NST], rd[NST], rpc[NST], rpu[NST], rpl[NST], pprev[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) pprev[it] = 0.f;
    // request plane j of u2, v2, w2 (and of p) / plane jd of d2 -- every address is clamped into its field: no load sits under a branch.
    // Buffer addressing: descriptor base = field + plane (uniform), voffset = the element's byte offset inside a plane (the same VGPR
    // for every plane): no per-plane vector address arithmetic and no 64-bit address registers.
    auto plane_rsrc = [](const float *base) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
    };
    auto ldb = [](__amdgpu_buffer_rsrc_t rs, int voff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0)); };
    auto request = [&](int j, int jd) {
        const int zu = clampi3(j, 0, D - 1), zw = clampi3(j, 0, D), zd = clampi3(jd, 0, D - 1);
        const __amdgpu_buffer_rsrc_t bu = plane_rsrc(u2 + (size_t)zu * pus), bv = plane_rsrc(v2 + (size_t)zu * pvs);
        const __amdgpu_buffer_rsrc_t bw = plane_rsrc(w2 + (size_t)zw * pcs), bd = plane_rsrc(d2 + (size_t)zd * pcs);
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            ru[it] = ldb(bu, gu[it]);
            rv[it] = ldb(bv, gv[it]);
            rw[it] = ldb(bw, gc[it]);
            rd[it] = ldb(bd, gc[it]);
        }
    };
    // the p values the staged plane's gradient subtraction needs (issued later in the plane than the fields: shorter live ranges)
    auto request_p = [&](int j) {
        if (!GRAD) return;
        const int zu = clampi3(j, 0, D - 1);
        const __amdgpu_buffer_rsrc_t bp = plane_rsrc(pp + (size_t)zu * pcs);
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            rpc[it] = ldb(bp, gc[it]);
            rpu[it] = ldb(bp, gc[it] - ((cond >> (8 + it)) & 1u ? 4 * pc : 0));
            rpl[it] = ldb(bp, gc[it] - ((cond >> (16 + it)) & 1u ? 4 : 0));
        }
    };
    // write the requested planes into ring slots (floats): u2, v2, w2 plane j with the projection's gradient subtraction if GRAD; d2
    auto commit = [&](int j, int slot, int dslot) {
        const bool won = j >= 1 && j <= D - 1;
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            float a = ru[it], bq = rv[it], c = rw[it];
            if (GRAD) {
                if ((cond >> (8 + it)) & 1u) { const float gr = rpc[it] - rpu[it]; a = a - dt * gr; }
                if ((cond >> (16 + it)) & 1u) { const float gr = rpc[it] - rpl[it]; bq = bq - dt * gr; }
                if (won) { const float gr = rpc[it] - pprev[it]; c = c - dt * gr; }
                pprev[it] = rpc[it];
            }
            if ((cond >> it) & 1u) {
                U2s[slot + so[it]] = a;
                V2s[slot + so[it]] = bq;
                W2s[slot + so[it]] = c;
                if (dslot >= 0) D2s[dslot + so[it]] = rd[it];
            }
        }
    };
    auto slot3 = [](int j) { return ((j + 3) % 3) * WPL; };             // ring slot (in floats) of plane j >= -3

    // ---- prologue: planes -1 (= 0 clamped), 0, 1 of u2, v2, w2; planes -1 (= 0), 0 of d2
    request(-1, -1); request_p(-1); commit(-1, slot3(-1), slot3(-1));
    request(0, 0);   request_p(0);  commit(0, slot3(0), slot3(0));
    request(1, 0);   request_p(1);  commit(1, slot3(1), -1);
    __syncthreads();

    const int rb = wv * R, yb = y0 + rb;                                  // this wave's rows: yb .. yb+R-1 (+ row yb+R for Un, Vn)
    const int x = x0 + lane;
    const float fxl = (float)x;
    const int x4 = 4 * x;
    // one value of this lane's column into row `row` (uniform) of a field plane: descriptor base = the row, voffset = 4 x
    auto stb = [&](float *rowbase, float v) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), __builtin_amdgcn_make_buffer_rsrc(rowbase, 0, 0x7fffffff, 0x00020000), x4, 0, 0);
    };
    const int wbase = (rb + 1) * WP + lane + 1;                           // window index of (row yb, column x); + r WP for row yb + r
    const int wex = (rb + 1 + (lane <= R ? lane : R)) * WP + TX + 1;      // extra-column unit: lane = row, column x0 + TX
    const float fxe = (float)(x0 + TX);
    // is this wave's whole neighbourhood inside the grid?  (rows yb-1 .. yb+R+1 and columns x0-1 .. x0+TX+1 exist in every field,
    // vel3_at's y / x rules hold on all its units, and no coordinate below 2: (int)NaN = 0 must fail the neighbourhood test)
    const bool inner_w = yb >= 2 && yb + R + 1 <= H - 2 && x0 >= 2 && x0 + TX + 1 <= W - 2;

    // ---- the two halves of the fast form.  prep: back-trace -> weights + the two LDS addresses, returns "lands in the neighbourhood"
    struct Prep { float wx0, wx1, wy0, wy1, wz0, wz1; int a0, a1; };
    auto prep = [&](auto which, auto edge, int om, int oc, int op, int widx, int z, int y, int xx, float fx, float ui, float vi, float wi,
                    bool exists, Prep &P) -> bool {
        constexpr int WHICH = decltype(which)::value;
        constexpr bool EDGE = decltype(edge)::value;
        const int Df = D + (WHICH == 2), Hf = H + (WHICH == 0), Wf = W + (WHICH == 1);
        const float tx = dt * ui, ty = dt * vi, tz = dt * wi;
        float px = fx - tx, py = (float)y - ty;
        if (EDGE) { px = clampf3(px, 0.f, (float)(Wf - 1)); py = clampf3(py, 0.f, (float)(Hf - 1)); }
        const float pz = clampf3((float)z - tz, 0.f, (float)(Df - 1));
        const float fx0 = floorf(px), fy0 = floorf(py), fz0 = floorf(pz);
        float fx1 = fx0 + 1.f, fy1 = fy0 + 1.f;
        if (EDGE) { fx1 = fminf(fx1, (float)(Wf - 1)); fy1 = fminf(fy1, (float)(Hf - 1)); }
        const float fz1 = fminf(fz0 + 1.f, (float)(Df - 1));
        P.wx0 = fx1 - px; P.wx1 = px - fx0; P.wy0 = fy1 - py; P.wy1 = py - fy0; P.wz0 = fz1 - pz; P.wz1 = pz - fz0;
        int rx = (int)fx0 - xx, ry = (int)fy0 - y, rz = (int)fz0 - z;
        bool fast = (unsigned)((rx + 1) | (ry + 1) | (rz + 1)) <= 1u;
        if (EDGE && !exists) { rx = 0; ry = 0; rz = 0; fast = true; }       // a lane without a cell reads its own (in-window) slot; result unused
        const int a = widx + ry * WP + rx;
        P.a0 = a + (rz ? om : oc);
        P.a1 = a + (rz ? oc : op);
        return fast;
    };
    auto finish = [&](const float *ring, const Prep &P) -> float {
        const float t0 = ring[P.a0], t1 = ring[P.a0 + 1], t2 = ring[P.a0 + WP], t3 = ring[P.a0 + WP + 1];
        const float t4 = ring[P.a1], t5 = ring[P.a1 + 1], t6 = ring[P.a1 + WP], t7 = ring[P.a1 + WP + 1];
        float acc = ((P.wx0 * P.wy0) * P.wz0) * t0;
        acc = acc + ((P.wx1 * P.wy0) * P.wz0) * t1;
        acc = acc + ((P.wx0 * P.wy1) * P.wz0) * t2;
        acc = acc + ((P.wx1 * P.wy1) * P.wz0) * t3;
        acc = acc + ((P.wx0 * P.wy0) * P.wz1) * t4;
        acc = acc + ((P.wx1 * P.wy0) * P.wz1) * t5;
        acc = acc + ((P.wx0 * P.wy1) * P.wz1) * t6;
        acc = acc + ((P.wx1 * P.wy1) * P.wz1) * t7;
        return acc;
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    // advected velocities of this wave's rows: plane k (un, unx = Un at x+1, vn, wn) and plane k-1 (p*)
    float un[R + 1], unx[R + 1], vn[R + 1], wn[R], pun[R], punx[R], pvn[R + 1], pwn[R];
#pragma unroll
    for (int r = 0; r <= R; ++r) { un[r] = 0.f; unx[r] = 0.f; vn[r] = 0.f; pvn[r] = 0.f; }
#pragma unroll
    for (int r = 0; r < R; ++r) { wn[r] = 0.f; pun[r] = 0.f; punx[r] = 0.f; pwn[r] = 0.f; }

    // one plane of this wave.  sm / sc / sp: ring slots of planes k-1, k, k+1 (u2, v2, w2); dm / dc / dp: of planes k-2, k-1, k (d2).
    // Units of a field run in batches of BU (what is live at once: BU x (weights + addresses + samples)).
    auto plane = [&](auto edge, int k, int sm, int sc, int sp, int dm, int dc, int dp) {
        constexpr bool EDGE = decltype(edge)::value;
        const bool zuv = k <= D - 2, zw = k <= D - 1;                      // vel3_at's z rule for samples of u / v, of w at plane k
        float ui[BU], vi[BU], wi[BU], val[BU];
        Prep P[BU];
        bool ex[BU];
        if (k < D) {
            // ---------------- Un(k): units 0 .. R = rows yb .. yb+R at column x; unit R+1 = the tile's extra column x0 + TX (lane = row)
            float uex = 0.f;
#pragma unroll
            for (int b0 = 0; b0 <= R + 1; b0 += BU) {
                bool ok = true;
#pragma unroll
                for (int j = 0; j < BU; ++j) {
                    const int r = b0 + j;
                    if (r > R + 1) continue;
                    const bool xt = r == R + 1;
                    const int y = xt ? yb + lane : yb + r, xx = xt ? x0 + TX : x, w0 = xt ? wex : wbase + r * WP;
                    const float a = 0.5f * U2s[sc + w0] + 0.5f * U2s[sc + w0 + 1];
                    const float c = 0.5f * V2s[sc + w0] + 0.5f * V2s[sc + w0 + WP];
                    const float e = 0.5f * W2s[sc + w0] + 0.5f * W2s[sp + w0];
                    // (lanes of the extra-column unit beyond the rows compute row R's point again; nothing reads them)
                    ex[j] = (!EDGE && !xt) || ((!xt || lane <= R) && y <= H && xx <= W - 1);
                    ui[j] = (!EDGE || (zuv && y <= H - 1 && xx <= W - 2)) ? a : 0.f;
                    vi[j] = (!EDGE || (zuv && y <= H - 2)) ? c : 0.f;
                    wi[j] = (!EDGE || (y <= H - 2 && xx <= W - 2)) ? e : 0.f;
                    if (xt) ok &= prep(I0{}, std::true_type{}, sm, sc, sp, w0, k, y, xx, fxe, ui[j], vi[j], wi[j], ex[j], P[j]);
                    else ok &= prep(I0{}, edge, sm, sc, sp, w0, k, y, xx, fxl, ui[j], vi[j], wi[j], ex[j], P[j]);
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
                        val[j] = ex[j] ? far_value3<0, GRAD>(u2, pp, D, H, W, pc, pv, dt, k, xt ? yb + lane : yb + r, xt ? x0 + TX : x, ui[j], vi[j], wi[j]) : 0.f;
                    }
                }
#pragma unroll
                for (int j = 0; j < BU; ++j) {
                    const int r = b0 + j;
                    if (r > R + 1) continue;
                    if (r == R + 1) { uex = val[j]; continue; }
                    un[r] = val[j];
                    const int y = yb + r;
                    if ((r < R || (EDGE && y == H)) && ex[j]) stb(uo + (size_t)(k * (H + 1) + y) * pc, val[j]);
                }
            }
#pragma unroll
            for (int r = 0; r <= R; ++r)                                       // Un at x + 1: the next lane's, lane 63 takes the extra column's row r
                unx[r] = shl1_with(un[r], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, uex), r)));
            // ---------------- Vn(k): rows 0 .. R (columns x0 .. x0+63; x = W of a narrower last tile is one of these lanes)
#pragma unroll
            for (int b0 = 0; b0 <= R; b0 += BU) {
                bool ok = true;
#pragma unroll
                for (int j = 0; j < BU; ++j) {
                    const int r = b0 + j;
                    if (r > R) continue;
                    const int y = yb + r, w0 = wbase + r * WP;
                    const float a = 0.5f * un[r] + 0.5f * unx[r];
                    const float c = 0.5f * V2s[sc + w0] + 0.5f * V2s[sc + w0 + WP];
                    const float e = 0.5f * W2s[sc + w0] + 0.5f * W2s[sp + w0];
                    ex[j] = !EDGE || (y <= H - 1 && x <= W);
                    ui[j] = (!EDGE || (zuv && x <= W - 2)) ? a : 0.f;
                    vi[j] = (!EDGE || (zuv && y <= H - 2 && x <= W - 1)) ? c : 0.f;
                    wi[j] = (!EDGE || (y <= H - 2 && x <= W - 2)) ? e : 0.f;
                    ok &= prep(I1{}, edge, sm, sc, sp, w0, k, y, x, fxl, ui[j], vi[j], wi[j], ex[j], P[j]);
                }
                if (__builtin_expect(__all(ok), 1)) {
#pragma unroll
                    for (int j = 0; j < BU; ++j)
                        if (b0 + j <= R) val[j] = finish(V2s, P[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < BU; ++j)
                        if (b0 + j <= R) val[j] = ex[j] ? far_value3<1, GRAD>(v2, pp, D, H, W, pc, pv, dt, k, yb + b0 + j, x, ui[j], vi[j], wi[j]) : 0.f;
                }
#pragma unroll
                for (int j = 0; j < BU; ++j) {
                    const int r = b0 + j;
                    if (r > R) continue;
                    vn[r] = val[j];
                    if (r < R && ex[j]) stb(vo + (size_t)(k * H + yb + r) * pv, val[j]);
                }
            }
            if (EDGE && x0 + TX == W) {                                    // the field's own extra column x = W (lane = row, rows 0 .. R-1)
                const int y = yb + lane;
                if (lane < R && y <= H - 1) {
                    const float c = 0.5f * V2s[sc + wex] + 0.5f * V2s[sc + wex + WP];
                    const float vie = (zuv && y <= H - 2) ? c : 0.f;          // (ui and wi need x <= W-2: zero)
                    vo[(unsigned)((k * H + y) * pv + W)] = far_value3<1, GRAD>(v2, pp, D, H, W, pc, pv, dt, k, y, W, 0.f, vie, 0.f);
                }
            }
            request_p(k + 2);
        }
        // ---------------- Wn(k), k = 0 .. D: rows 0 .. R-1
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
                const float e = 0.5f * W2s[sc + w0] + 0.5f * W2s[sp + w0];
                ex[j] = !EDGE || (y <= H - 1 && x <= W - 1);
                ui[j] = (!EDGE || (zuv && x <= W - 2)) ? a : 0.f;
                vi[j] = (!EDGE || (zuv && y <= H - 2)) ? c : 0.f;
                wi[j] = (!EDGE || (zw && y <= H - 2 && x <= W - 2)) ? e : 0.f;
                ok &= prep(I2{}, edge, sm, sc, sp, w0, k, y, x, fxl, ui[j], vi[j], wi[j], ex[j], P[j]);
            }
            if (__builtin_expect(__all(ok), 1)) {
#pragma unroll
                for (int j = 0; j < BU; ++j)
                    if (b0 + j < R) val[j] = finish(W2s, P[j]);
            } else {
#pragma unroll
                for (int j = 0; j < BU; ++j)
                    if (b0 + j < R) val[j] = ex[j] ? far_value3<2, GRAD>(w2, pp, D, H, W, pc, pv, dt, k, yb + b0 + j, x, ui[j], vi[j], wi[j]) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < BU; ++j) {
                const int r = b0 + j;
                if (r >= R) continue;
                if (ex[j]) stb(wo + (size_t)(k * H + yb + r) * pc, val[j]);
                // Dn(k-1) below still needs Wn(k-1): it lives in pwn; wn[r] is plane k's from here on
                wn[r] = val[j];
            }
        }
        // ---------------- Dn(k-1): rows 0 .. R-1 (+ decay, frame)
        if (k >= 1) {
            const int z = k - 1;
            const bool zuvd = z <= D - 2;
#pragma unroll
            for (int b0 = 0; b0 < R; b0 += BU) {
                bool ok = true;
#pragma unroll
                for (int j = 0; j < BU; ++j) {
                    const int r = b0 + j;
                    if (r >= R) continue;
                    const int y = yb + r, w0 = wbase + r * WP;
                    const float a = 0.5f * pun[r] + 0.5f * punx[r];
                    const float c = 0.5f * pvn[r] + 0.5f * pvn[r + 1];
                    const float e = 0.5f * pwn[r] + 0.5f * wn[r];
                    ex[j] = !EDGE || (y <= H - 1 && x <= W - 1);
                    ui[j] = (!EDGE || (zuvd && x <= W - 2)) ? a : 0.f;
                    vi[j] = (!EDGE || (zuvd && y <= H - 2)) ? c : 0.f;
                    wi[j] = (!EDGE || (y <= H - 2 && x <= W - 2)) ? e : 0.f;
                    ok &= prep(I3{}, edge, dm, dc, dp, w0, z, y, x, fxl, ui[j], vi[j], wi[j], ex[j], P[j]);
                }
                if (__builtin_expect(__all(ok), 1)) {
#pragma unroll
                    for (int j = 0; j < BU; ++j)
                        if (b0 + j < R) val[j] = finish(D2s, P[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < BU; ++j)
                        if (b0 + j < R) val[j] = ex[j] ? far_value3<3, GRAD>(d2, pp, D, H, W, pc, pv, dt, z, yb + b0 + j, x, ui[j], vi[j], wi[j]) : 0.f;
                }
#pragma unroll
                for (int j = 0; j < BU; ++j) {
                    const int r = b0 + j;
                    if (r >= R) continue;
                    const float v = val[j] * 0.995f;                        // navier_stokes.py:171
                    if (ex[j]) {
                        const int y = yb + r;
                        if (fr) stb(fr + (size_t)(z * H + y) * W, v);
                        stb(dn + (size_t)(z * H + y) * pc, v);
                    }
                }
            }
        }
        // plane k becomes plane k-1
#pragma unroll
        for (int r = 0; r < R; ++r) { pun[r] = un[r]; punx[r] = unx[r]; pwn[r] = wn[r]; }
#pragma unroll
        for (int r = 0; r <= R; ++r) pvn[r] = vn[r];
    };

    int sm = slot3(-1), sc = slot3(0), sp = slot3(1);                      // ring slots of planes k-1, k, k+1 (u2, v2, w2)
    int dm = slot3(-2), dc = slot3(-1), dp = slot3(0);                     // ring slots of planes k-2, k-1, k (d2)
    for (int k = 0; k <= D; ++k) {
        if (k < D) request(k + 2, k + 1);
        // the lean form needs every z rule of this plane to hold as well: samples at plane k (Un, Vn, Wn) and k-1 (Dn)
        if (inner_w && k <= D - 2) plane(std::false_type{}, k, sm, sc, sp, dm, dc, dp);
        else plane(std::true_type{}, k, sm, sc, sp, dm, dc, dp);
        __syncthreads();                                                   // every wave is done with planes k-1 (u2, v2, w2) and k-2 (d2)
        if (k < D) commit(k + 2, sm, dm);
        __syncthreads();
        { const int t = sm; sm = sc; sc = sp; sp = t; }
        { const int t = dm; dm = dc; dc = dp; dp = t; }
    }
}

// ---------------------------------------------------------------- buoyancy + the four diffusions + the divergence, marching along z
// SPEC_3D.md sections 3, 4 (navier_stokes.py:50-72,136,154-160).  Same organisation as k3_advect_march: a workgroup owns a TY x 64 column of
// cells, the inputs live in LDS as rings of three planes of the (TY+3) x 67 window whose out-of-grid elements hold the CLAMPED in-grid
// value -- which IS the replicate padding of diffusion_step -- with the step's buoyancy (v[..., :-1] += dt (0.1 density)) applied to v as
// it is staged.  Every neighbour of a cell is an LDS read; every input element leaves HBM / L2 once per workgroup.  A wave owns R rows
// (thread = column): it forms the diffused u on its rows + 1, v on its rows and on the tile's extra column (lane = row; lane 63 takes its
// x + 1 neighbour from there, the others from the next lane by DPP), w, density, and -- one plane later, when w2(z+1) exists -- the
// divergence ((((u2[y+1] - u2[y]) + v2[x+1]) - v2[x]) + w2[z+1]) - w2[z]) / dt from its registers: launch3_divergence's re-read of the
// three diffused velocities (1.6 GB at configs[4]) and its launch disappear.  Per cell the expression trees of diffuse3_at / k3_divergence.
template <int R, int NW>
__global__ __launch_bounds__(NW * 64, 2) void k3_diffuse_div_march(Geom3 g, State3 in, State3 out, float *__restrict__ divf) {
    constexpr int TY = R * NW, TX = 64, NT = NW * 64;
    constexpr int WR = TY + 3, WC = TX + 3, WP = 68, WPL = WR * WP;
    constexpr int NST = (WR * WC + NT - 1) / NT;
    __shared__ float Us[3 * WPL], Vs[3 * WPL], Ws[3 * WPL], Ds[3 * WPL];
    const int D = g.D, H = g.H, W = g.W, pc = g.pc, pv = g.pv;
    const int ntx = (W + TX - 1) / TX, nty = (H + TY - 1) / TY;
    unsigned tile = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tix = tile % ntx; tile /= ntx;
    const int tiy = tile % nty;
    const int b = tile / nty;
    const int x0 = tix * TX, y0 = tiy * TY;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *u = in.u + b * g.su, *v = in.v + b * g.sv, *w = in.w + b * g.sw, *d = in.d + b * g.sc;
    float *uo = out.u + b * g.su, *vo = out.v + b * g.sv, *wo = out.w + b * g.sw, *dn = out.d + b * g.sc, *dv = divf + b * g.sc;
    const int pus = (H + 1) * pc, pvs = H * pv, pcs = H * pc;
    const float dt = g.dt, cuv = g.coef_uv, cd = g.coef_d;

    int so[NST], gc[NST], gu[NST], gv[NST];
    unsigned cond = 0;                                                    // bit it: element exists; 8 + it: v's element takes buoyancy (column < W)
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int e0 = tid + it * NT, e = e0 < WR * WC ? e0 : WR * WC - 1;
        const int r = e / WC, c = e - r * WC;
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        const int yc = clampi3(gy, 0, H - 1), xc = clampi3(gx, 0, W - 1), yu = clampi3(gy, 0, H), xv = clampi3(gx, 0, W);
        so[it] = r * WP + c;
        gc[it] = 4 * (yc * pc + xc);
        gu[it] = 4 * (yu * pc + xc);
        gv[it] = 4 * (yc * pv + xv);
        if (e0 < WR * WC) cond |= 1u << it;
        if (xv < W) cond |= 1u << (8 + it);
    }
    // (one plane in flight.  A second register set with plane k+3 requested while plane k+2 waits for its commit was built and measured:
    //  1.36 ms against 1.22 -- as with the Jacobi's third plane, more loads in flight cost more than the latency they hide)
    float ru[NST], rv[NST], rw[NST], rd[NST];
    auto plane_rsrc = [](const float *base) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
    };
    auto ldb = [](__amdgpu_buffer_rsrc_t rs, int voff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0)); };
    auto request = [&](int j) {                                         // planes j of u, v, density (clamped to D-1) and of w (clamped to D)
        const int zu = clampi3(j, 0, D - 1), zw = clampi3(j, 0, D);
        const __amdgpu_buffer_rsrc_t bu = plane_rsrc(u + (size_t)zu * pus), bv = plane_rsrc(v + (size_t)zu * pvs);
        const __amdgpu_buffer_rsrc_t bw = plane_rsrc(w + (size_t)zw * pcs), bd = plane_rsrc(d + (size_t)zu * pcs);
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            ru[it] = ldb(bu, gu[it]);
            rv[it] = ldb(bv, gv[it]);
            rw[it] = ldb(bw, gc[it]);
            rd[it] = ldb(bd, gc[it]);
        }
    };
    auto commit = [&](int slot) {
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            float bq = rv[it];
            if ((cond >> (8 + it)) & 1u) {                                  // navier_stokes.py:154-155: two fp32 roundings
                const float bb = rd[it] * 0.1f;
                bq = bq + dt * bb;
            }
            if ((cond >> it) & 1u) {
                Us[slot + so[it]] = ru[it];
                Vs[slot + so[it]] = bq;
                Ws[slot + so[it]] = rw[it];
                Ds[slot + so[it]] = rd[it];
            }
        }
    };
    auto slot3 = [](int j) { return ((j + 3) % 3) * WPL; };
    request(-1); commit(slot3(-1));
    request(0);  commit(slot3(0));
    request(1);  commit(slot3(1));
    __syncthreads();

    const int rb = wv * R, yb = y0 + rb;
    const int x = x0 + lane, x4 = 4 * x;
    const int wbase = (rb + 1) * WP + lane + 1;
    const int wex = (rb + 1 + (lane < R ? lane : R - 1)) * WP + TX + 1;    // extra-column unit: lane = row, column x0 + TX
    auto stb = [&](float *rowbase, int voff, float val) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), __builtin_amdgcn_make_buffer_rsrc(rowbase, 0, 0x7fffffff, 0x00020000), voff, 0, 0);
    };
    // diffuse3_at on the rings: centre, up (row-1), down, left, right, front (plane-1), back
    auto diffuse = [&](const float *ring, int sm, int sc, int sp, int w0, float coef) -> float {
        const float c = ring[sc + w0];
        float lap = ring[sc + w0 - WP] + ring[sc + w0 + WP];
        lap = lap + ring[sc + w0 - 1];
        lap = lap + ring[sc + w0 + 1];
        lap = lap + ring[sm + w0];
        lap = lap + ring[sp + w0];
        lap = lap - 6.0f * c;
        return c + coef * lap;
    };
    float a3[R], wprev[R];                                                 // ((u2[y+1] - u2[y]) + v2[x+1]) - v2[x] and w2 of the previous plane
#pragma unroll
    for (int r = 0; r < R; ++r) { a3[r] = 0.f; wprev[r] = 0.f; }
    const bool xin = x < W;
    int sm = slot3(-1), sc = slot3(0), sp = slot3(1);
    for (int k = 0; k <= D; ++k) {
        if (k < D) request(k + 2);
        float u2[R + 1], v2[R], v2x[R], w2[R];
        // ---- w2(k), k = 0 .. D; then the divergence of plane k-1
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int y = yb + r;
            w2[r] = diffuse(Ws, sm, sc, sp, wbase + r * WP, cuv);
            if (y < H && xin) {
                stb(wo + (size_t)(k * H + y) * pc, x4, w2[r]);
                if (k >= 1) {
                    float a = a3[r] + w2[r];
                    a = a - wprev[r];
                    stb(dv + (size_t)((k - 1) * H + y) * pc, x4, __fdiv_rn(a, dt));
                }
            }
        }
        if (k < D) {
            // ---- u2 on rows 0 .. R, v2 on rows 0 .. R-1 (+ the extra column), density
#pragma unroll
            for (int r = 0; r <= R; ++r) {
                const int y = yb + r;
                u2[r] = diffuse(Us, sm, sc, sp, wbase + r * WP, cuv);
                if ((r < R || y == H) && y <= H && xin) stb(uo + (size_t)(k * (H + 1) + y) * pc, x4, u2[r]);
            }
            const float vex = diffuse(Vs, sm, sc, sp, wex, cuv);            // lane l: v2 at (row yb + l, column x0 + TX)
            if (x0 + TX == W && lane < R && yb + lane < H) stb(vo + (size_t)(k * H + yb + lane) * pv, 4 * W, vex);      // the field's own last column
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = yb + r;
                v2[r] = diffuse(Vs, sm, sc, sp, wbase + r * WP, cuv);
                v2x[r] = shl1_with(v2[r], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vex), r)));
                const float d2 = diffuse(Ds, sm, sc, sp, wbase + r * WP, cd);
                if (y < H) {
                    if (x <= W) stb(vo + (size_t)(k * H + y) * pv, x4, v2[r]);      // (x = W: the last column of a tile narrower than 64)
                    if (xin) stb(dn + (size_t)(k * H + y) * pc, x4, d2);
                }
                float a = u2[r + 1] - u2[r];
                a = a + v2x[r];
                a3[r] = a - v2[r];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) wprev[r] = w2[r];
        __syncthreads();
        if (k < D) commit(sm);
        __syncthreads();
        { const int t = sm; sm = sc; sc = sp; sp = t; }
    }
}

hipError_t launch3_diffuse_div_march(const Geom3 &g, State3 in, State3 out, float *div, hipStream_t st) {
    constexpr int R = 4, NW = 4;
    const long long nb = (long long)cdiv(g.W, 64) * cdiv(g.H, R * NW) * g.B;
    if (nb > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k3_diffuse_div_march<R, NW>), dim3((unsigned)nb), dim3(NW * 64), 0, st, g, in, out, div);
    return hipGetLastError();
}

template <int R, int NW, int MINW, int BU, bool GRAD>
static hipError_t launch3_advect_march_t(const Geom3 &g, State3 in, const float *p, State3 out, float *frames, int64_t fsb, hipStream_t st) {
    const long long nb = (long long)cdiv(g.W, 64) * cdiv(g.H, R * NW) * g.B;
    if (nb > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k3_advect_march<R, NW, MINW, BU, GRAD>), dim3((unsigned)nb), dim3(NW * 64), 0, st, g, in, p, out, frames, fsb);
    return hipGetLastError();
}

hipError_t launch3_advect_march(const Geom3 &g, State3 in, const float *p, State3 out, float *frames, int64_t fsb, hipStream_t st) {
    static const int shape = [] { const char *e = getenv("SMK_ADVECT3_ROWS"); return e ? atoi(e) : 4; }();      // rows per wave (diagnostic: 2, 3; default 4)
    if (shape == 4) return p ? launch3_advect_march_t<4, 4, 2, 3, true>(g, in, p, out, frames, fsb, st) : launch3_advect_march_t<4, 4, 2, 3, false>(g, in, nullptr, out, frames, fsb, st);
    if (shape == 2) return p ? launch3_advect_march_t<2, 4, 4, 2, true>(g, in, p, out, frames, fsb, st) : launch3_advect_march_t<2, 4, 4, 2, false>(g, in, nullptr, out, frames, fsb, st);
    return p ? launch3_advect_march_t<3, 4, 3, 2, true>(g, in, p, out, frames, fsb, st) : launch3_advect_march_t<3, 4, 3, 2, false>(g, in, nullptr, out, frames, fsb, st);
}

}  // namespace smk
