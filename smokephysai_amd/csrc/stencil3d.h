// 3-D stepper (BASELINE configs[4]); semantics: SPEC_3D.md, the rule-by-rule generalisation of
// /root/reference/src/physics/navier_stokes.py:24-173 (the reference itself is 2-D only).
#pragma once
#include "common.h"

namespace smk {

// One batch of B grids [D][H][W].  Layout: [B][planes][rows][pitch]; pc = pitch of W-wide rows (u, w, p, density), pv = pitch of v's W+1.
struct Geom3 {
    int B, D, H, W;
    int pc, pv;
    size_t su, sv, sw, sc;   // per-grid strides in floats: u D(H+1)pc, v D H pv, w (D+1)H pc, cell fields D H pc
    float dt, coef_uv, coef_d, sixth;
};
struct State3 {
    float *u, *v, *w, *p, *d;
};
struct Src3Dev { int x, y, z, radius; float denom, fint; };

hipError_t launch3_zero(const Geom3 &g, State3 s, const uint8_t *dev_mask, hipStream_t st);
hipError_t launch3_add_sources(const Geom3 &g, float *density, const Src3Dev *src, const int *first, hipStream_t st);
hipError_t launch3_buoy_diffuse(const Geom3 &g, State3 in, State3 out, hipStream_t st);
hipError_t launch3_divergence(const Geom3 &g, State3 s, float *div, hipStream_t st);
// buoyancy + the four diffusions + the divergence of the diffused velocities as ONE z-marching launch (bit-identical to the two launches)
hipError_t launch3_diffuse_div_march(const Geom3 &g, State3 in, State3 out, float *div, hipStream_t st);
// `iters` sweeps on `div`; result in p (p2, and p3 if not null, scratch of the same layout: a third buffer lets an odd number of launches
// end in p without a copy)
hipError_t launch3_jacobi(const Geom3 &g, float *p, float *p2, float *p3, const float *div, int iters, hipStream_t st);
hipError_t launch3_grad_subtract(const Geom3 &g, State3 s, const float *p, hipStream_t st);
// which 0..3 = u, v, w, density (x 0.995, + optional frame [B][D][H][W] dense at frame_stride_b floats per grid)
hipError_t launch3_advect(const Geom3 &g, int which, const float *field, float *out, const float *u, const float *v, const float *w,
                          float *frames, int64_t frame_stride_b, hipStream_t st);

// the four advections of one step as one launch: in = (u2, v2, w2, -, d2), out = (u, v, w, -, density); bit-identical to the four launches
hipError_t launch3_advect_fused(const Geom3 &g, State3 in, State3 out, float *frames, int64_t frame_stride_b, hipStream_t st);
// the same as a z-marching launch (inputs staged once into LDS rings).  p != nullptr: `in` holds the velocities BEFORE the projection's
// gradient subtraction and the launch applies it on the fly (launch3_grad_subtract is then not run); bit-identical either way
hipError_t launch3_advect_march(const Geom3 &g, State3 in, const float *p, State3 out, float *frames, int64_t frame_stride_b, hipStream_t st);


// ---- device helpers shared by the 3-D kernel files
__device__ __forceinline__ float clampf3(float x, float lo, float hi) {
    float t = x < lo ? lo : x;   // torch.clamp = min(max(x, lo), hi)
    return t > hi ? hi : t;
}
__device__ __forceinline__ int clampi3(int x, int lo, int hi) {
    int t = x < lo ? lo : x;
    return t > hi ? hi : t;
}

}  // namespace smk
