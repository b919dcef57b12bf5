#pragma once
#include "common.h"

namespace smk {

// All launchers enqueue on `st` and return a hipError_t from hipGetLastError().
hipError_t launch_zero_state(const Geom &g, StateView s, const uint8_t *dev_mask, hipStream_t st);
// device-side source record: denom = (float)(2*(radius/3)^2), fint = (float)intensity (host-computed doubles)
struct SrcDev { int x, y, radius; float denom, fint; };
// dev_first[b]..dev_first[b+1] indexes the sources of grid b (stable order).
hipError_t launch_add_sources(const Geom &g, float *density, const SrcDev *dev_src, const int *dev_first, hipStream_t st);
hipError_t launch_buoy_diffuse(const Geom &g, StateView in, StateView out, hipStream_t st);
hipError_t launch_diffuse(const float *in, float *out, int B, int R, int C, int pitch, float coef, hipStream_t st);
hipError_t launch_divergence(const Geom &g, const float *u, const float *v, float *div, int div_pitch, size_t div_stride,
                             hipStream_t st);
// `iters` Jacobi sweeps; result ends in p. p2 is scratch of the same layout as p.
hipError_t launch_jacobi(const Geom &g, float *p, float *p2, const float *div, int iters, hipStream_t st);
// Per-handle state of the single-launch (persistent) projection: hand-off flags of the bands (device), a host-visible status word
// the kernel sets when a bounded wait times out, and the running hand-off count the flags are compared with.
struct ProjectSync {
    unsigned *flags = nullptr;
    volatile unsigned *status = nullptr;
    unsigned seq = 0;
    int flags_len = 0;
    bool disabled = false;
};
hipError_t project_sync_create(ProjectSync &ps, int B);
void project_sync_destroy(ProjectSync &ps);
// true exactly once after a persistent launch of this handle reported a timed-out hand-off (acknowledges the word, disables the persistent form)
bool project_sync_take_timeout(ProjectSync &ps);
// divergence + `iters` Jacobi sweeps + gradient subtraction on (u, v, p); p2 and div are scratch.  With `ps` the projection runs as one
// persistent launch where the plan allows (stencil.hip: k_jacobi_band<..., PERSIST>); returns hipErrorLaunchTimeOut once if an earlier
// persistent launch reported a timed-out wait, and uses the multi-launch form afterwards.
// fold_in / fold_d_out: when given and the persistent form is taken, the step's buoyancy + diffusion stage (fold_in -> u, v, fold_d_out)
// runs as that launch's prologue; *folded says whether it did (otherwise the caller's u, v must already hold the diffused fields).
hipError_t launch_project(const Geom &g, float *u, float *v, float *p, float *p2, float *div, int iters, hipStream_t st,
                          ProjectSync *ps = nullptr, const StateView *fold_in = nullptr, float *fold_d_out = nullptr, bool *folded = nullptr);
// buoyancy + diffusion (in -> out.u, out.v, out.d) followed by the projection of (out.u, out.v) with p: one launch where possible.
hipError_t launch_buoy_project(const Geom &g, StateView in, StateView out, float *p, float *div, int iters, hipStream_t st, ProjectSync *ps);
// JSON description of what launch_project does for this geometry (kernel, bands, launches, sweeps per launch, on-chip estimates).
std::string describe_projection(const Geom &g, int iters, const ProjectSync *ps = nullptr);
hipError_t launch_grad_subtract(const Geom &g, float *u, float *v, const float *p, hipStream_t st);
// kind 0: field=u (H+1 x W), 1: field=v (H x W+1), 2: density (H x W) with *0.995 decay and optional frame emit.
hipError_t launch_advect(const Geom &g, int kind, const float *field, float *out, const float *u, const float *v,
                         float *frames, int64_t frame_stride_b, const float *fractal, float fractal_intensity,
                         int32_t *x0, int32_t *y0, hipStream_t st);
// The three advections of one step (u, v, density with decay + frame emit) as one launch: in = (u2, v2, -, d2), out = (u, v, -, density).
hipError_t launch_advect_fused(const Geom &g, StateView in, StateView out, float *frames, int64_t frame_stride_b, const float *fractal,
                               float fractal_intensity, hipStream_t st);
// mode 0 bilinear_interpolate, 1 interpolate_velocity_u, 2 interpolate_velocity_v on n caller-given coordinates per field
// (cstride 0: the B fields share one coordinate list).
hipError_t launch_interp(int mode, const float *field, int B, int h, int w, int pitch, size_t fstride, const float *y,
                         const float *x, size_t cstride, size_t n, float *out, hipStream_t st);
hipError_t launch_fractal_constants(int N, float *perlin, float *mandel, float *field, hipStream_t st);
hipError_t launch_apply_fractal(const float *in, float *out, const float *fractal, int n_fields, int N, float intensity,
                                hipStream_t st);

}  // namespace smk
