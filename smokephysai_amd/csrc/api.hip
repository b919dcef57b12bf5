// C ABI of libsmokehip.so (see include/smokehip.h for the contract and the reference interfaces replaced).
#include <stdlib.h>

#include <string.h>

#include <map>
#include <mutex>
#include <vector>

#include "chaos.h"
#include "conv3d.h"
#include "decoder.h"
#include "elementwise.h"
#include "encoder.h"
#include "linear.h"
#include "norm.h"
#include "stencil.h"
#include "stencil3d.h"
#include "transformer.h"

namespace smk {
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

namespace {
std::mutex g_dev_mu;
std::map<int, int> g_num_cu;                                  // device -> CU count
std::map<std::pair<int, const void *>, int> g_dev_int;        // (device, key) -> cached value / "seen" marker
int current_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev;
}
}  // namespace

int device_num_cu() {
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(g_dev_mu);
    auto it = g_num_cu.find(dev);
    if (it != g_num_cu.end()) return it->second;
    hipDeviceProp_t prop;
    int n = 256;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    g_num_cu[dev] = n;
    return n;
}

bool first_use_begin(const void *key) {
    const int dev = current_device();
    g_dev_mu.lock();
    if (g_dev_int.count(std::make_pair(dev, key))) {
        g_dev_mu.unlock();
        return false;
    }
    return true;                                              // lock held until first_use_end
}
void first_use_end(const void *key) {
    g_dev_int.emplace(std::make_pair(current_device(), key), 1);
    g_dev_mu.unlock();
}

int device_cached_int(const void *key, int (*compute)()) {
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        auto it = g_dev_int.find(std::make_pair(dev, key));
        if (it != g_dev_int.end()) return it->second;
    }
    const int v = compute();                                  // (occupancy queries etc.: outside the lock)
    std::lock_guard<std::mutex> lk(g_dev_mu);
    return g_dev_int.emplace(std::make_pair(dev, key), v).first->second;
}
}  // namespace smk

using namespace smk;

struct smk_sim {
    Geom g;
    StateView s;          // caller-owned
    StateView t;          // library scratch (u2, v2, p2, d2)
    float *div = nullptr;
    float *perlin = nullptr, *mandel = nullptr, *fractal = nullptr;   // [N][N], square grids only
    uint8_t *dev_mask = nullptr;
    SrcDev *dev_src = nullptr;
    int *dev_first = nullptr;
    int src_cap = 0;
    int jacobi_iters = 20;
    int device = 0;
    ProjectSync psync;    // hand-off flags / status word of the single-launch projection
};

struct smk_sim3d {
    Geom3 g;
    State3 s;             // caller-owned
    State3 t;             // library scratch (u2, v2, w2, p2, d2)
    float *div = nullptr;
    float *p3 = nullptr;   // third pressure buffer: only when the Jacobi plan has an odd number (>= 3) of launches (launch3_jacobi)
    uint8_t *dev_mask = nullptr;
    Src3Dev *dev_src = nullptr;
    int *dev_first = nullptr;
    int src_cap = 0;
    int jacobi_iters = 20;
    int device = 0;
};

struct smk_decoder {
    DecoderDev d;
    float *blob = nullptr;
    int device = 0;
};

struct smk_linear {
    LinearDev l;
    int device = 0;
};

struct smk_encoder {
    EncoderDev e;
    float *blob = nullptr;
    unsigned short *blob16 = nullptr;
    int device = 0;
};

namespace {

// Every entry point runs on its handle's device and leaves the CALLER's current device as it found it (PyTorch tracks its own).
struct DeviceGuard {
    int prev = -1, rc = SMK_OK;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev && hipSetDevice(dev) != hipSuccess) {
            set_error("hipSetDevice failed for the handle's device");
            rc = SMK_ERR_HIP;
        }
        if (prev == dev) prev = -1;                           // nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

int check_launch(hipError_t e, const char *what) {
    if (e != hipSuccess) {
        set_error(std::string(what) + ": " + hipGetErrorString(e));
        return SMK_ERR_HIP;
    }
    return SMK_OK;
}

int report_timeout() {
    set_error("persistent projection: a band waited longer than 0.5 s for its neighbour's hand-off (its workgroups were not all "
              "resident at once -- is another kernel or process holding compute units?).  The pressure, velocity and density of the "
              "grids of that projection were set to NaN, and so is every frame emitted since: reset the simulator (smk_sim_reset).  "
              "Later projections of this handle use the multi-launch form (SMK_JACOBI_PERSIST=0 selects it from the start)");
    return SMK_ERR_TIMEOUT;
}

int project_status(hipError_t e) {
    if (e == hipErrorLaunchTimeOut) return report_timeout();
    return check_launch(e, "project");
}

// Every smk_sim_* call starts here: a time-out that a persistent projection of an EARLIER call has reported by now is returned by this
// call (once).  What the launches of the current call report is seen by smk_sim_status after the stream has been synchronised.
int sim_entry(smk_sim *sim) {
    return project_sync_take_timeout(sim->psync) ? report_timeout() : SMK_OK;
}

int run_stage(smk_sim *sim, int stage, float *frames, int64_t fsb, const float *fractal, float fint, hipStream_t st) {
    const Geom &g = sim->g;
    StateView &s = sim->s, &t = sim->t;
    switch (stage) {
        case SMK_STAGE_BUOY_DIFFUSE:   // s -> t (u2, v2, d2)
            return check_launch(launch_buoy_diffuse(g, s, t, st), "buoy_diffuse");
        case SMK_STAGE_PROJECT: {      // on t.u, t.v with s.p
            return project_status(launch_project(g, t.u, t.v, s.p, t.p, sim->div, sim->jacobi_iters, st, &sim->psync));
        }
        case SMK_STAGE_ADVECT_U:       // u <- adv(u2; u2, v2)
            return check_launch(launch_advect(g, 0, t.u, s.u, t.u, t.v, nullptr, 0, nullptr, 0.f, nullptr, nullptr, st),
                                "advect_u");
        case SMK_STAGE_ADVECT_V:       // v <- adv(v2; u, v2)
            return check_launch(launch_advect(g, 1, t.v, s.v, s.u, t.v, nullptr, 0, nullptr, 0.f, nullptr, nullptr, st),
                                "advect_v");
        case SMK_STAGE_ADVECT_D:       // density <- adv(d2; u, v) * 0.995 (+ frame emit)
            return check_launch(launch_advect(g, 2, t.d, s.d, s.u, s.v, frames, fsb, fractal, fint, nullptr, nullptr, st),
                                "advect_d");
    }
    set_error("unknown stage");
    return SMK_ERR_INVALID;
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

int smk_abi_version(void) { return SMK_ABI_VERSION; }
const char *smk_last_error(void) { return g_err.c_str(); }

int smk_sim_create(const smk_sim_desc *d, smk_sim **out) {
    SMK_REQUIRE(d && out, "null desc/out");
    SMK_REQUIRE(d->batch >= 1 && d->height >= 3 && d->width >= 3, "batch>=1, height,width>=3");
    SMK_REQUIRE(d->pitch_c >= d->width && d->pitch_v >= d->width + 1, "pitch_c >= W and pitch_v >= W+1");
    SMK_REQUIRE(d->jacobi_iters >= 0, "jacobi_iters >= 0");
    SMK_REQUIRE(d->u && d->v && d->p && d->density, "null state pointer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible");
        return SMK_ERR_NO_DEVICE;
    }
    DeviceGuard guard(d->device_id);
    int rc = guard.rc;
    if (rc) return rc;
    smk_sim *sim = new smk_sim();
    Geom &g = sim->g;
    g.B = d->batch; g.H = d->height; g.W = d->width; g.pc = d->pitch_c; g.pv = d->pitch_v;
    g.su = (size_t)(g.H + 1) * g.pc; g.sv = (size_t)g.H * g.pv; g.sc = (size_t)g.H * g.pc;
    g.dt = (float)d->dt;
    g.coef_uv = (float)(d->dt * d->viscosity);              // navier_stokes.py:72,158-159
    g.coef_d = (float)(d->dt * (d->viscosity * 0.1));       // navier_stokes.py:160
    sim->s = {d->u, d->v, d->p, d->density};
    sim->jacobi_iters = d->jacobi_iters;
    sim->device = d->device_id;
    size_t B = g.B;
    hipError_t e = hipSuccess;
    auto alloc = [&](float **p, size_t n) { if (e == hipSuccess) e = hipMalloc((void **)p, n * sizeof(float)); };
    alloc(&sim->t.u, B * g.su); alloc(&sim->t.v, B * g.sv); alloc(&sim->t.p, B * g.sc); alloc(&sim->t.d, B * g.sc);
    alloc(&sim->div, B * g.sc);
    if (e == hipSuccess) e = project_sync_create(sim->psync, g.B);
    if (e == hipSuccess) e = hipMalloc((void **)&sim->dev_mask, B);
    if (e == hipSuccess) e = hipMalloc((void **)&sim->dev_first, (B + 1) * sizeof(int));
    if (g.H == g.W && e == hipSuccess) {
        size_t n = (size_t)g.H * g.W;
        alloc(&sim->perlin, n); alloc(&sim->mandel, n); alloc(&sim->fractal, n);
        if (e == hipSuccess) e = launch_fractal_constants(g.H, sim->perlin, sim->mandel, sim->fractal, 0);
        if (e == hipSuccess) e = hipStreamSynchronize(0);
    }
    if (e != hipSuccess) {
        set_error(std::string("smk_sim_create: ") + hipGetErrorString(e));
        smk_sim_destroy(sim);
        return SMK_ERR_HIP;
    }
    *out = sim;
    return SMK_OK;
}

int smk_sim_destroy(smk_sim *sim) {
    if (!sim) return SMK_OK;
    DeviceGuard guard(sim->device);              // frees run on the handle's device; the caller's device is restored
    float *ptrs[] = {sim->t.u, sim->t.v, sim->t.p, sim->t.d, sim->div, sim->perlin, sim->mandel, sim->fractal};
    for (float *p : ptrs) if (p) (void)hipFree(p);           // (hipFree waits for the device: every launch of this handle has finished)
    const bool timed_out = project_sync_take_timeout(sim->psync);
    project_sync_destroy(sim->psync);
    if (sim->dev_mask) (void)hipFree(sim->dev_mask);
    if (sim->dev_first) (void)hipFree(sim->dev_first);
    if (sim->dev_src) (void)hipFree(sim->dev_src);
    delete sim;
    return timed_out ? report_timeout() : SMK_OK;            // the handle is gone either way; the caller learns its last frames were NaN
}

int smk_sim_status(smk_sim *sim) {
    SMK_REQUIRE(sim, "null sim");
    return sim_entry(sim);
}

int smk_sim_reset(smk_sim *sim, const uint8_t *grid_mask, void *stream) {
    SMK_REQUIRE(sim, "null sim");
    // a pending time-out report is acknowledged here like at every entry point, but the reset -- the recovery the report asks for -- is
    // still carried out: ONE call both tells the caller and leaves the requested grids zeroed (the message stays in smk_last_error)
    const int pending = sim_entry(sim);
    if (pending && pending != SMK_ERR_TIMEOUT) return pending;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(sim->device);
    int rc = guard.rc;
    if (rc) return rc;
    const uint8_t *dm = nullptr;
    if (grid_mask) {
        SMK_HIP_TRY(hipMemcpyAsync(sim->dev_mask, grid_mask, sim->g.B, hipMemcpyHostToDevice, st));
        dm = sim->dev_mask;
    }
    const hipError_t le = launch_zero_state(sim->g, sim->s, dm, st);
    if (le != hipSuccess) return check_launch(le, "zero_state");
    return pending;                                           // SMK_OK, or SMK_ERR_TIMEOUT once with the reset done
}

int smk_sim_add_sources(smk_sim *sim, const smk_source *src, int32_t n, void *stream) {
    SMK_REQUIRE(sim && (src || n == 0) && n >= 0, "null sim/sources");
    if (int e = sim_entry(sim)) return e;
    if (n == 0) return SMK_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(sim->device);
    int rc = guard.rc;
    if (rc) return rc;
    const int B = sim->g.B;
    std::vector<int> first(B + 1, 0);
    for (int k = 0; k < n; ++k) {
        SMK_REQUIRE(src[k].grid >= 0 && src[k].grid < B, "source grid index out of range");
        SMK_REQUIRE(src[k].radius >= 0, "radius >= 0");
        first[src[k].grid + 1]++;
    }
    for (int b = 0; b < B; ++b) first[b + 1] += first[b];
    std::vector<SrcDev> host(n);
    std::vector<int> fill(first.begin(), first.end() - 1);
    for (int k = 0; k < n; ++k) {      // stable counting sort by grid keeps the caller's order within a grid
        const smk_source &q = src[k];
        double r3 = (double)q.radius / 3.0;
        host[fill[q.grid]++] = SrcDev{q.x, q.y, q.radius, (float)(2.0 * (r3 * r3)), (float)q.intensity};
    }
    if (n > sim->src_cap) {
        if (sim->dev_src) SMK_HIP_TRY(hipFree(sim->dev_src));
        sim->dev_src = nullptr;
        SMK_HIP_TRY(hipMalloc((void **)&sim->dev_src, (size_t)n * sizeof(SrcDev)));
        sim->src_cap = n;
    }
    SMK_HIP_TRY(hipMemcpyAsync(sim->dev_src, host.data(), (size_t)n * sizeof(SrcDev), hipMemcpyHostToDevice, st));
    SMK_HIP_TRY(hipMemcpyAsync(sim->dev_first, first.data(), (size_t)(B + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    rc = check_launch(launch_add_sources(sim->g, sim->s.d, sim->dev_src, sim->dev_first, st), "add_sources");
    if (rc) return rc;
    SMK_HIP_TRY(hipStreamSynchronize(st));   // host staging vectors die here
    return SMK_OK;
}

int smk_sim_step(smk_sim *sim, int32_t n_steps, float *frames, int64_t fsb, int64_t fst, int32_t add_fractal,
                 double fractal_intensity, void *stream) {
    SMK_REQUIRE(sim && n_steps >= 0, "null sim / negative n_steps");
    if (int e = sim_entry(sim)) return e;
    if (add_fractal && frames && !sim->fractal) {
        // fractal_generator.py:44,49: the [w,h] mask indexes an [h,w] buffer -> the reference raises for h != w
        set_error("fractal perturbation needs a square grid (reference raises an IndexError for H != W)");
        return SMK_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(sim->device);
    int rc = guard.rc;
    if (rc) return rc;
    const float *fr = (add_fractal && frames) ? sim->fractal : nullptr;
    for (int t = 0; t < n_steps; ++t) {
        float *ft = frames ? frames + (size_t)t * fst : nullptr;
        static const bool staged = getenv("SMK_ADVECT_STAGED") != nullptr;      // diagnostic: the three advections as three launches
        if (staged) {
            for (int stage = SMK_STAGE_BUOY_DIFFUSE; stage <= SMK_STAGE_ADVECT_D; ++stage) {
                rc = run_stage(sim, stage, stage == SMK_STAGE_ADVECT_D ? ft : nullptr, fsb, fr, (float)fractal_intensity, st);
                if (rc) return rc;
            }
        } else {
            // buoyancy + diffusion + projection: one persistent launch where the plan allows (stencil.hip: launch_buoy_project)
            rc = project_status(launch_buoy_project(sim->g, sim->s, sim->t, sim->s.p, sim->div, sim->jacobi_iters, st, &sim->psync));
            if (rc) return rc;
        }
        if (!staged) {      // u <- adv(u2; u2, v2), v <- adv(v2; u, v2), density <- adv(d2; u, v) * 0.995 (+ frame) in one launch
            rc = check_launch(launch_advect_fused(sim->g, sim->t, sim->s, ft, fsb, fr, (float)fractal_intensity, st), "advect_fused");
            if (rc) return rc;
        }
    }
    return SMK_OK;
}

int smk_sim_run_stage(smk_sim *sim, int32_t stage, void *stream) {
    SMK_REQUIRE(sim, "null sim");
    if (int e = sim_entry(sim)) return e;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(sim->device);
    int rc = guard.rc;
    if (rc) return rc;
    const Geom &g = sim->g;
    const size_t B = g.B;
    StateView &s = sim->s, &t = sim->t;
    // Stand-alone stage semantics: caller state in, caller state out.  Internally the step ping-pongs between the
    // caller's tensors (s) and scratch (t); here the stage's inputs are first mirrored into the side it reads.
    auto cp = [&](float *dst, const float *src, size_t n) {
        return hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    };
    switch (stage) {
        case SMK_STAGE_BUOY_DIFFUSE:
            rc = run_stage(sim, stage, nullptr, 0, nullptr, 0.f, st);
            if (rc) return rc;
            SMK_HIP_TRY(cp(s.u, t.u, B * g.su)); SMK_HIP_TRY(cp(s.v, t.v, B * g.sv)); SMK_HIP_TRY(cp(s.d, t.d, B * g.sc));
            return SMK_OK;
        case SMK_STAGE_PROJECT:
            SMK_HIP_TRY(cp(t.u, s.u, B * g.su)); SMK_HIP_TRY(cp(t.v, s.v, B * g.sv));
            rc = run_stage(sim, stage, nullptr, 0, nullptr, 0.f, st);
            if (rc) return rc;
            SMK_HIP_TRY(cp(s.u, t.u, B * g.su)); SMK_HIP_TRY(cp(s.v, t.v, B * g.sv));
            return SMK_OK;
        case SMK_STAGE_ADVECT_U:      // reads (u, v) -> writes u
            SMK_HIP_TRY(cp(t.u, s.u, B * g.su)); SMK_HIP_TRY(cp(t.v, s.v, B * g.sv));
            return run_stage(sim, stage, nullptr, 0, nullptr, 0.f, st);
        case SMK_STAGE_ADVECT_V:      // reads v (field), u (already advected), v -> writes v
            SMK_HIP_TRY(cp(t.v, s.v, B * g.sv));
            return run_stage(sim, stage, nullptr, 0, nullptr, 0.f, st);
        case SMK_STAGE_ADVECT_D:
            SMK_HIP_TRY(cp(t.d, s.d, B * g.sc));
            return run_stage(sim, stage, nullptr, 0, nullptr, 0.f, st);
    }
    set_error("unknown stage");
    return SMK_ERR_INVALID;
}

int smk_sim_divergence(smk_sim *sim, float *out, void *stream) {
    SMK_REQUIRE(sim && out, "null sim/out");
    if (int e = sim_entry(sim)) return e;
    DeviceGuard guard(sim->device);
    int rc = guard.rc;
    if (rc) return rc;
    const Geom &g = sim->g;
    return check_launch(launch_divergence(g, sim->s.u, sim->s.v, out, g.W, (size_t)g.H * g.W, (hipStream_t)stream),
                        "divergence");
}

int smk_sim_backtrace(smk_sim *sim, int32_t which, int32_t *x0, int32_t *y0, void *stream) {
    SMK_REQUIRE(sim && x0 && y0 && which >= 0 && which <= 2, "null sim/x0/y0 or which not in 0..2");
    if (int e = sim_entry(sim)) return e;
    DeviceGuard guard(sim->device);
    int rc = guard.rc;
    if (rc) return rc;
    const StateView &s = sim->s;
    const float *field = which == 0 ? s.u : (which == 1 ? s.v : s.d);
    return check_launch(launch_advect(sim->g, which, field, nullptr, s.u, s.v, nullptr, 0, nullptr, 0.f, x0, y0,
                                      (hipStream_t)stream), "backtrace");
}

int smk_sim_fractal(smk_sim *sim, int32_t kind, const float **dev_ptr) {
    SMK_REQUIRE(sim && dev_ptr && kind >= 0 && kind <= 2, "null sim/dev_ptr or kind not in 0..2");
    if (!sim->fractal) {
        set_error("fractal constants exist for square grids only");
        return SMK_ERR_UNSUPPORTED;
    }
    *dev_ptr = kind == 0 ? sim->perlin : (kind == 1 ? sim->mandel : sim->fractal);
    return SMK_OK;
}

int smk_sim_describe(smk_sim *sim, char *buf, int64_t capacity) {
    SMK_REQUIRE(sim && buf && capacity > 0, "null sim / buf or no capacity");
    DeviceGuard guard(sim->device);
    if (guard.rc) return guard.rc;
    const std::string d = "{\"projection\": " + describe_projection(sim->g, sim->jacobi_iters, &sim->psync) +
                          ", \"advection\": \"k_advect_fused (u, v, density + frame in one LDS-tiled launch)\", \"launches_per_step\": null}";
    if ((int64_t)d.size() + 1 > capacity) {
        set_error("smk_sim_describe: buffer too small");
        return SMK_ERR_INVALID;
    }
    memcpy(buf, d.c_str(), d.size() + 1);
    return SMK_OK;
}

// ------------------------------------------------------------------ 3-D stepper (SPEC_3D.md)
int smk_sim3d_create(const smk_sim3d_desc *d, smk_sim3d **out) {
    SMK_REQUIRE(d && out, "null desc/out");
    SMK_REQUIRE(d->batch >= 1 && d->depth >= 3 && d->height >= 3 && d->width >= 3, "batch>=1, depth,height,width>=3");
    SMK_REQUIRE(d->pitch_c >= d->width && d->pitch_v >= d->width + 1, "pitch_c >= W and pitch_v >= W+1");
    SMK_REQUIRE(d->jacobi_iters >= 0, "jacobi_iters >= 0");
    SMK_REQUIRE((int64_t)d->batch * (d->depth + 1) <= 65535, "batch * (depth + 1) <= 65535 (one grid z-slice per plane)");
    SMK_REQUIRE((int64_t)(d->depth + 1) * (d->height + 1) * d->pitch_v < (1LL << 31), "one grid's field must stay below 2^31 elements (32-bit in-grid offsets)");
    SMK_REQUIRE(d->u && d->v && d->w && d->p && d->density, "null state pointer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible");
        return SMK_ERR_NO_DEVICE;
    }
    DeviceGuard guard(d->device_id);
    if (guard.rc) return guard.rc;
    smk_sim3d *sim = new smk_sim3d();
    Geom3 &g = sim->g;
    g.B = d->batch; g.D = d->depth; g.H = d->height; g.W = d->width; g.pc = d->pitch_c; g.pv = d->pitch_v;
    g.su = (size_t)g.D * (g.H + 1) * g.pc; g.sv = (size_t)g.D * g.H * g.pv; g.sw = (size_t)(g.D + 1) * g.H * g.pc;
    g.sc = (size_t)g.D * g.H * g.pc;
    g.dt = (float)d->dt;
    g.coef_uv = (float)(d->dt * d->viscosity);
    g.coef_d = (float)(d->dt * (d->viscosity * 0.1));
    g.sixth = (float)(1.0 / 6.0);
    sim->s = {d->u, d->v, d->w, d->p, d->density};
    sim->t = {nullptr, nullptr, nullptr, nullptr, nullptr};
    sim->jacobi_iters = d->jacobi_iters;
    sim->device = d->device_id;
    const size_t B = g.B;
    hipError_t e = hipSuccess;
    auto alloc = [&](float **p, size_t n) { if (e == hipSuccess) e = hipMalloc((void **)p, n * sizeof(float)); };
    alloc(&sim->t.u, B * g.su); alloc(&sim->t.v, B * g.sv); alloc(&sim->t.w, B * g.sw); alloc(&sim->t.p, B * g.sc); alloc(&sim->t.d, B * g.sc);
    alloc(&sim->div, B * g.sc);
    {   // J = 4 n with n odd and >= 3 (the default J = 20): five 4-sweep launches through a third buffer instead of four + two 2-sweep ones
        const int J = sim->jacobi_iters;
        if (J % 4 == 0 && ((J / 4) & 1) && J / 4 >= 3) alloc(&sim->p3, B * g.sc);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&sim->dev_mask, B);
    if (e == hipSuccess) e = hipMalloc((void **)&sim->dev_first, (B + 1) * sizeof(int));
    if (e != hipSuccess) {
        set_error(std::string("smk_sim3d_create: ") + hipGetErrorString(e));
        smk_sim3d_destroy(sim);
        return SMK_ERR_HIP;
    }
    *out = sim;
    return SMK_OK;
}

int smk_sim3d_destroy(smk_sim3d *sim) {
    if (!sim) return SMK_OK;
    DeviceGuard guard(sim->device);
    float *ptrs[] = {sim->t.u, sim->t.v, sim->t.w, sim->t.p, sim->t.d, sim->div, sim->p3};
    for (float *p : ptrs) if (p) (void)hipFree(p);
    if (sim->dev_mask) (void)hipFree(sim->dev_mask);
    if (sim->dev_first) (void)hipFree(sim->dev_first);
    if (sim->dev_src) (void)hipFree(sim->dev_src);
    delete sim;
    return SMK_OK;
}

int smk_sim3d_reset(smk_sim3d *sim, const uint8_t *grid_mask, void *stream) {
    SMK_REQUIRE(sim, "null sim");
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(sim->device);
    if (guard.rc) return guard.rc;
    const uint8_t *dm = nullptr;
    if (grid_mask) {
        SMK_HIP_TRY(hipMemcpyAsync(sim->dev_mask, grid_mask, sim->g.B, hipMemcpyHostToDevice, st));
        dm = sim->dev_mask;
    }
    return check_launch(launch3_zero(sim->g, sim->s, dm, st), "zero_state3d");
}

int smk_sim3d_add_sources(smk_sim3d *sim, const smk_source3d *src, int32_t n, void *stream) {
    SMK_REQUIRE(sim && (src || n == 0) && n >= 0, "null sim/sources");
    if (n == 0) return SMK_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(sim->device);
    if (guard.rc) return guard.rc;
    const int B = sim->g.B;
    std::vector<int> first(B + 1, 0);
    for (int k = 0; k < n; ++k) {
        SMK_REQUIRE(src[k].grid >= 0 && src[k].grid < B, "source grid index out of range");
        SMK_REQUIRE(src[k].radius >= 0, "radius >= 0");
        first[src[k].grid + 1]++;
    }
    for (int b = 0; b < B; ++b) first[b + 1] += first[b];
    std::vector<Src3Dev> host(n);
    std::vector<int> fill(first.begin(), first.end() - 1);
    for (int k = 0; k < n; ++k) {      // stable counting sort by grid keeps the caller's order within a grid
        const smk_source3d &q = src[k];
        const double r3 = (double)q.radius / 3.0;
        host[fill[q.grid]++] = Src3Dev{q.x, q.y, q.z, q.radius, (float)(2.0 * (r3 * r3)), (float)q.intensity};
    }
    if (n > sim->src_cap) {
        if (sim->dev_src) SMK_HIP_TRY(hipFree(sim->dev_src));
        sim->dev_src = nullptr;
        SMK_HIP_TRY(hipMalloc((void **)&sim->dev_src, (size_t)n * sizeof(Src3Dev)));
        sim->src_cap = n;
    }
    SMK_HIP_TRY(hipMemcpyAsync(sim->dev_src, host.data(), (size_t)n * sizeof(Src3Dev), hipMemcpyHostToDevice, st));
    SMK_HIP_TRY(hipMemcpyAsync(sim->dev_first, first.data(), (size_t)(B + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    const int rc = check_launch(launch3_add_sources(sim->g, sim->s.d, sim->dev_src, sim->dev_first, st), "add_sources3d");
    if (rc) return rc;
    SMK_HIP_TRY(hipStreamSynchronize(st));   // host staging vectors die here
    return SMK_OK;
}

namespace {
// one stage on the ping-pong pair: s = caller's tensors, t = scratch.  A step runs
//   (u, v, w, d) --buoy+diffuse--> t --project (p in place via p2)--> t --advect u--> s.u --advect v--> s.v --advect w--> s.w --advect d--> s.d
int run_stage3d(smk_sim3d *sim, int stage, float *frames, int64_t fsb, hipStream_t st, bool keep_gradient = false) {
    const Geom3 &g = sim->g;
    State3 &s = sim->s, &t = sim->t;
    switch (stage) {
        case SMK_STAGE3D_BUOY_DIFFUSE:
            return check_launch(launch3_buoy_diffuse(g, s, t, st), "buoy_diffuse3d");
        case SMK_STAGE3D_PROJECT: {
            int rc = check_launch(launch3_divergence(g, t, sim->div, st), "divergence3d");
            if (rc) return rc;
            rc = check_launch(launch3_jacobi(g, s.p, t.p, sim->p3, sim->div, sim->jacobi_iters, st), "jacobi3d");
            if (rc || keep_gradient) return rc;          // keep_gradient: the advection launch subtracts dt grad p while it stages its inputs
            return check_launch(launch3_grad_subtract(g, t, s.p, st), "grad_subtract3d");
        }
        case SMK_STAGE3D_ADVECT_U:
            return check_launch(launch3_advect(g, 0, t.u, s.u, t.u, t.v, t.w, nullptr, 0, st), "advect_u3d");
        case SMK_STAGE3D_ADVECT_V:
            return check_launch(launch3_advect(g, 1, t.v, s.v, s.u, t.v, t.w, nullptr, 0, st), "advect_v3d");
        case SMK_STAGE3D_ADVECT_W:
            return check_launch(launch3_advect(g, 2, t.w, s.w, s.u, s.v, t.w, nullptr, 0, st), "advect_w3d");
        case SMK_STAGE3D_ADVECT_D:
            return check_launch(launch3_advect(g, 3, t.d, s.d, s.u, s.v, s.w, frames, fsb, st), "advect_d3d");
    }
    set_error("unknown 3-D stage");
    return SMK_ERR_INVALID;
}
}  // namespace

int smk_sim3d_step(smk_sim3d *sim, int32_t n_steps, float *frames, int64_t fsb, int64_t fst, void *stream) {
    SMK_REQUIRE(sim && n_steps >= 0, "null sim / negative n_steps");
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(sim->device);
    if (guard.rc) return guard.rc;
    // diagnostics (read once): SMK_ADVECT3_STAGED the four advections as four launches; SMK_ADVECT3_TILE the 8 x 8 x 32 tile launch of
    // round 3; SMK_ADVECT3_GRAD=0 the z-marching launch behind a separate gradient-subtraction launch.  Default: the z-marching launch
    // with the gradient subtraction applied while it stages its inputs.
    static const bool staged = getenv("SMK_ADVECT3_STAGED") != nullptr;
    static const bool tiled = getenv("SMK_ADVECT3_TILE") != nullptr;
    static const bool fold_grad = !staged && !tiled && !(getenv("SMK_ADVECT3_GRAD") && atoi(getenv("SMK_ADVECT3_GRAD")) == 0);
    // SMK_DIFFUSE3_FUSED=0: buoyancy + diffusion and the divergence as two launches (the per-stage forms) instead of the z-marching one
    static const bool fused_dd = !(getenv("SMK_DIFFUSE3_FUSED") && atoi(getenv("SMK_DIFFUSE3_FUSED")) == 0);
    const Geom3 &g = sim->g;
    for (int t = 0; t < n_steps; ++t) {
        float *ft = frames ? frames + (size_t)t * fst : nullptr;
        int rc;
        if (fused_dd) {
            rc = check_launch(launch3_diffuse_div_march(g, sim->s, sim->t, sim->div, st), "diffuse_div_march3d");
            if (rc) return rc;
            rc = check_launch(launch3_jacobi(g, sim->s.p, sim->t.p, sim->p3, sim->div, sim->jacobi_iters, st), "jacobi3d");
            if (rc) return rc;
            if (!fold_grad) {
                rc = check_launch(launch3_grad_subtract(g, sim->t, sim->s.p, st), "grad_subtract3d");
                if (rc) return rc;
            }
        } else {
            for (int stage = SMK_STAGE3D_BUOY_DIFFUSE; stage <= SMK_STAGE3D_PROJECT; ++stage) {
                rc = run_stage3d(sim, stage, nullptr, fsb, st, fold_grad);
                if (rc) return rc;
            }
        }
        if (staged) {
            for (int stage = SMK_STAGE3D_ADVECT_U; stage <= SMK_STAGE3D_ADVECT_D; ++stage) {
                rc = run_stage3d(sim, stage, stage == SMK_STAGE3D_ADVECT_D ? ft : nullptr, fsb, st);
                if (rc) return rc;
            }
        } else {
            rc = tiled ? check_launch(launch3_advect_fused(g, sim->t, sim->s, ft, fsb, st), "advect_fused3d")
                       : check_launch(launch3_advect_march(g, sim->t, fold_grad ? sim->s.p : nullptr, sim->s, ft, fsb, st), "advect_march3d");
            if (rc) return rc;
        }
    }
    return SMK_OK;
}

int smk_sim3d_run_stage(smk_sim3d *sim, int32_t stage, void *stream) {
    SMK_REQUIRE(sim, "null sim");
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(sim->device);
    if (guard.rc) return guard.rc;
    const Geom3 &g = sim->g;
    const size_t B = g.B;
    State3 &s = sim->s, &t = sim->t;
    // stand-alone stage semantics: caller state in, caller state out -- the stage's inputs are first mirrored into the side it reads
    auto cp = [&](float *dst, const float *src, size_t n) { return hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st); };
    int rc;
    switch (stage) {
        case SMK_STAGE3D_BUOY_DIFFUSE:
            rc = run_stage3d(sim, stage, nullptr, 0, st);
            if (rc) return rc;
            SMK_HIP_TRY(cp(s.u, t.u, B * g.su)); SMK_HIP_TRY(cp(s.v, t.v, B * g.sv)); SMK_HIP_TRY(cp(s.w, t.w, B * g.sw));
            SMK_HIP_TRY(cp(s.d, t.d, B * g.sc));
            return SMK_OK;
        case SMK_STAGE3D_PROJECT:
            SMK_HIP_TRY(cp(t.u, s.u, B * g.su)); SMK_HIP_TRY(cp(t.v, s.v, B * g.sv)); SMK_HIP_TRY(cp(t.w, s.w, B * g.sw));
            rc = run_stage3d(sim, stage, nullptr, 0, st);
            if (rc) return rc;
            SMK_HIP_TRY(cp(s.u, t.u, B * g.su)); SMK_HIP_TRY(cp(s.v, t.v, B * g.sv)); SMK_HIP_TRY(cp(s.w, t.w, B * g.sw));
            return SMK_OK;
        case SMK_STAGE3D_ADVECT_U:      // reads (u, v, w) -> writes u
            SMK_HIP_TRY(cp(t.u, s.u, B * g.su)); SMK_HIP_TRY(cp(t.v, s.v, B * g.sv)); SMK_HIP_TRY(cp(t.w, s.w, B * g.sw));
            return run_stage3d(sim, stage, nullptr, 0, st);
        case SMK_STAGE3D_ADVECT_V:      // reads v (field), u (already advected), v, w -> writes v
            SMK_HIP_TRY(cp(t.v, s.v, B * g.sv)); SMK_HIP_TRY(cp(t.w, s.w, B * g.sw));
            return run_stage3d(sim, stage, nullptr, 0, st);
        case SMK_STAGE3D_ADVECT_W:
            SMK_HIP_TRY(cp(t.w, s.w, B * g.sw));
            return run_stage3d(sim, stage, nullptr, 0, st);
        case SMK_STAGE3D_ADVECT_D:
            SMK_HIP_TRY(cp(t.d, s.d, B * g.sc));
            return run_stage3d(sim, stage, nullptr, 0, st);
    }
    set_error("unknown 3-D stage");
    return SMK_ERR_INVALID;
}

int smk_conv3d_im2col(const float *src, int32_t C, int32_t D, int32_t H, int32_t W, int32_t ksize, int32_t z0, int32_t nz, float *cols,
                      int32_t kpad, void *stream) {
    SMK_REQUIRE(src && cols, "null src/cols");
    SMK_REQUIRE(C == 1 || C % 4 == 0, "C must be 1 or a multiple of 4");
    SMK_REQUIRE(D >= 1 && H >= 1 && W >= 1 && ksize >= 1 && (ksize & 1), "D, H, W >= 1; ksize odd");
    SMK_REQUIRE(z0 >= 0 && nz >= 1 && z0 + nz <= D, "plane range outside the volume");
    SMK_REQUIRE(kpad >= ksize * ksize * ksize * C && kpad % 4 == 0, "kpad >= ksize^3 * C and a multiple of 4");
    return check_launch(launch_im2col3d(src, C, D, H, W, ksize, z0, nz, cols, kpad, (hipStream_t)stream), "im2col3d");
}

int smk_pool3d_accumulate(const float *act, int32_t C, int32_t H, int32_t W, int32_t nz, float *sums, void *stream) {
    SMK_REQUIRE(act && sums && C >= 1 && nz >= 1, "null act/sums or bad sizes");
    if (H % 32 != 0 || W % 32 != 0) {
        set_error("pool3d: H and W must be multiples of 32 (the two adaptive pools then compose to a block mean)");
        return SMK_ERR_UNSUPPORTED;
    }
    return check_launch(launch_pool3d_accum(act, C, H, W, nz, sums, (hipStream_t)stream), "pool3d_accum");
}

int smk_diffuse(const float *in, float *out, int32_t B, int32_t R, int32_t C, int32_t pitch, double dt, double viscosity,
                void *stream) {
    SMK_REQUIRE(in && out && in != out, "null or aliased in/out");
    SMK_REQUIRE(B >= 1 && R >= 1 && C >= 1 && pitch >= C, "B,R,C >= 1, pitch >= C");
    return check_launch(launch_diffuse(in, out, B, R, C, pitch, (float)(dt * viscosity), (hipStream_t)stream), "diffuse");
}

// fractal constant cache for the stateless smk_apply_fractal: one per (device, N)
static std::mutex g_fr_mu;
static std::map<std::pair<int, int>, float *> g_fr_cache;

int smk_apply_fractal(const float *in, float *out, int32_t n_fields, int32_t N, double intensity, void *stream) {
    SMK_REQUIRE(in && out && n_fields >= 1 && N >= 2, "null in/out or bad sizes");
    int dev = 0;
    SMK_HIP_TRY(hipGetDevice(&dev));
    float *F = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_fr_mu);
        auto it = g_fr_cache.find({dev, N});
        if (it == g_fr_cache.end()) {
            float *blob = nullptr;
            size_t n = (size_t)N * N;
            SMK_HIP_TRY(hipMalloc((void **)&blob, 3 * n * sizeof(float)));
            int rc = check_launch(launch_fractal_constants(N, blob, blob + n, blob + 2 * n, (hipStream_t)stream), "fractal");
            if (rc) return rc;
            F = blob + 2 * n;
            g_fr_cache[{dev, N}] = F;
        } else {
            F = it->second;
        }
    }
    return check_launch(launch_apply_fractal(in, out, F, n_fields, N, (float)intensity, (hipStream_t)stream), "apply_fractal");
}

int smk_advect(const float *field, float *out, int32_t which, const float *u, const float *v, int32_t B, int32_t H,
               int32_t W, int32_t pitch_c, int32_t pitch_v, double dt, void *stream) {
    SMK_REQUIRE(field && out && u && v && field != out, "null or aliased pointers");
    SMK_REQUIRE(which >= 0 && which <= 2 && B >= 1 && H >= 2 && W >= 2, "which in 0..2, B>=1, H,W>=2");
    SMK_REQUIRE(pitch_c >= W && pitch_v >= W + 1, "pitch_c >= W and pitch_v >= W+1");
    Geom g{};
    g.B = B; g.H = H; g.W = W; g.pc = pitch_c; g.pv = pitch_v;
    g.su = (size_t)(H + 1) * pitch_c; g.sv = (size_t)H * pitch_v; g.sc = (size_t)H * pitch_c;
    g.dt = (float)dt;
    // which == 2 here is a plain advect (no decay / frame emit): reuse the u/v-style path via a dedicated flag
    return check_launch(launch_advect(g, which == 2 ? 3 : which, field, out, u, v, nullptr, 0, nullptr, 0.f, nullptr,
                                      nullptr, (hipStream_t)stream), "advect");
}

int smk_interpolate(int32_t mode, const float *field, int32_t B, int32_t h, int32_t w, int32_t pitch, int64_t field_stride,
                    const float *y, const float *x, int64_t coord_stride, int64_t n, float *out, void *stream) {
    SMK_REQUIRE(field && y && x && out, "null pointer");
    SMK_REQUIRE(mode >= 0 && mode <= 2, "mode: 0 bilinear_interpolate, 1 interpolate_velocity_u, 2 interpolate_velocity_v");
    SMK_REQUIRE(B >= 1 && B <= 65535 && h >= 1 && w >= 1 && pitch >= w && n >= 0 && field_stride >= 0 && coord_stride >= 0,
                "1 <= B <= 65535, h,w >= 1, pitch >= w, n >= 0");
    SMK_REQUIRE(n < ((int64_t)1 << 38), "n < 2^38 coordinates per field");
    if (n == 0) return SMK_OK;
    return check_launch(launch_interp(mode, field, B, h, w, pitch, (size_t)field_stride, y, x, (size_t)coord_stride, (size_t)n, out,
                                      (hipStream_t)stream), "interpolate");
}

int smk_fractal_constants(int32_t N, float *perlin, float *mandel, float *field, void *stream) {
    SMK_REQUIRE(N >= 2, "N >= 2");
    hipStream_t st = (hipStream_t)stream;
    size_t n = (size_t)N * N;
    float *tmp = nullptr;
    SMK_HIP_TRY(hipMalloc((void **)&tmp, 3 * n * sizeof(float)));
    int rc = check_launch(launch_fractal_constants(N, tmp, tmp + n, tmp + 2 * n, st), "fractal_constants");
    hipError_t e = hipSuccess;
    if (!rc && perlin) e = hipMemcpyAsync(perlin, tmp, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (!rc && e == hipSuccess && mandel) e = hipMemcpyAsync(mandel, tmp + n, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (!rc && e == hipSuccess && field) e = hipMemcpyAsync(field, tmp + 2 * n, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (rc) return rc;
    return check_launch(e, "fractal_constants copy");
}

int smk_chaos_stats(const float *frames, int64_t frame_stride, int32_t n, int32_t H, int32_t W, float *means,
                    int32_t *box_counts, int32_t *hist, void *stream) {
    SMK_REQUIRE(frames && means && box_counts && hist, "null pointer");
    SMK_REQUIRE(n >= 1 && H >= 2 && W >= 2 && frame_stride >= (int64_t)H * W, "n>=1, H,W>=2, frame_stride >= H*W");
    if ((int64_t)(H / 2) * (W / 2) > 65536) {
        set_error("chaos_stats: frames larger than 512x512 are not built");
        return SMK_ERR_UNSUPPORTED;
    }
    return check_launch(launch_chaos_stats(frames, frame_stride, n, H, W, means, box_counts, hist, (hipStream_t)stream),
                        "chaos_stats");
}

int smk_frame_diff_norms(const float *frames, int64_t frame_stride, int32_t n, int32_t H, int32_t W, float *norms,
                         void *stream) {
    SMK_REQUIRE(frames && norms && n >= 2 && H >= 1 && W >= 1 && frame_stride >= (int64_t)H * W, "bad arguments");
    return check_launch(launch_diff_norms(frames, frame_stride, n - 1, H * W, norms, (hipStream_t)stream), "diff_norms");
}

// ------------------------------------------------------------------ encoder
int smk_encoder_create(const smk_encoder_weights *w, int32_t device_id, void *stream, smk_encoder **out) {
    SMK_REQUIRE(w && out, "null weights/out");
    const float *const *pp = (const float *const *)w;
    for (size_t k = 0; k < sizeof(*w) / sizeof(float *); ++k) SMK_REQUIRE(pp[k], "null weight pointer");
    DeviceGuard guard(device_id);
    int rc = guard.rc;
    if (rc) return rc;
    smk_encoder *enc = new smk_encoder();
    enc->device = device_id;
    const size_t n = 64 * 49 + 64 + 64 + 9 * 64 * 128 + 128 + 128 + 128 + 256;
    hipError_t e = hipMalloc((void **)&enc->blob, n * sizeof(float));
    if (e != hipSuccess) {
        delete enc;
        set_error(std::string("smk_encoder_create: ") + hipGetErrorString(e));
        return SMK_ERR_HIP;
    }
    float *p = enc->blob;
    enc->e.w2t = p; p += 9 * 64 * 128;     // 16-byte aligned first
    enc->e.w1 = p; p += 64 * 49;
    enc->e.s1 = p; p += 64;
    enc->e.t1 = p; p += 64;
    enc->e.s2 = p; p += 128;
    enc->e.t2 = p; p += 128;
    enc->e.sw2 = p; p += 128;
    enc->e.wsum = reinterpret_cast<int *>(p);
    const size_t n16 = 2 * 64 * 64 + 2 * (36 * 2 * 128 * 16) + 18 * 2 * 128 * 32 / 2;   // + int8 limbs (bytes / 2)
    e = hipMalloc((void **)&enc->blob16, n16 * sizeof(unsigned short));
    if (e != hipSuccess) {
        smk_encoder_destroy(enc);
        set_error(std::string("smk_encoder_create: ") + hipGetErrorString(e));
        return SMK_ERR_HIP;
    }
    enc->e.w2q = enc->blob16;                       // 16-byte aligned rows first
    enc->e.w2s = enc->blob16 + 36 * 2 * 128 * 16;
    enc->e.w1p = enc->blob16 + 2 * (36 * 2 * 128 * 16);
    enc->e.w2i = reinterpret_cast<signed char *>(enc->blob16 + 2 * (36 * 2 * 128 * 16) + 2 * 64 * 64);
    rc = check_launch(launch_fold_weights(*w, enc->e, (hipStream_t)stream), "fold_weights");
    if (rc) { smk_encoder_destroy(enc); return rc; }
    *out = enc;
    return SMK_OK;
}

int smk_encoder_destroy(smk_encoder *enc) {
    if (!enc) return SMK_OK;
    DeviceGuard guard(enc->device);              // frees run on the handle's device; the caller's device is restored
    if (enc->blob) (void)hipFree(enc->blob);
    if (enc->blob16) (void)hipFree(enc->blob16);
    delete enc;
    return SMK_OK;
}

static int encoder_shape_ok(int32_t B, int32_t H, int32_t W, int32_t input_dim) {
    SMK_REQUIRE(B >= 1, "B >= 1");
    if (H != W || H % 32 != 0 || (H / 32 != 2 && H / 32 != 4 && H / 32 != 8)) {
        set_error("encoder: HIP path is built for square frames of 64, 128 or 256");
        return SMK_ERR_UNSUPPORTED;
    }
    if (input_dim <= 0 || input_dim % 32 != 0 || (input_dim % H != 0 && H % input_dim != 0)) {
        set_error("encoder: input_dim must be a multiple of 32 and a multiple or divisor of H (pools must compose to a block mean)");
        return SMK_ERR_UNSUPPORTED;
    }
    return SMK_OK;
}

int smk_encoder_forward(smk_encoder *enc, const float *frames, int64_t frame_stride, int32_t B, int32_t H, int32_t W,
                        int32_t input_dim, float *features, int32_t dtype, void *stream) {
    SMK_REQUIRE(enc && frames && features, "null enc/frames/features");
    SMK_REQUIRE(frame_stride >= (int64_t)H * W, "frame_stride >= H*W");
    int rc = encoder_shape_ok(B, H, W, input_dim);
    if (rc) return rc;
    DeviceGuard guard(enc->device);
    rc = guard.rc;
    if (rc) return rc;
    if (dtype == SMK_F32)
        return check_launch(launch_encoder_f32(frames, frame_stride, B, H, W, enc->e, features, (hipStream_t)stream), "encoder_f32");
    if (dtype == SMK_BF16X3 || dtype == SMK_BF16)
        return check_launch(launch_encoder_bf16(frames, frame_stride, B, H, W, enc->e, features, dtype == SMK_BF16X3, false,
                                                (hipStream_t)stream), "encoder_bf16");
    if (dtype == SMK_I8X3)
        return check_launch(launch_encoder_i8(frames, frame_stride, B, H, W, enc->e, features, false, (hipStream_t)stream),
                            "encoder_i8");
    set_error("unknown encoder dtype");
    return SMK_ERR_INVALID;
}

int smk_encoder_forward_tokens(smk_encoder *enc, const float *frames, int64_t frame_stride, int32_t B, int32_t H,
                               int32_t W, int32_t input_dim, float *tokens, int32_t dtype, void *stream) {
    SMK_REQUIRE(enc && frames && tokens, "null enc/frames/tokens");
    SMK_REQUIRE(frame_stride >= (int64_t)H * W, "frame_stride >= H*W");
    int rc = encoder_shape_ok(B, H, W, input_dim);
    if (rc) return rc;
    DeviceGuard guard(enc->device);
    rc = guard.rc;
    if (rc) return rc;
    if (dtype == SMK_I8X3)
        return check_launch(launch_encoder_i8(frames, frame_stride, B, H, W, enc->e, tokens, true, (hipStream_t)stream),
                            "encoder_i8_tokens");
    if (dtype != SMK_BF16X3 && dtype != SMK_BF16) {
        set_error("token-major output is built for the MFMA kernels SMK_BF16X3, SMK_BF16, SMK_I8X3");
        return SMK_ERR_UNSUPPORTED;
    }
    return check_launch(launch_encoder_bf16(frames, frame_stride, B, H, W, enc->e, tokens, dtype == SMK_BF16X3, true,
                                            (hipStream_t)stream), "encoder_bf16_tokens");
}

int smk_encoder_conv1(smk_encoder *enc, const float *frames, int64_t frame_stride, int32_t B, int32_t H, int32_t W,
                      float *act, void *stream) {
    SMK_REQUIRE(enc && frames && act && B >= 1 && H >= 1 && W >= 1, "null enc/frames/act or bad shape");
    DeviceGuard guard(enc->device);
    int rc = guard.rc;
    if (rc) return rc;
    return check_launch(launch_conv1_only(frames, frame_stride, B, H, W, enc->e, act, (hipStream_t)stream), "conv1");
}

// ------------------------------------------------------------------ transformer-body linear layers
int smk_linear_create(const float *weight, const float *bias, int32_t out_features, int32_t in_features,
                      int32_t device_id, void *stream, smk_linear **out) {
    SMK_REQUIRE(weight && out, "null weight/out");
    SMK_REQUIRE(out_features >= 1 && in_features >= 1, "positive feature counts");
    if (in_features % 64 != 0 || out_features % 32 != 0 || (int64_t)in_features * out_features > (1 << 28)) {
        set_error("linear: HIP path is built for in_features % 64 == 0, out_features % 32 == 0, at most 2^28 weights");
        return SMK_ERR_UNSUPPORTED;
    }
    DeviceGuard guard(device_id);
    int rc = guard.rc;
    if (rc) return rc;
    smk_linear *lin = new smk_linear();
    lin->device = device_id;
    lin->l.N = out_features;
    lin->l.K = in_features;
    lin->l.wq = nullptr;
    lin->l.bias = nullptr;
    hipError_t e = hipMalloc((void **)&lin->l.wq, (size_t)in_features * out_features * 2 * sizeof(unsigned short));
    if (e == hipSuccess) e = hipMalloc((void **)&lin->l.bias, (size_t)out_features * sizeof(float));
    if (e != hipSuccess) {
        smk_linear_destroy(lin);
        set_error(std::string("smk_linear_create: ") + hipGetErrorString(e));
        return SMK_ERR_HIP;
    }
    rc = check_launch(launch_split_linear_weights(weight, bias, lin->l, (hipStream_t)stream), "split_linear_weights");
    if (rc) { smk_linear_destroy(lin); return rc; }
    *out = lin;
    return SMK_OK;
}

int smk_linear_destroy(smk_linear *lin) {
    if (!lin) return SMK_OK;
    DeviceGuard guard(lin->device);              // frees run on the handle's device; the caller's device is restored
    if (lin->l.wq) (void)hipFree(lin->l.wq);
    if (lin->l.bias) (void)hipFree(lin->l.bias);
    delete lin;
    return SMK_OK;
}

int smk_linear_update(smk_linear *lin, const float *weight, int32_t transposed, const float *bias, void *stream) {
    SMK_REQUIRE(lin && weight, "null lin/weight");
    SMK_REQUIRE(transposed == 0 || transposed == 1, "transposed is 0 or 1");
    DeviceGuard guard(lin->device);
    int rc = guard.rc;
    if (rc) return rc;
    return check_launch(launch_split_linear_weights(weight, bias, lin->l, (hipStream_t)stream, transposed), "split_linear_weights");
}

int64_t smk_linear_wgrad_workspace(int64_t rows, int32_t out_features, int32_t in_features) {
    if (rows < 1 || out_features < 1 || in_features < 1) return 0;
    return (int64_t)plan_linear_wgrad(rows, out_features, in_features).bytes;
}

int smk_linear_wgrad(const float *dy, int64_t ld_dy, const float *x, int64_t ldx, int64_t rows, int32_t out_features,
                     int32_t in_features, float *dw, float *db, void *workspace, int64_t workspace_bytes, void *stream) {
    SMK_REQUIRE(dy && x && dw && workspace, "null dy/x/dw/workspace");
    SMK_REQUIRE(rows >= 1 && out_features >= 1 && in_features >= 1, "positive sizes");
    if (in_features % 32 != 0 || out_features % 4 != 0) {
        set_error("linear_wgrad: HIP path is built for in_features % 32 == 0, out_features % 4 == 0");
        return SMK_ERR_UNSUPPORTED;
    }
    SMK_REQUIRE(ld_dy >= out_features && ldx >= in_features, "row pitches >= feature counts");
    SMK_REQUIRE((int64_t)(out_features + 256) * (rows + 4096) < (1LL << 30), "(out_features + 256) * (rows + 4096) < 2^30: chunk the rows");
    SMK_REQUIRE(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)dw & 15) == 0, "workspace / dw 16-byte aligned");
    SMK_REQUIRE(workspace_bytes >= smk_linear_wgrad_workspace(rows, out_features, in_features), "workspace too small");
    return check_launch(launch_linear_wgrad(dy, ld_dy, x, ldx, rows, out_features, in_features, dw, db, workspace, (hipStream_t)stream),
                        "linear_wgrad");
}

int smk_linear_forward(smk_linear *lin, const void *x, int64_t rows, int64_t ldx, void *y, int64_t ldy,
                       const float *residual, int64_t ldr, const float *periodic_add, int32_t rows_per_group,
                       int32_t period, int32_t activation, int32_t x_format, int32_t y_format, void *stream) {
    SMK_REQUIRE(lin && x && y, "null lin/x/y");
    SMK_REQUIRE(rows >= 1 && rows < (1LL << 31) - 256, "1 <= rows < 2^31 - 256");
    SMK_REQUIRE(ldx >= lin->l.K && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0, "x rows: pitch >= in_features, 16-byte aligned");
    SMK_REQUIRE((rows + 256) * ldx < (1LL << 30), "x: (rows + 256) * ldx < 2^30 floats (32-bit buffer offsets)");
    SMK_REQUIRE(ldy >= lin->l.N && ldy % 4 == 0 && ((uintptr_t)y & 15) == 0, "y rows: pitch >= out_features, 16-byte aligned");
    SMK_REQUIRE(!residual || (ldr >= lin->l.N && ldr % 4 == 0 && ((uintptr_t)residual & 15) == 0),
                "residual rows: pitch >= out_features, 16-byte aligned");
    SMK_REQUIRE(!periodic_add || ((uintptr_t)periodic_add & 15) == 0, "periodic_add 16-byte aligned");
    SMK_REQUIRE(activation == SMK_ACT_NONE || activation == SMK_ACT_GELU || activation == SMK_ACT_RELU, "activation");
    SMK_REQUIRE(!(residual && periodic_add), "residual and periodic_add are exclusive (no layer of the path needs both)");
    SMK_REQUIRE((x_format == SMK_FMT_F32 || x_format == SMK_FMT_SPLIT_BF16) && (y_format == SMK_FMT_F32 || y_format == SMK_FMT_SPLIT_BF16),
                "x_format / y_format");
    SMK_REQUIRE(x_format == SMK_FMT_F32 || ldx == lin->l.K, "split x: dense rows (ldx == in_features)");
    SMK_REQUIRE(y_format == SMK_FMT_F32 || (ldy == lin->l.N && !residual), "split y: dense rows (ldy == out_features), no residual");
    if (periodic_add)
        SMK_REQUIRE(period >= 1 && rows_per_group >= 32 && rows_per_group % 32 == 0 && rows % rows_per_group == 0,
                    "periodic_add: period >= 1, rows_per_group a multiple of 32 that divides rows");
    DeviceGuard guard(lin->device);
    int rc = guard.rc;
    if (rc) return rc;
    LinearCall c;
    c.x = (const float *)x; c.ldx = ldx;
    c.y = (float *)y; c.ldy = ldy;
    c.x_split = x_format == SMK_FMT_SPLIT_BF16; c.y_split = y_format == SMK_FMT_SPLIT_BF16;
    c.res = residual; c.ldr = ldr;
    c.padd = periodic_add; c.rows_per_group = periodic_add ? rows_per_group : 1; c.period = periodic_add ? period : 1;
    c.M = (int)rows;
    c.act = activation;
    return check_launch(launch_linear_x3(lin->l, c, (hipStream_t)stream), "linear_x3");
}

int64_t smk_linear_ln_max_rows(smk_linear *lin) {
    if (!lin) return 0;
    DeviceGuard guard(lin->device);
    if (guard.rc) return 0;
    // (round 3: one tile per workgroup, i.e. 2 CUs / column tiles x 32 rows.)  The statistics now follow the chunk stream across the tiles a
    // workgroup walks, so the only bound is the 32-bit offset range of the activation buffer
    return ((1LL << 30) / lin->l.K) - 256;
}

int smk_linear_forward_ln(smk_linear *lin, const float *x, int64_t rows, int64_t ldx, float *y, int64_t ldy, const float *wsum, double eps,
                          const float *periodic_add, int32_t rows_per_group, int32_t period, int32_t activation, void *stream) {
    return smk_linear_forward_ln_split(lin, x, rows, ldx, y, ldy, wsum, eps, periodic_add, rows_per_group, period, activation, -1, stream);
}

int smk_linear_forward_ln_split(smk_linear *lin, const float *x, int64_t rows, int64_t ldx, float *y, int64_t ldy, const float *wsum, double eps,
                                const float *periodic_add, int32_t rows_per_group, int32_t period, int32_t activation, int32_t split_from_col,
                                void *stream) {
    SMK_REQUIRE(lin && x && y && wsum, "null lin/x/y/wsum");
    SMK_REQUIRE(split_from_col < 0 || (split_from_col % 32 == 0 && split_from_col < lin->l.N), "split_from_col: negative (none) or a multiple of 32 below out_features");
    SMK_REQUIRE(rows >= 1 && ldx >= lin->l.K && ldy >= lin->l.N && ldx % 4 == 0 && ldy % 4 == 0, "rows >= 1, row pitches >= features, multiples of 4");
    SMK_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)wsum) & 15) == 0, "16-byte aligned x / y / wsum");
    SMK_REQUIRE((rows + 256) * ldx < (1LL << 30), "(rows + 256) * ldx < 2^30 floats");
    SMK_REQUIRE(activation == SMK_ACT_NONE || activation == SMK_ACT_GELU || activation == SMK_ACT_RELU, "activation");
    SMK_REQUIRE(!periodic_add || (rows_per_group >= 32 && rows_per_group % 32 == 0 && period >= 1 && rows % rows_per_group == 0),
                "periodic_add: rows_per_group a multiple of 32 that divides rows, period >= 1");
    SMK_REQUIRE(eps > 0.0, "eps > 0");
    DeviceGuard guard(lin->device);
    if (guard.rc) return guard.rc;
    LinearCall c;
    c.x = x; c.ldx = ldx; c.y = y; c.ldy = ldy; c.x_split = 0; c.y_split = 0; c.res = nullptr; c.ldr = 0;
    c.padd = periodic_add; c.rows_per_group = periodic_add ? rows_per_group : 1; c.period = periodic_add ? period : 1;
    c.M = (int)rows; c.act = activation;
    c.ln_wsum = wsum; c.ln_eps = (float)eps;
    c.split_from = split_from_col < 0 ? -1 : split_from_col;
    return check_launch(launch_linear_x3(lin->l, c, (hipStream_t)stream), "linear_x3 (fused LayerNorm)");
}

int smk_conv3d_cl_forward(smk_linear *lin, const float *src, int32_t D, int32_t H, int32_t W, int32_t z0, int32_t nz, float *y, int64_t ldy,
                          int32_t activation, void *stream) {
    SMK_REQUIRE(lin && src && y, "null lin/src/y");
    SMK_REQUIRE(lin->l.K == 27 * 64, "the layer handle must have in_features = 27 * 64 (3 x 3 x 3 taps of 64 channels)");
    SMK_REQUIRE(D >= 1 && H >= 1 && W >= 1 && z0 >= 0 && nz >= 1 && z0 + nz <= D, "plane range outside the volume");
    SMK_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)y & 15) == 0 && ldy >= lin->l.N && ldy % 4 == 0, "16-byte aligned src / y, ldy >= out_features");
    SMK_REQUIRE((int64_t)(nz + 2) * H * W * 256 < (1LL << 32) - 256 && (int64_t)nz * H * W < (1LL << 31) - 256, "slab too large for 32-bit offsets: fewer planes per call");
    SMK_REQUIRE(activation == SMK_ACT_NONE || activation == SMK_ACT_GELU || activation == SMK_ACT_RELU, "activation");
    DeviceGuard guard(lin->device);
    if (guard.rc) return guard.rc;
    const int zlo = z0 > 0 ? z0 - 1 : 0, zhi = z0 + nz + 1 < D ? z0 + nz + 1 : D;
    return check_launch(launch_conv3d_cl_b16(lin->l, src + (size_t)zlo * H * W * 64, zhi - zlo, H, W, z0 - zlo, nz, y, ldy, activation,
                                             (hipStream_t)stream), "conv3d_cl_b16");
}

int smk_conv3d_cl_zsum_forward(smk_linear *lin, const float *src, int32_t D, int32_t H, int32_t W, float *zsum, int32_t activation, void *stream) {
    SMK_REQUIRE(lin && src && zsum, "null lin/src/zsum");
    SMK_REQUIRE(lin->l.K == 27 * 64 && lin->l.N == 128, "the layer handle must be 27 * 64 -> 128 (3 x 3 x 3 taps of 64 channels, 128 outputs)");
    SMK_REQUIRE(D >= 1 && H >= 8 && W >= 16 && H % 8 == 0 && W % 16 == 0, "H must be a multiple of 8 and W of 16 (the 8 x 16 voxel column a workgroup marches)");
    SMK_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)zsum & 15) == 0, "16-byte aligned src / zsum");
    SMK_REQUIRE((int64_t)H * W * 256 < (1LL << 31), "a plane must stay below 2^31 bytes (32-bit offsets within a plane)");
    SMK_REQUIRE(activation == SMK_ACT_NONE || activation == SMK_ACT_RELU, "activation: none or ReLU");
    DeviceGuard guard(lin->device);
    if (guard.rc) return guard.rc;
    return check_launch(launch_conv3d_march(lin->l, src, D, H, W, zsum, activation, (hipStream_t)stream), "conv3d_march");
}

int smk_conv3d_s7_march_forward(smk_linear *lin, const float *src, int32_t D, int32_t H, int32_t W, float *a1, int32_t activation, void *stream) {
    SMK_REQUIRE(lin && src && a1, "null lin/src/a1");
    SMK_REQUIRE(lin->l.K == 448 && lin->l.N == 64, "the layer handle must be 448 -> 64 (7 kz x 8 ky slots x 8 kx slots, 64 outputs)");
    SMK_REQUIRE(D >= 1 && H >= 8 && W >= 16 && H % 8 == 0 && W % 16 == 0, "H must be a multiple of 8 and W of 16 (the 8 x 16 voxel column a workgroup marches)");
    SMK_REQUIRE(((uintptr_t)src & 3) == 0 && ((uintptr_t)a1 & 15) == 0, "aligned src / a1");
    SMK_REQUIRE((int64_t)H * W * 256 < (1LL << 31), "a plane of the output must stay below 2^31 bytes");
    SMK_REQUIRE(activation == SMK_ACT_NONE || activation == SMK_ACT_RELU, "activation: none or ReLU");
    DeviceGuard guard(lin->device);
    if (guard.rc) return guard.rc;
    return check_launch(launch_conv3d_s7_march(lin->l, src, D, H, W, a1, activation, (hipStream_t)stream), "conv3d_s7_march");
}

int smk_conv3d_s7_forward(smk_linear *lin, const float *src, int32_t D, int32_t H, int32_t W, int32_t z0, int32_t nz, float *y, int64_t ldy,
                          int32_t activation, void *stream) {
    SMK_REQUIRE(lin && src && y, "null lin/src/y");
    SMK_REQUIRE(lin->l.K == 448, "the layer handle must have in_features = 448 (56 window rows x 8 kx slots)");
    SMK_REQUIRE(D >= 1 && H >= 1 && W >= 1 && H <= 1023 && W <= 1023 && z0 >= 0 && nz >= 1 && z0 + nz <= D, "plane range outside the volume, or H / W > 1023");
    SMK_REQUIRE(((uintptr_t)src & 3) == 0 && ((uintptr_t)y & 15) == 0 && ldy >= lin->l.N && ldy % 4 == 0, "aligned src / y, ldy >= out_features");
    SMK_REQUIRE((int64_t)(nz + 6) * H * W * 4 < (1LL << 32) - 256 && nz + 6 <= 1022 && (int64_t)nz * H * W < (1LL << 31) - 256, "slab too large: fewer planes per call");
    SMK_REQUIRE(activation == SMK_ACT_NONE || activation == SMK_ACT_GELU || activation == SMK_ACT_RELU, "activation");
    DeviceGuard guard(lin->device);
    if (guard.rc) return guard.rc;
    const int zlo = z0 > 3 ? z0 - 3 : 0, zhi = z0 + nz + 3 < D ? z0 + nz + 3 : D;
    return check_launch(launch_conv3d_s7_b16(lin->l, src + (size_t)zlo * H * W, zhi - zlo, H, W, z0 - zlo, nz, y, ldy, activation,
                                             (hipStream_t)stream), "conv3d_s7_b16");
}

// ------------------------------------------------------------------ chaos term of ChaosAttention
int smk_chaos_addend(const float *noise, int32_t B, int32_t D, const float *proj_w, const float *proj_b,
                     const float *gate_w, const float *gate_b, double strength, double sigma, double rho, double beta,
                     double dt, float *addend, int64_t ld_addend, void *stream) {
    SMK_REQUIRE(noise && proj_w && proj_b && gate_w && gate_b && addend, "null pointer");
    SMK_REQUIRE(B >= 1 && D >= 1 && ld_addend >= D && ld_addend < (1LL << 30), "B >= 1, D >= 1, ld_addend >= D");
    ChaosAddendArgs a;
    a.noise = noise; a.proj_w = proj_w; a.proj_b = proj_b; a.gate_w = gate_w; a.gate_b = gate_b; a.addend = addend;
    a.B = B; a.D = D; a.ld = (int)ld_addend;
    a.strength = (float)strength; a.sigma = (float)sigma; a.rho = (float)rho; a.beta = (float)beta; a.dt = (float)dt;
    return check_launch(launch_chaos_addend(a, (hipStream_t)stream), "chaos_addend");
}

int smk_chaos_addend_batched(int32_t n_layers, const smk_chaos_layer *layers, int32_t B, int32_t D, double sigma, double rho, double beta,
                             double dt, void *stream) {
    SMK_REQUIRE(layers && n_layers >= 1 && n_layers <= 8, "1 .. 8 layers");
    SMK_REQUIRE(B >= 1 && D >= 1, "B >= 1, D >= 1");
    ChaosAddendBatch a;
    a.NL = n_layers;
    for (int i = 0; i < n_layers; ++i) {
        const smk_chaos_layer &l = layers[i];
        SMK_REQUIRE(l.noise && l.proj_w && l.proj_b && l.gate_w && l.gate_b && l.addend, "null pointer in a layer");
        SMK_REQUIRE(l.ld_addend >= D && l.ld_addend < (1LL << 30), "ld_addend >= D");
        ChaosAddendArgs &x = a.layer[i];
        x.noise = l.noise; x.proj_w = l.proj_w; x.proj_b = l.proj_b; x.gate_w = l.gate_w; x.gate_b = l.gate_b; x.addend = l.addend;
        x.B = B; x.D = D; x.ld = (int)l.ld_addend;
        x.strength = (float)l.strength; x.sigma = (float)sigma; x.rho = (float)rho; x.beta = (float)beta; x.dt = (float)dt;
    }
    for (int i = n_layers; i < 8; ++i) a.layer[i] = a.layer[0];
    return check_launch(launch_chaos_addend_batch(a, (hipStream_t)stream), "chaos_addend_batch");
}

int smk_pooled_head(const float *x, int32_t B, int32_t L, int32_t D, int64_t ldx, const float *w1, const float *b1, int32_t H1,
                    const float *w2, const float *b2, int32_t H2, float *pooled, float *out, float *workspace, void *stream) {
    SMK_REQUIRE(x && w1 && b1 && w2 && b2 && pooled && out && workspace, "null pointer");
    SMK_REQUIRE(B >= 1 && B <= 65535 && L >= 1 && D >= 1 && H1 >= 1 && H2 >= 1 && ldx >= D, "B in 1 .. 65535, L, D, H1, H2 >= 1, ldx >= D");
    SMK_REQUIRE((int64_t)D * 4 <= 64 * 1024 && D % 4 == 0 && (H1 + 31) / 32 <= 65535, "D <= 16,384 (the pooled vector sits in LDS), D % 4 == 0");
    SMK_REQUIRE(((uintptr_t)w1 & 15) == 0, "16-byte aligned w1");
    PooledHeadArgs a;
    a.x = x; a.ldx = ldx; a.B = B; a.L = L; a.D = D; a.w1 = w1; a.b1 = b1; a.H1 = H1; a.w2 = w2; a.b2 = b2; a.H2 = H2;
    a.pooled = pooled; a.out = out; a.ws = workspace;
    return check_launch(launch_pooled_head(a, (hipStream_t)stream), "pooled_head");
}

int smk_lorenz_states(const float *noise, int32_t B, double sigma, double rho, double beta, double dt, float *states, void *stream) {
    SMK_REQUIRE(noise && states && B >= 1, "null noise / states or B < 1");
    return check_launch(launch_lorenz_states(noise, B, (float)sigma, (float)rho, (float)beta, (float)dt, states, (hipStream_t)stream), "lorenz_states");
}

// ------------------------------------------------------------------ element-wise chain of the FFN block in training
int smk_ffn_elementwise(int32_t op, const float *a, const float *b, float *out, int64_t n, double p, uint64_t seed, void *stream) {
    SMK_REQUIRE(a && out && n >= 0, "null a / out or negative n");
    SMK_REQUIRE(op >= SMK_ELT_GELU_DROPOUT_FWD && op <= SMK_ELT_DROPOUT_BWD, "op: smk_elt_op");
    SMK_REQUIRE((op != SMK_ELT_GELU_DROPOUT_BWD && op != SMK_ELT_DROPOUT_ADD_FWD) || b, "this op needs the second operand");
    SMK_REQUIRE(p >= 0.0 && p < 1.0, "0 <= p < 1");
    if (n % 4 != 0) {
        set_error("ffn_elementwise: HIP path is built for element counts that are multiples of 4");
        return SMK_ERR_UNSUPPORTED;
    }
    SMK_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "16-byte aligned tensors");
    if (n == 0) return SMK_OK;
    EltArgs e{a, b, out, (long long)n, (float)p, (unsigned long long)seed};
    hipStream_t st = (hipStream_t)stream;
    switch (op) {
        case SMK_ELT_GELU_DROPOUT_FWD: return check_launch(launch_gelu_dropout_fwd(e, st), "gelu_dropout_fwd");
        case SMK_ELT_GELU_DROPOUT_BWD: return check_launch(launch_gelu_dropout_bwd(e, st), "gelu_dropout_bwd");
        case SMK_ELT_DROPOUT_ADD_FWD: return check_launch(launch_dropout_add_fwd(e, st), "dropout_add_fwd");
        default: return check_launch(launch_dropout_bwd(e, st), "dropout_bwd");
    }
}

int smk_reduce_shards(const void *shards, int32_t in_dtype, int32_t world, int64_t n, int64_t stride, void *out, int32_t out_dtype, void *stream) {
    SMK_REQUIRE(shards && out && n >= 0 && world >= 1, "null shards / out, negative n or world < 1");
    SMK_REQUIRE((in_dtype == SMK_WIRE_F32 || in_dtype == SMK_WIRE_BF16) && (out_dtype == SMK_WIRE_F32 || out_dtype == SMK_WIRE_BF16), "dtype: smk_wire_dtype");
    SMK_REQUIRE(world == 1 || (stride >= n && stride % 4 == 0), "stride >= n and a multiple of 4");
    SMK_REQUIRE(((uintptr_t)shards & (in_dtype == SMK_WIRE_F32 ? 15 : 7)) == 0 && ((uintptr_t)out & (out_dtype == SMK_WIRE_F32 ? 15 : 7)) == 0,
                "fp32 operands 16-byte aligned, bf16 operands 8-byte aligned");
    SMK_REQUIRE(shards != out || in_dtype == out_dtype, "in place needs equal dtypes");
    return check_launch(launch_reduce_shards(shards, in_dtype == SMK_WIRE_BF16, world, n, stride, out, out_dtype == SMK_WIRE_BF16, (hipStream_t)stream), "reduce_shards");
}

// ------------------------------------------------------------------ softmax attention (chaos term folded into Q)
int smk_attention(const float *q, const float *k, const float *v, void *out, int32_t B, int32_t L, int32_t H,
                  int32_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, double scale, int32_t out_format,
                  void *stream) {
    return smk_attention_ws(q, k, v, out, B, L, H, head_dim, ldq, ldk, ldv, ldo, scale, out_format, nullptr, 0, stream);
}

int64_t smk_attention_workspace_bytes(int32_t B, int32_t L, int32_t H, int32_t head_dim) {
    if (B < 1 || H < 1 || L < 128 || head_dim != 64 || L % 128 != 0) return 0;
    return (int64_t)attention_workspace_bytes(B, L, H);
}

int smk_attention_ws(const float *q, const float *k, const float *v, void *out, int32_t B, int32_t L, int32_t H,
                     int32_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, double scale, int32_t out_format,
                     void *workspace, int64_t workspace_bytes, void *stream) {
    return smk_attention_kv(q, k, v, out, B, L, H, head_dim, ldq, ldk, ldv, ldo, scale, out_format, SMK_FMT_F32, workspace, workspace_bytes, stream);
}

int smk_attention_kv(const float *q, const void *k, const void *v, void *out, int32_t B, int32_t L, int32_t H,
                     int32_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, double scale, int32_t out_format, int32_t kv_format,
                     void *workspace, int64_t workspace_bytes, void *stream) {
    SMK_REQUIRE(q && k && v && out, "null q/k/v/out");
    SMK_REQUIRE(kv_format == SMK_FMT_F32 || kv_format == SMK_FMT_SPLIT4_INPLACE, "kv_format: SMK_FMT_F32 or SMK_FMT_SPLIT4_INPLACE");
    SMK_REQUIRE(B >= 1 && H >= 1 && L >= 128, "B >= 1, H >= 1, L >= 128");
    if (head_dim != 64 || L % 128 != 0) {
        set_error("attention: HIP path is built for head_dim 64 and L a multiple of 128");
        return SMK_ERR_UNSUPPORTED;
    }
    const int64_t cols = (int64_t)H * 64;
    SMK_REQUIRE(ldq >= cols && ldk >= cols && ldv >= cols && ldo >= cols, "row pitches >= H * head_dim");
    SMK_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "row pitches multiples of 4 floats");
    SMK_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0, "16-byte aligned tensors");
    SMK_REQUIRE((int64_t)B * L * ldk < (1LL << 29) && (int64_t)B * L * ldv < (1LL << 29) && (int64_t)B * H * (L / 128) < (1LL << 31),
                "B * L * ld < 2^29 floats (32-bit buffer offsets)");
    AttnArgs a;
    SMK_REQUIRE(out_format == SMK_FMT_F32 || (out_format == SMK_FMT_SPLIT_BF16 && ldo == cols), "out_format (split: ldo == H * head_dim)");
    a.q = q; a.k = (const float *)k; a.v = (const float *)v; a.o = (float *)out; a.o_split = out_format == SMK_FMT_SPLIT_BF16;
    a.kv_split = kv_format == SMK_FMT_SPLIT4_INPLACE;
    a.ldq = (int)ldq; a.ldk = (int)ldk; a.ldv = (int)ldv; a.ldo = (int)ldo;
    a.B = B; a.L = L; a.H = H;
    a.scale_log2e = (float)(scale * 1.4426950408889634074);
    if (workspace && out_format == SMK_FMT_F32) {            // enough for the split the launcher would choose, or none at all
        const int64_t need = (int64_t)attention_workspace_bytes(B, L, H);
        SMK_REQUIRE(((uintptr_t)workspace & 15) == 0, "16-byte aligned workspace");
        if (need > 0 && workspace_bytes >= need) a.ws = (float *)workspace;
    }
    return check_launch(launch_attention_x3(a, (hipStream_t)stream), "attention_x3");
}

int smk_attention_forward_lse(const float *q, const float *k, const float *v, float *out, float *lse, int32_t B, int32_t L,
                              int32_t H, int32_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, double scale,
                              void *stream) {
    SMK_REQUIRE(q && k && v && out && lse, "null q/k/v/out/lse");
    SMK_REQUIRE(B >= 1 && H >= 1 && L >= 128, "B >= 1, H >= 1, L >= 128");
    if (head_dim != 64 || L % 128 != 0) {
        set_error("attention: HIP path is built for head_dim 64 and L a multiple of 128");
        return SMK_ERR_UNSUPPORTED;
    }
    const int64_t cols = (int64_t)H * 64;
    SMK_REQUIRE(ldq >= cols && ldk >= cols && ldv >= cols && ldo >= cols, "row pitches >= H * head_dim");
    SMK_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "row pitches multiples of 4 floats");
    SMK_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0, "16-byte aligned tensors");
    SMK_REQUIRE((int64_t)B * L * ldk < (1LL << 29) && (int64_t)B * L * ldv < (1LL << 29) && (int64_t)B * H * (L / 128) < (1LL << 31),
                "B * L * ld < 2^29 floats (32-bit buffer offsets)");
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.o = out; a.o_split = 0; a.lse = lse;
    a.ldq = (int)ldq; a.ldk = (int)ldk; a.ldv = (int)ldv; a.ldo = (int)ldo;
    a.B = B; a.L = L; a.H = H;
    a.scale_log2e = (float)(scale * 1.4426950408889634074);
    return check_launch(launch_attention_x3(a, (hipStream_t)stream), "attention_x3");
}

int smk_attention_delta(const float *dout, const float *out, int64_t rows, int32_t H, int32_t head_dim, int64_t ld_dout, int64_t ld_out,
                        float *delta, void *stream) {
    SMK_REQUIRE(dout && out && delta && rows >= 0 && H >= 1, "null pointer / rows < 0 / H < 1");
    if (head_dim != 64) {
        set_error("attention_delta: HIP path is built for head_dim 64");
        return SMK_ERR_UNSUPPORTED;
    }
    SMK_REQUIRE(ld_dout >= (int64_t)H * 64 && ld_out >= (int64_t)H * 64 && ld_dout % 4 == 0 && ld_out % 4 == 0, "row pitches >= H * 64, multiples of 4");
    SMK_REQUIRE((((uintptr_t)dout | (uintptr_t)out) & 15) == 0, "16-byte aligned tensors");
    if (rows == 0) return SMK_OK;
    return check_launch(launch_attn_delta(dout, out, rows, H, ld_dout, ld_out, delta, (hipStream_t)stream), "attn_delta");
}

int smk_attention_backward(const float *q, const float *k, const float *v, const float *dout, const float *lse,
                           const float *delta, float *dq, float *dk, float *dv, int32_t B, int32_t L, int32_t H,
                           int32_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk,
                           int64_t lddv, double scale, void *stream) {
    SMK_REQUIRE(q && k && v && dout && lse && delta && dq && dk && dv, "null pointer");
    SMK_REQUIRE(B >= 1 && H >= 1 && L >= 128, "B >= 1, H >= 1, L >= 128");
    if (head_dim != 64 || L % 128 != 0) {
        set_error("attention: HIP path is built for head_dim 64 and L a multiple of 128");
        return SMK_ERR_UNSUPPORTED;
    }
    const int64_t cols = (int64_t)H * 64;
    SMK_REQUIRE(ldq >= cols && ldk >= cols && ldv >= cols && ldo >= cols && lddq >= cols && lddk >= cols && lddv >= cols,
                "row pitches >= H * head_dim");
    SMK_REQUIRE((ldq | ldk | ldv | ldo | lddq | lddk | lddv) % 4 == 0, "row pitches multiples of 4 floats");
    SMK_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
                "16-byte aligned tensors");
    SMK_REQUIRE((int64_t)B * H * (L / 128) < (1LL << 31), "B * H * L / 128 < 2^31");
    AttnBwdArgs a;
    a.q = q; a.k = k; a.v = v; a.dout = dout; a.lse = lse; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv;
    a.ldq = (int)ldq; a.ldk = (int)ldk; a.ldv = (int)ldv; a.ldo = (int)ldo; a.lddq = (int)lddq; a.lddk = (int)lddk; a.lddv = (int)lddv;
    a.B = B; a.L = L; a.H = H;
    a.scale = (float)scale;
    a.scale_log2e = (float)(scale * 1.4426950408889634074);
    return check_launch(launch_attention_bwd_x3(a, (hipStream_t)stream), "attention_bwd_x3");
}

// ------------------------------------------------------------------ training-mode BatchNorm + ReLU + pool
static int bn_check(int32_t B, int32_t C, int32_t H, int32_t W, int32_t pool) {
    SMK_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1, "positive sizes");
    const int64_t chunk = pool == 8 ? 16384 : 4096;
    if (!(pool == 1 || pool == 4 || pool == 8) || ((int64_t)H * W) % chunk != 0 || (pool > 1 && (W != 32 * pool || H % pool != 0)) ||
        (int64_t)B * (H * (int64_t)W / chunk) >= (1LL << 31) || C > 65535) {
        set_error("bn_relu_pool: HIP path is built for pool in {1, 4, 8}, H * W a multiple of 4096 (pool 8: 16384), W == 32 * pool when pooling");
        return SMK_ERR_UNSUPPORTED;
    }
    return SMK_OK;
}

int smk_conv1_train_forward(const float *x, const float *weight, const float *bias, int32_t B, int32_t H, int32_t W, float *z1, void *stream) {
    SMK_REQUIRE(x && weight && z1, "null x/weight/z1");
    if (B < 1 || H < 1 || W < 4 || W % 4 != 0 || B > 65535) {
        set_error("conv1_train_forward: 1 <= B <= 65535, W a multiple of 4");
        return SMK_ERR_UNSUPPORTED;
    }
    return check_launch(launch_conv1_train_forward(x, weight, bias, B, H, W, z1, (hipStream_t)stream), "conv1_train_forward");
}

int64_t smk_conv1_train_wgrad_workspace(void) { return (int64_t)conv1_wgrad_workspace_bytes(); }

int smk_conv1_train_wgrad(const float *dz, const float *x, int32_t B, int32_t H, int32_t W, float *dw, float *db, void *workspace, void *stream) {
    SMK_REQUIRE(dz && x && dw && workspace, "null dz/x/dw/workspace");
    if (B < 1 || H < 4 || W < 64 || H % 4 != 0 || W % 64 != 0) {
        set_error("conv1_train_wgrad: B >= 1, H a multiple of 4, W a multiple of 64");
        return SMK_ERR_UNSUPPORTED;
    }
    return check_launch(launch_conv1_train_wgrad(dz, x, B, H, W, dw, db, workspace, (hipStream_t)stream), "conv1_train_wgrad");
}

int64_t smk_conv2_train_workspace(void) { return (int64_t)conv2_train_workspace_bytes(); }

int smk_conv2_train_forward(const float *a1, const float *weight, const float *bias, int32_t B, int32_t H, int32_t W, float *z2,
                            void *workspace, void *stream) {
    SMK_REQUIRE(a1 && weight && z2 && workspace, "null a1/weight/z2/workspace");
    if (B < 1 || H < 8 || W < 16 || H % 8 != 0 || W % 16 != 0 || (int64_t)B * 128 * H * W >= (1ll << 40)) {
        set_error("conv2_train_forward: B >= 1, H a multiple of 8, W a multiple of 16");
        return SMK_ERR_UNSUPPORTED;
    }
    return check_launch(launch_conv2_train_forward(a1, weight, bias, B, H, W, z2, workspace, (hipStream_t)stream), "conv2_train_forward");
}

int smk_conv2_train_dgrad(const float *dz, const float *weight, int32_t B, int32_t H, int32_t W, float *dx, void *workspace, void *stream) {
    SMK_REQUIRE(dz && weight && dx && workspace, "null dz/weight/dx/workspace");
    if (B < 1 || H < 8 || W < 16 || H % 8 != 0 || W % 16 != 0 || (int64_t)B * 128 * H * W >= (1ll << 40)) {
        set_error("conv2_train_dgrad: B >= 1, H a multiple of 8, W a multiple of 16");
        return SMK_ERR_UNSUPPORTED;
    }
    return check_launch(launch_conv2_train_dgrad(dz, weight, B, H, W, dx, workspace, (hipStream_t)stream), "conv2_train_dgrad");
}

int64_t smk_conv2_train_wgrad_workspace(void) { return (int64_t)conv2_wgrad_workspace_bytes(conv2_wgrad_streams()); }

int smk_conv2_train_wgrad(const float *dz, const float *a1, int32_t B, int32_t H, int32_t W, float *dw, float *db, void *workspace, void *stream) {
    SMK_REQUIRE(dz && a1 && dw && workspace, "null dz/a1/dw/workspace");
    if (B < 1 || H < 8 || W < 16 || H % 8 != 0 || W % 16 != 0 || (int64_t)B * 128 * H * W >= (1ll << 40)) {
        set_error("conv2_train_wgrad: B >= 1, H a multiple of 8, W a multiple of 16");
        return SMK_ERR_UNSUPPORTED;
    }
    return check_launch(launch_conv2_train_wgrad(dz, a1, B, H, W, dw, db, workspace, (hipStream_t)stream), "conv2_train_wgrad");
}

int64_t smk_bn_train_workspace(int32_t B, int32_t C, int32_t H, int32_t W, int32_t pool) {
    if (B < 1 || C < 1 || H < 1 || W < 1) return 0;
    return bn_train_workspace_floats(B, C, H, W, pool) * (int64_t)sizeof(float);
}

int smk_bn_relu_pool_forward(const float *z, int32_t B, int32_t C, int32_t H, int32_t W, const float *gamma, const float *beta,
                             double eps, int32_t pool, float *out, float *mean, float *var, float *rstd, void *workspace,
                             void *stream) {
    SMK_REQUIRE(z && gamma && beta && out && mean && var && rstd && workspace, "null pointer");
    int rc = bn_check(B, C, H, W, pool);
    if (rc) return rc;
    SMK_REQUIRE((((uintptr_t)z | (uintptr_t)out | (uintptr_t)workspace) & 15) == 0, "16-byte aligned tensors");
    BnTrainArgs a = {};
    a.z = z; a.gamma = gamma; a.beta = beta; a.B = B; a.C = C; a.H = H; a.W = W; a.pool = pool; a.eps = (float)eps;
    a.part = (float *)workspace; a.out = out; a.mean = mean; a.var = var; a.rstd = rstd;
    return check_launch(launch_bn_relu_pool_forward(a, (hipStream_t)stream), "bn_relu_pool_forward");
}

int smk_bn_relu_pool_backward(const float *z, const float *dout, int32_t B, int32_t C, int32_t H, int32_t W, const float *gamma,
                              const float *beta, const float *mean, const float *rstd, int32_t pool, float *dz, float *dgamma,
                              float *dbeta, void *workspace, void *stream) {
    SMK_REQUIRE(z && dout && gamma && beta && mean && rstd && dz && dgamma && dbeta && workspace, "null pointer");
    int rc = bn_check(B, C, H, W, pool);
    if (rc) return rc;
    SMK_REQUIRE((((uintptr_t)z | (uintptr_t)dz | (uintptr_t)dout | (uintptr_t)workspace) & 15) == 0, "16-byte aligned tensors");
    BnTrainArgs a = {};
    a.z = z; a.gamma = gamma; a.beta = beta; a.B = B; a.C = C; a.H = H; a.W = W; a.pool = pool;
    a.part = (float *)workspace; a.mean = const_cast<float *>(mean); a.rstd = const_cast<float *>(rstd);
    a.dout = dout; a.dz = dz; a.dgamma = dgamma; a.dbeta = dbeta;
    return check_launch(launch_bn_relu_pool_backward(a, (hipStream_t)stream), "bn_relu_pool_backward");
}

// The passes of smk_bn_relu_pool_forward / _backward one at a time, for BatchNorm statistics that span several processes
// (models/sync_bn.py all-reduces the per-channel statistics between them).  phase:
//   SMK_BN_STATS        z -> mean / var (biased) / rstd of THIS process's batch                      (needs workspace)
//   SMK_BN_APPLY        out = blockmean(relu(bn(z))) from the GIVEN mean / rstd
//   SMK_BN_BWD_SUMS     dout, z, given mean / rstd -> this process's dgamma = sum dy * zhat, dbeta = sum dy   (needs workspace)
//   SMK_BN_BWD_DZ       dz from the GIVEN (all-reduced) dgamma / dbeta and `count` = elements per channel of the global batch
int smk_bn_relu_pool_phase(int32_t phase, const float *z, const float *dout, int32_t B, int32_t C, int32_t H, int32_t W, const float *gamma,
                           const float *beta, double eps, float *mean, float *var, float *rstd, int32_t pool, float *out, float *dz,
                           float *dgamma, float *dbeta, double count, void *workspace, void *stream) {
    SMK_REQUIRE(z && gamma && beta && mean && rstd, "null pointer");
    int rc = bn_check(B, C, H, W, pool);
    if (rc) return rc;
    SMK_REQUIRE(((uintptr_t)z & 15) == 0, "16-byte aligned z");
    BnTrainArgs a = {};
    a.z = z; a.gamma = gamma; a.beta = beta; a.B = B; a.C = C; a.H = H; a.W = W; a.pool = pool; a.eps = (float)eps;
    a.part = (float *)workspace; a.mean = mean; a.var = var; a.rstd = rstd;
    a.out = out; a.dout = dout; a.dz = dz; a.dgamma = dgamma; a.dbeta = dbeta; a.count = (float)count;
    hipStream_t st = (hipStream_t)stream;
    switch (phase) {
        case SMK_BN_STATS:
            SMK_REQUIRE(var && workspace && ((uintptr_t)workspace & 15) == 0, "stats: var and an aligned workspace");
            return check_launch(launch_bn_stats(a, st), "bn_stats");
        case SMK_BN_APPLY:
            SMK_REQUIRE(out && ((uintptr_t)out & 15) == 0, "apply: aligned out");
            return check_launch(launch_bn_relu_pool_apply(a, st), "bn_relu_pool_apply");
        case SMK_BN_BWD_SUMS:
            SMK_REQUIRE(dout && dgamma && dbeta && workspace && (((uintptr_t)dout | (uintptr_t)workspace) & 15) == 0, "sums: dout, dgamma, dbeta, workspace");
            return check_launch(launch_bn_relu_pool_backward_sums(a, st), "bn_relu_pool_backward_sums");
        case SMK_BN_BWD_DZ:
            SMK_REQUIRE(dout && dz && dgamma && dbeta && count > 0 && (((uintptr_t)dout | (uintptr_t)dz) & 15) == 0, "dz: dout, dz, dgamma, dbeta, count > 0");
            return check_launch(launch_bn_relu_pool_backward_dz(a, st), "bn_relu_pool_backward_dz");
    }
    set_error("bn_relu_pool_phase: unknown phase");
    return SMK_ERR_INVALID;
}

// ------------------------------------------------------------------ LayerNorm
int smk_layernorm(const float *x, int64_t rows, int32_t D, int64_t ldx, const float *weight, const float *bias, double eps,
                  void *y, int64_t ldy, int32_t y_format, void *stream) {
    SMK_REQUIRE(x && y && weight && bias, "null x/y/weight/bias");
    SMK_REQUIRE(rows >= 1 && rows < (1LL << 31) - 4, "1 <= rows < 2^31");
    if (D < 4 || D % 4 != 0 || D > 2048) {
        set_error("layernorm: HIP path is built for D a multiple of 4, at most 2048");
        return SMK_ERR_UNSUPPORTED;
    }
    SMK_REQUIRE(ldx >= D && ldy >= D && ldx % 4 == 0 && ldy % 4 == 0, "row pitches >= D, multiples of 4 floats");
    SMK_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)weight | (uintptr_t)bias) & 15) == 0, "16-byte aligned tensors");
    LayerNormArgs a;
    SMK_REQUIRE(y_format == SMK_FMT_F32 || (y_format == SMK_FMT_SPLIT_BF16 && D % 8 == 0 && ldy == D), "y_format (split: D % 8 == 0, ldy == D)");
    a.x = x; a.y = (float *)y; a.w = weight; a.b = bias; a.ldx = ldx; a.ldy = ldy; a.rows = (int)rows; a.D = D; a.eps = (float)eps;
    a.y_split = y_format == SMK_FMT_SPLIT_BF16;
    return check_launch(launch_layernorm(a, (hipStream_t)stream), "layernorm");
}

int64_t smk_layernorm_bwd_workspace(int32_t D) { return D >= 1 ? (int64_t)LN_BWD_WGS * 2 * D * (int64_t)sizeof(float) : 0; }

int smk_layernorm_backward(const float *x, const float *dy, int64_t rows, int32_t D, int64_t ldx, int64_t ld_dy,
                           const float *weight, double eps, float *dx, int64_t ld_dx, float *dweight, float *dbias,
                           void *workspace, void *stream) {
    SMK_REQUIRE(x && dy && weight && dx && dweight && dbias && workspace, "null pointer");
    SMK_REQUIRE(rows >= 1 && rows < (1LL << 31) - 4, "1 <= rows < 2^31");
    if (D < 4 || D % 4 != 0 || D > 2048) {
        set_error("layernorm: HIP path is built for D a multiple of 4, at most 2048");
        return SMK_ERR_UNSUPPORTED;
    }
    SMK_REQUIRE(ldx >= D && ld_dy >= D && ld_dx >= D && ldx % 4 == 0 && ld_dy % 4 == 0 && ld_dx % 4 == 0, "row pitches >= D, multiples of 4 floats");
    SMK_REQUIRE((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)weight | (uintptr_t)workspace) & 15) == 0, "16-byte aligned tensors");
    LayerNormBwdArgs a;
    a.x = x; a.dy = dy; a.w = weight; a.dx = dx; a.dw = dweight; a.db = dbias; a.part = (float *)workspace;
    a.ldx = ldx; a.lddy = ld_dy; a.lddx = ld_dx; a.rows = (int)rows; a.D = D; a.eps = (float)eps;
    return check_launch(launch_layernorm_bwd(a, (hipStream_t)stream), "layernorm_bwd");
}

// ------------------------------------------------------------------ reconstruction head
int smk_decoder_create(const smk_decoder_weights *w, int32_t device_id, void *stream, smk_decoder **out) {
    SMK_REQUIRE(w && out, "null weights/out");
    const float *const *pp = (const float *const *)w;
    for (size_t k = 0; k < sizeof(*w) / sizeof(float *); ++k) SMK_REQUIRE(pp[k], "null weight pointer");
    DeviceGuard guard(device_id);
    int rc = guard.rc;
    if (rc) return rc;
    smk_decoder *dec = new smk_decoder();
    dec->device = device_id;
    const size_t n = 64 * 32 * 16 + 32 * 16 * 16 + 16 * 9 + 32 + 16 + 16;
    hipError_t e = hipMalloc((void **)&dec->blob, n * sizeof(float));
    if (e != hipSuccess) {
        delete dec;
        set_error(std::string("smk_decoder_create: ") + hipGetErrorString(e));
        return SMK_ERR_HIP;
    }
    float *p = dec->blob;
    dec->d.w1 = p; p += 64 * 32 * 16;
    dec->d.w2 = p; p += 32 * 16 * 16;
    dec->d.w3 = p; p += 16 * 9;
    dec->d.t1 = p; p += 32;
    dec->d.t2 = p; p += 16;
    dec->d.b3 = p;
    rc = check_launch(launch_fold_decoder(*w, dec->d, (hipStream_t)stream), "fold_decoder");
    if (rc) { smk_decoder_destroy(dec); return rc; }
    *out = dec;
    return SMK_OK;
}

int smk_decoder_destroy(smk_decoder *dec) {
    if (!dec) return SMK_OK;
    DeviceGuard guard(dec->device);              // frees run on the handle's device; the caller's device is restored
    if (dec->blob) (void)hipFree(dec->blob);
    delete dec;
    return SMK_OK;
}

int smk_decoder_forward(smk_decoder *dec, const float *tokens, int32_t B, int32_t S, float *tmp1, float *tmp2,
                        float *recon, void *stream) {
    SMK_REQUIRE(dec && tokens && tmp1 && tmp2 && recon, "null dec/tokens/tmp/recon");
    SMK_REQUIRE(B >= 1 && B <= 65535, "1 <= B <= 65535");
    if (S < 16 || S % 16 != 0) {
        set_error("decoder: HIP path is built for token grids whose side is a multiple of 16");
        return SMK_ERR_UNSUPPORTED;
    }
    SMK_REQUIRE(((uintptr_t)tokens & 15) == 0 && ((uintptr_t)tmp1 & 7) == 0 && ((uintptr_t)tmp2 & 7) == 0, "alignment");
    DeviceGuard guard(dec->device);
    int rc = guard.rc;
    if (rc) return rc;
    return check_launch(launch_decoder(dec->d, tokens, B, S, tmp1, tmp2, recon, (hipStream_t)stream), "decoder");
}

}  // extern "C"
#pragma GCC visibility pop
