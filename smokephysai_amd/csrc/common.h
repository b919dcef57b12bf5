// Shared host/device helpers of libsmokehip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/smokehip.h"

namespace smk {

void set_error(const std::string &msg);

#define SMK_HIP_TRY(expr)                                                                         \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            smk::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                    \
            return SMK_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

#define SMK_REQUIRE(cond, msg)                                                                    \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            smk::set_error(std::string("invalid argument: ") + (msg));                            \
            return SMK_ERR_INVALID;                                                               \
        }                                                                                         \
    } while (0)

// Geometry of one batch of grids. Plane strides in floats.
struct Geom {
    int B, H, W;
    int pc, pv;          // row pitches of (u,p,density) and of v
    size_t su, sv, sc;   // plane strides: u (H+1 rows of pc), v (H rows of pv), cell fields (H rows of pc)
    float dt;            // (float)dt
    float coef_uv;       // (float)(dt*viscosity)
    float coef_d;        // (float)(dt*(viscosity*0.1))
};

struct StateView {
    float *u, *v, *p, *d;
};

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Launch-time state is PER DEVICE (a process may hold handles on several GPUs): the CU count, and "has this kernel's dynamic-LDS
// limit been raised / what did the occupancy query say" keyed by (kernel, current device).  Thread-safe (api.hip).
int device_num_cu();                                         // CUs of the CURRENT device (cached)
// One-time per-(key, current device) setup (hipFuncSetAttribute before a kernel's first launch): first_use_begin returns true exactly once
// and then HOLDS the per-device lock until first_use_end, so a second host thread cannot launch before the attribute is set.
bool first_use_begin(const void *key);
void first_use_end(const void *key);
template <class F>
inline void once_per_device(const void *key, F &&setup) {
    if (first_use_begin(key)) {
        setup();
        first_use_end(key);
    }
}
int device_cached_int(const void *key, int (*compute)());    // compute() once per (key, current device), then the cached value

}  // namespace smk
