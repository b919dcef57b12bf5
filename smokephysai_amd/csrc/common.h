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

// ---- device helpers shared by the stencil kernel files
// Workgroups are dealt round-robin over the 8 XCDs (ids i and i + 8 share an XCD and its 4 MiB L2: MI355X_MICROARCH.md, speed only).  A
// kernel whose tiles overlap (halos) or read each other's edge lines wants NEIGHBOURING tiles on one XCD: this maps the dispatch id to a
// logical tile id such that XCD c works through one contiguous range of tiles, in order.  A bijection on [0, n) for every n.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned id, unsigned n) {
    const unsigned c = id & 7u, q = n >> 3, r = n & 7u;
    return c * q + (c < r ? c : r) + (id >> 3);
}
// lane i <- lane i+1 of x; lane 63 <- `last` (DPP wave_shl:1 with the destination preloaded)
__device__ __forceinline__ float shl1_with(float x, float last) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, last), __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
}

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
