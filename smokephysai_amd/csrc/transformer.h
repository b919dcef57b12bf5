#pragma once
#include "common.h"

namespace smk {

struct ChaosAddendArgs {
    const float *noise;                  // [3][B] standard-normal draws (x0, y0, z0 before the 0.1 scale)
    const float *proj_w, *proj_b;        // chaos_proj: [D][3], [D]
    const float *gate_w, *gate_b;        // chaos_gate: [D] (= [1][D]), [1]
    float *addend;                       // [B][5][D]
    int B, D;
    float strength, sigma, rho, beta, dt;
};
hipError_t launch_chaos_addend(const ChaosAddendArgs &a, hipStream_t st);

}  // namespace smk
