// Small kernels of SmokePhysNet's transformer body that are not GEMMs (the GEMMs: linear.hip).
#include "transformer.h"

namespace smk {

// The chaos term of one ChaosAttention layer, folded into Q (chaos_attention.py:39-66 lorenz_system / generate_chaos_field,
// :85-100 chaos_proj, chaos_gate): per batch element three N(0,1) draws * 0.1 seed five explicit-Euler Lorenz steps; each
// state s_t gives C_t = chaos_proj(s_t) [D], g_t = sigmoid(chaos_gate(C_t)) and the addend strength * g_t * C_t that row
// l = t (mod 5) of the sequence adds to its query.  In the reference this is ~90 one-element-per-batch elementwise
// launches per layer; here one workgroup per batch element.  fp32 with the reference's operation order (no contraction).
__global__ __launch_bounds__(256) void k_chaos_addend(const ChaosAddendArgs a) {
    __shared__ float red[5][4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float x = a.noise[b] * 0.1f, y = a.noise[a.B + b] * 0.1f, z = a.noise[2 * a.B + b] * 0.1f;
    float sx[5], sy[5], sz[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const float dx = a.sigma * (y - x);
        const float dy = x * (a.rho - z) - y;
        const float dz = x * y - a.beta * z;
        x = x + a.dt * dx;
        y = y + a.dt * dy;
        z = z + a.dt * dz;
        sx[t] = x; sy[t] = y; sz[t] = z;
    }
    float part[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int d = tid; d < a.D; d += 256) {
        const float w0 = a.proj_w[3 * d], w1 = a.proj_w[3 * d + 1], w2 = a.proj_w[3 * d + 2], pb = a.proj_b[d], gw = a.gate_w[d];
#pragma unroll
        for (int t = 0; t < 5; ++t) part[t] += gw * (((sx[t] * w0 + sy[t] * w1) + sz[t] * w2) + pb);
    }
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        float v = part[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[t][wave] = v;
    }
    __syncthreads();
    float g[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const float s = ((red[t][0] + red[t][1]) + (red[t][2] + red[t][3])) + a.gate_b[0];
        g[t] = a.strength * (1.0f / (1.0f + expf(-s)));
    }
    for (int d = tid; d < a.D; d += 256) {
        const float w0 = a.proj_w[3 * d], w1 = a.proj_w[3 * d + 1], w2 = a.proj_w[3 * d + 2], pb = a.proj_b[d];
#pragma unroll
        for (int t = 0; t < 5; ++t)
            a.addend[((size_t)b * 5 + t) * a.D + d] = g[t] * (((sx[t] * w0 + sy[t] * w1) + sz[t] * w2) + pb);
    }
}

hipError_t launch_chaos_addend(const ChaosAddendArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(k_chaos_addend, dim3(a.B), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace smk
