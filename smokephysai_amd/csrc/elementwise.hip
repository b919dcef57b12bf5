// Fused element-wise kernels of the FFN block in training (see elementwise.h).  One float4 per lane and step, grid-stride.
#include "elementwise.h"

namespace smk {

// exact-erf GELU (nn.GELU() default, smokephys_net.py:155) and its derivative; erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7), as the
// linear kernel's epilogue evaluates it in eval mode
__device__ __forceinline__ void erf_parts(float v, float &erf_abs, float &gauss) {
    const float z = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    gauss = __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);      // exp(-v^2 / 2)
    erf_abs = fmaf(-(p * t), gauss, 1.0f);
}
__device__ __forceinline__ float gelu_f(float v) {
    float ea, g;
    erf_parts(v, ea, g);
    return 0.5f * v * (1.0f + copysignf(ea, v));
}
__device__ __forceinline__ float gelu_grad(float v) {                         // Phi(v) + v * phi(v)
    float ea, g;
    erf_parts(v, ea, g);
    return fmaf(v * 0.39894228040143267794f, g, 0.5f * (1.0f + copysignf(ea, v)));
}

// keep-mask bits for the four elements of float4 number i: one 64-bit mix (splitmix64 finaliser) of (seed, i), 16 bits per element
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}
struct Keep4 { float k[4]; };
__device__ __forceinline__ Keep4 keep4(unsigned long long seed, long long i, unsigned thresh, float scale) {
    const unsigned long long r = mix64(seed + 0x9e3779b97f4a7c15ull * (unsigned long long)(i + 1));
    Keep4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.k[j] = ((unsigned)(r >> (16 * j)) & 0xffffu) >= thresh ? scale : 0.f;
    return o;
}

template <int OP>   // 0 gelu_dropout fwd, 1 gelu_dropout bwd, 2 dropout_add fwd, 3 dropout bwd
__global__ __launch_bounds__(256) void k_elementwise(const EltArgs e) {
    const long long n4 = e.n / 4;
    const unsigned thresh = (unsigned)(e.p * 65536.0f + 0.5f);               // keep when the 16-bit draw >= p * 2^16
    const float scale = 1.0f / (1.0f - (float)thresh * (1.0f / 65536.0f));   // 1 / (1 - p) for the p actually applied
    const float4 *a4 = reinterpret_cast<const float4 *>(e.a), *b4 = reinterpret_cast<const float4 *>(e.b);
    float4 *o4 = reinterpret_cast<float4 *>(e.out);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 av = a4[i];
        const float a[4] = {av.x, av.y, av.z, av.w};
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (OP == 1 || OP == 2) {
            const float4 bv = b4[i];
            b[0] = bv.x; b[1] = bv.y; b[2] = bv.z; b[3] = bv.w;
        }
        const Keep4 m = keep4(e.seed, i, thresh, scale);
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (OP == 0) r[j] = gelu_f(a[j]) * m.k[j];
            else if (OP == 1) r[j] = b[j] * m.k[j] * gelu_grad(a[j]);
            else if (OP == 2) r[j] = fmaf(a[j], m.k[j], b[j]);
            else r[j] = a[j] * m.k[j];
        }
        o4[i] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

template <int OP>
static hipError_t launch_elt(const EltArgs &e, hipStream_t st) {
    const long long n4 = e.n / 4;
    long long blocks = (n4 + 255) / 256;
    const long long cap = (long long)device_num_cu() * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_elementwise<OP>, dim3((unsigned)blocks), dim3(256), 0, st, e);
    return hipGetLastError();
}
hipError_t launch_gelu_dropout_fwd(const EltArgs &e, hipStream_t st) { return launch_elt<0>(e, st); }
hipError_t launch_gelu_dropout_bwd(const EltArgs &e, hipStream_t st) { return launch_elt<1>(e, st); }
hipError_t launch_dropout_add_fwd(const EltArgs &e, hipStream_t st) { return launch_elt<2>(e, st); }
hipError_t launch_dropout_bwd(const EltArgs &e, hipStream_t st) { return launch_elt<3>(e, st); }

}  // namespace smk
