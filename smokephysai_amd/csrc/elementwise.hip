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


// ---------------------------------------------------------------- the owner's pass of the direct gradient exchange
// out[i] = (((s_0[i] + s_1[i]) + ...) + s_{W-1}[i]) / W over W shards (one per source rank, `stride` elements apart), the sum in rank
// order in fp32 -- the same order on every rank and run -- and a true divide, written in the wire dtype (fp32 or bf16).  One read of
// every shard and one write: replaces zeros + copy_ + W casts + W adds + divide + cast of the eager form.  W = 1 is the dtype converter.
template <class TI, class TO>
__global__ __launch_bounds__(256) void k_reduce_shards(const TI *__restrict__ in, int world, long long n, long long stride, TO *__restrict__ out) {
    const float fw = (float)world;
    const long long step = (long long)gridDim.x * 256 * 4;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += step) {
        float acc[4];
        if (i + 4 <= n) {
            auto load4 = [&](const TI *p) {
                if constexpr (sizeof(TI) == 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(p);
                    acc[0] = v.x; acc[1] = v.y; acc[2] = v.z; acc[3] = v.w;
                } else {
                    const uint2 v = *reinterpret_cast<const uint2 *>(p);
                    acc[0] = __uint_as_float(v.x << 16); acc[1] = __uint_as_float(v.x & 0xffff0000u);
                    acc[2] = __uint_as_float(v.y << 16); acc[3] = __uint_as_float(v.y & 0xffff0000u);
                }
            };
            float sum[4];
            load4(in + i);
            for (int k = 0; k < 4; ++k) sum[k] = acc[k];
            for (int r = 1; r < world; ++r) {
                load4(in + r * stride + i);
                for (int k = 0; k < 4; ++k) sum[k] = sum[k] + acc[k];
            }
            if (world > 1)
                for (int k = 0; k < 4; ++k) sum[k] = __fdiv_rn(sum[k], fw);
            if constexpr (sizeof(TO) == 4) {
                *reinterpret_cast<float4 *>(out + i) = make_float4(sum[0], sum[1], sum[2], sum[3]);
            } else {
                __bf16 h[4];
                for (int k = 0; k < 4; ++k) h[k] = (__bf16)sum[k];          // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
                *reinterpret_cast<uint2 *>(out + i) = *reinterpret_cast<const uint2 *>(h);
            }
        } else {
            for (long long j = i; j < n; ++j) {                              // ragged tail (n % 4)
                auto ld = [&](const TI *p) -> float {
                    if constexpr (sizeof(TI) == 4) return *reinterpret_cast<const float *>(p);
                    else return __uint_as_float((unsigned)*reinterpret_cast<const unsigned short *>(p) << 16);
                };
                float sm = ld(in + j);
                for (int r = 1; r < world; ++r) sm = sm + ld(in + r * stride + j);
                if (world > 1) sm = __fdiv_rn(sm, fw);
                if constexpr (sizeof(TO) == 4) out[j] = sm;
                else { const __bf16 h = (__bf16)sm; out[j] = *reinterpret_cast<const unsigned short *>(&h); }
            }
        }
    }
}

hipError_t launch_reduce_shards(const void *in, int in_bf16, int world, long long n, long long stride, void *out, int out_bf16, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    long long blocks = (n / 4 + 255) / 256;
    const long long cap = (long long)device_num_cu() * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks), block(256);
    if (!in_bf16 && !out_bf16) hipLaunchKernelGGL((k_reduce_shards<float, float>), grid, block, 0, st, (const float *)in, world, n, stride, (float *)out);
    else if (!in_bf16) hipLaunchKernelGGL((k_reduce_shards<float, unsigned short>), grid, block, 0, st, (const float *)in, world, n, stride, (unsigned short *)out);
    else if (!out_bf16) hipLaunchKernelGGL((k_reduce_shards<unsigned short, float>), grid, block, 0, st, (const unsigned short *)in, world, n, stride, (float *)out);
    else hipLaunchKernelGGL((k_reduce_shards<unsigned short, unsigned short>), grid, block, 0, st, (const unsigned short *)in, world, n, stride, (unsigned short *)out);
    return hipGetLastError();
}

}  // namespace smk
