// Training-mode BatchNorm2d + ReLU + mean pooling of the CNN encoder as HBM-rate kernels (smokephys_net.py:24-32: Conv -> BatchNorm2d
// -> ReLU twice, :87-91: two adaptive average pools = one P x P block mean; under autograd in train.py:88-89).  PyTorch-ROCm runs
// this as MIOpen BatchNorm + ReLU + two pooling kernels, each a full pass over the 2.1 GB of 256 x 256 maps of a batch of 64 (and
// the pooling backward as an atomic scatter); here the forward is two passes over z (statistics; normalise + ReLU + pool, writing
// only the pooled map) and the backward two passes over z plus the write of dz, the ReLU mask being recomputed from z.
//
// Layout: NCHW fp32, planes contiguous.  A workgroup owns one chunk of one (b, c) plane -- 4,096 floats, or 64 rows x 256 columns
// when pooling 8 x 8 -- as float4 per lane: consecutive lanes read consecutive 16- or 32-byte pieces of a row (coalesced), and with
// pooling each thread holds exactly one output cell's P x P block (256 cells per chunk: no cross-lane traffic for the pool).
// Reductions over (B, H, W) are two-stage and deterministic: per-chunk partial sums in a fixed lane/wave order, then one workgroup
// per channel adds the partials in a fixed order.  Variance uses sums shifted by the channel's first element (no cancellation).
#include "norm.h"

namespace smk {

template <int P> struct BnShape {
    static constexpr int CHUNK = P == 8 ? 16384 : 4096;      // floats of one plane per workgroup
    static constexpr int NK = P == 8 ? 16 : 4;               // float4 per thread
};

// float offset (inside the chunk) of this thread's k-th float4, and for P > 1 the chunk-local output cell = threadIdx.x
template <int P>
__device__ __forceinline__ int bn_off(int tid, int k, int W) {
    if (P == 1) return (tid + 256 * k) * 4;
    const int oi = tid >> 5, oj = tid & 31;
    if (P == 8) return (8 * oi + (k >> 1)) * W + 8 * oj + 4 * (k & 1);
    return (4 * oi + k) * W + 4 * oj;                        // P == 4
}

__device__ __forceinline__ float2 wg_sum2(float a, float b) {   // sum over the 256 threads in a fixed order; valid in thread 0
    __shared__ float2 red[4];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        a += __shfl_xor(a, m);
        b += __shfl_xor(b, m);
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = make_float2(a, b);
    __syncthreads();
    return make_float2((red[0].x + red[1].x) + (red[2].x + red[3].x), (red[0].y + red[1].y) + (red[2].y + red[3].y));
}

// chunk id within a channel: b * chunks_per_plane + chunk_in_plane = blockIdx.x; channel = blockIdx.y
template <int P>
__global__ __launch_bounds__(256) void k_bn_stats(const BnTrainArgs a) {
    constexpr int CHUNK = BnShape<P>::CHUNK, NK = BnShape<P>::NK;
    const int c = blockIdx.y, cpp = a.H * a.W / CHUNK;
    const int b = blockIdx.x / cpp, ch = blockIdx.x - b * cpp;
    const float *zp = a.z + ((size_t)b * a.C + c) * a.H * a.W + (size_t)ch * CHUNK;
    const float shift = a.z[(size_t)c * a.H * a.W];                      // the channel's first element (batch 0)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const float4 v = *reinterpret_cast<const float4 *>(zp + bn_off<P>(threadIdx.x, k, a.W));
        const float d0 = v.x - shift, d1 = v.y - shift, d2 = v.z - shift, d3 = v.w - shift;
        s1 += (d0 + d1) + (d2 + d3);
        s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    const float2 t = wg_sum2(s1, s2);
    if (threadIdx.x == 0) reinterpret_cast<float2 *>(a.part)[(size_t)c * gridDim.x + blockIdx.x] = t;
}

// one workgroup per channel: partials -> (mean, var, rstd) or (dgamma, dbeta)
__global__ __launch_bounds__(256) void k_bn_finish(const BnTrainArgs a, int nchunks, int backward) {
    const int c = blockIdx.x;
    const float2 *pp = reinterpret_cast<const float2 *>(a.part) + (size_t)c * nchunks;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < nchunks; i += 256) { s1 += pp[i].x; s2 += pp[i].y; }
    const float2 t = wg_sum2(s1, s2);
    if (threadIdx.x != 0) return;
    if (backward) {
        a.dbeta[c] = t.x;
        a.dgamma[c] = t.y;
    } else {
        const float n = (float)a.B * (float)a.H * (float)a.W;
        const float shift = a.z[(size_t)c * a.H * a.W];
        const float m1 = t.x / n;
        float var = t.y / n - m1 * m1;
        var = var > 0.f ? var : 0.f;
        a.mean[c] = shift + m1;
        a.var[c] = var;
        a.rstd[c] = 1.0f / sqrtf(var + a.eps);
    }
}

template <int P>
__global__ __launch_bounds__(256) void k_bn_relu_pool_fwd(const BnTrainArgs a) {
    constexpr int CHUNK = BnShape<P>::CHUNK, NK = BnShape<P>::NK;
    const int c = blockIdx.y, cpp = a.H * a.W / CHUNK;
    const int b = blockIdx.x / cpp, ch = blockIdx.x - b * cpp;
    const size_t plane = ((size_t)b * a.C + c) * a.H * a.W;
    const float *zp = a.z + plane + (size_t)ch * CHUNK;
    const float sc = a.gamma[c] * a.rstd[c], sh = a.beta[c] - a.mean[c] * sc;      // y = z * sc + sh
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int off = bn_off<P>(threadIdx.x, k, a.W);
        const float4 v = *reinterpret_cast<const float4 *>(zp + off);
        float4 y = make_float4(fmaxf(v.x * sc + sh, 0.f), fmaxf(v.y * sc + sh, 0.f), fmaxf(v.z * sc + sh, 0.f), fmaxf(v.w * sc + sh, 0.f));
        if (P == 1) *reinterpret_cast<float4 *>(a.out + plane + (size_t)ch * CHUNK + off) = y;
        else acc += (y.x + y.y) + (y.z + y.w);
    }
    if (P > 1) {     // this thread's cell: chunk-local (tid >> 5, tid & 31) -> plane row ch * (CHUNK / W / P) + (tid >> 5)
        const int ow = a.W / P, oh = a.H / P;
        const int oi = ch * (CHUNK / (32 * P) / P) + (threadIdx.x >> 5), oj = threadIdx.x & 31;
        a.out[(((size_t)b * a.C + c) * oh + oi) * ow + oj] = acc * (1.0f / (P * P));
    }
}

// dy = dout (spread over the P x P block) where y > 0; MODE 0: partial sums of (dy, dy * zhat); MODE 1: dz
template <int P, int MODE>
__global__ __launch_bounds__(256) void k_bn_relu_pool_bwd(const BnTrainArgs a) {
    constexpr int CHUNK = BnShape<P>::CHUNK, NK = BnShape<P>::NK;
    const int c = blockIdx.y, cpp = a.H * a.W / CHUNK;
    const int b = blockIdx.x / cpp, ch = blockIdx.x - b * cpp;
    const size_t plane = ((size_t)b * a.C + c) * a.H * a.W;
    const float *zp = a.z + plane + (size_t)ch * CHUNK;
    const float mean = a.mean[c], rstd = a.rstd[c], g = a.gamma[c], be = a.beta[c];
    float gcell = 0.f;
    if (P > 1) {
        const int ow = a.W / P, oh = a.H / P;
        const int oi = ch * (CHUNK / (32 * P) / P) + (threadIdx.x >> 5), oj = threadIdx.x & 31;
        gcell = a.dout[(((size_t)b * a.C + c) * oh + oi) * ow + oj] * (1.0f / (P * P));
    }
    float k1 = 0.f, k2 = 0.f;
    if (MODE == 1) {
        const float n = a.count > 0.f ? a.count : (float)a.B * (float)a.H * (float)a.W;
        k1 = a.dbeta[c] / n;
        k2 = a.dgamma[c] / n;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int off = bn_off<P>(threadIdx.x, k, a.W);
        const float4 v = *reinterpret_cast<const float4 *>(zp + off);
        float4 go = make_float4(gcell, gcell, gcell, gcell);
        if (P == 1) go = *reinterpret_cast<const float4 *>(a.dout + plane + (size_t)ch * CHUNK + off);
        const float zv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {go.x, go.y, go.z, go.w};
        float r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float zh = (zv[i] - mean) * rstd;
            const float dy = (zh * g + be) > 0.f ? gv[i] : 0.f;
            if (MODE == 0) { s1 += dy; s2 += dy * zh; }
            else r[i] = g * rstd * (dy - k1 - zh * k2);
        }
        if (MODE == 1) *reinterpret_cast<float4 *>(a.dz + plane + (size_t)ch * CHUNK + off) = make_float4(r[0], r[1], r[2], r[3]);
    }
    if (MODE == 0) {
        const float2 t = wg_sum2(s1, s2);
        if (threadIdx.x == 0) reinterpret_cast<float2 *>(a.part)[(size_t)c * gridDim.x + blockIdx.x] = t;
    }
}

static int bn_chunks(const BnTrainArgs &a) {
    const int chunk = a.pool == 8 ? 16384 : 4096;
    return a.B * (a.H * a.W / chunk);
}

long long bn_train_workspace_floats(int B, int C, int H, int W, int pool) {
    const int chunk = pool == 8 ? 16384 : 4096;
    return 2LL * C * B * ((long long)H * W / chunk);
}

hipError_t launch_bn_relu_pool_forward(const BnTrainArgs &a, hipStream_t st) {
    const int nch = bn_chunks(a);
    dim3 grid(nch, a.C), block(256);
    switch (a.pool) {
        case 1: hipLaunchKernelGGL(k_bn_stats<1>, grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL(k_bn_stats<4>, grid, block, 0, st, a); break;
        case 8: hipLaunchKernelGGL(k_bn_stats<8>, grid, block, 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(k_bn_finish, dim3(a.C), block, 0, st, a, nch, 0);
    switch (a.pool) {
        case 1: hipLaunchKernelGGL(k_bn_relu_pool_fwd<1>, grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL(k_bn_relu_pool_fwd<4>, grid, block, 0, st, a); break;
        case 8: hipLaunchKernelGGL(k_bn_relu_pool_fwd<8>, grid, block, 0, st, a); break;
    }
    return hipGetLastError();
}

hipError_t launch_bn_stats(const BnTrainArgs &a, hipStream_t st) {
    const int nch = bn_chunks(a);
    dim3 grid(nch, a.C), block(256);
    switch (a.pool) {
        case 1: hipLaunchKernelGGL(k_bn_stats<1>, grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL(k_bn_stats<4>, grid, block, 0, st, a); break;
        case 8: hipLaunchKernelGGL(k_bn_stats<8>, grid, block, 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(k_bn_finish, dim3(a.C), block, 0, st, a, nch, 0);
    return hipGetLastError();
}

hipError_t launch_bn_relu_pool_apply(const BnTrainArgs &a, hipStream_t st) {
    dim3 grid(bn_chunks(a), a.C), block(256);
    switch (a.pool) {
        case 1: hipLaunchKernelGGL(k_bn_relu_pool_fwd<1>, grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL(k_bn_relu_pool_fwd<4>, grid, block, 0, st, a); break;
        case 8: hipLaunchKernelGGL(k_bn_relu_pool_fwd<8>, grid, block, 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_bn_relu_pool_backward_sums(const BnTrainArgs &a, hipStream_t st) {
    const int nch = bn_chunks(a);
    dim3 grid(nch, a.C), block(256);
    switch (a.pool) {
        case 1: hipLaunchKernelGGL((k_bn_relu_pool_bwd<1, 0>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((k_bn_relu_pool_bwd<4, 0>), grid, block, 0, st, a); break;
        case 8: hipLaunchKernelGGL((k_bn_relu_pool_bwd<8, 0>), grid, block, 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(k_bn_finish, dim3(a.C), block, 0, st, a, nch, 1);
    return hipGetLastError();
}

hipError_t launch_bn_relu_pool_backward_dz(const BnTrainArgs &a, hipStream_t st) {
    dim3 grid(bn_chunks(a), a.C), block(256);
    switch (a.pool) {
        case 1: hipLaunchKernelGGL((k_bn_relu_pool_bwd<1, 1>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((k_bn_relu_pool_bwd<4, 1>), grid, block, 0, st, a); break;
        case 8: hipLaunchKernelGGL((k_bn_relu_pool_bwd<8, 1>), grid, block, 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_bn_relu_pool_backward(const BnTrainArgs &a, hipStream_t st) {
    const int nch = bn_chunks(a);
    dim3 grid(nch, a.C), block(256);
    switch (a.pool) {
        case 1: hipLaunchKernelGGL((k_bn_relu_pool_bwd<1, 0>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((k_bn_relu_pool_bwd<4, 0>), grid, block, 0, st, a); break;
        case 8: hipLaunchKernelGGL((k_bn_relu_pool_bwd<8, 0>), grid, block, 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(k_bn_finish, dim3(a.C), block, 0, st, a, nch, 1);
    switch (a.pool) {
        case 1: hipLaunchKernelGGL((k_bn_relu_pool_bwd<1, 1>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((k_bn_relu_pool_bwd<4, 1>), grid, block, 0, st, a); break;
        case 8: hipLaunchKernelGGL((k_bn_relu_pool_bwd<8, 1>), grid, block, 0, st, a); break;
    }
    return hipGetLastError();
}

}  // namespace smk
