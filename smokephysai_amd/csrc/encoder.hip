// SmokePhysNet.input_encoder + pooling as ONE fused kernel per precision (gfx950).
//
// Reference: /root/reference/src/models/smokephys_net.py:24-32 (Conv7x7 1->64, BN, ReLU, Conv3x3 64->128, BN, ReLU,
// AdaptiveAvgPool2d(input_dim)) and :90-91 (adaptive_avg_pool2d -> 32x32).  Eval-mode BN is folded to a per-channel
// scale/shift; for H a multiple of 32 and input_dim a multiple/divisor of H the two adaptive pools compose to an
// (H/32)^2 block mean.
//
// Tiling (one workgroup = 512 threads = 8 waves, one wave per output row of the tile):
//   output tile 8 rows x 32 cols x 128 channels;  a1 (conv1 activations) for the 10 x 34 halo tile x 64 channels is
//   computed on the fly into LDS (never touches HBM); conv2 is an implicit GEMM  D[pixel][o] = sum_k A[pixel][k] B[k][o]
//   with k = (tap, channel): A fragments are shifted reads of the LDS a1 tile (no im2col), B = conv2 weights re-laid out
//   [tap][c][o] and streamed tap by tap through a double-buffered LDS stage.
//   fp32 path: v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain).
#include "encoder.h"

namespace smk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ENC_TH = 8, ENC_TW = 32;          // output tile
constexpr int ENC_AH = ENC_TH + 2, ENC_AW = ENC_TW + 2;   // a1 halo tile (conv2 3x3)
constexpr int ENC_XH = ENC_TH + 8, ENC_XW = ENC_TW + 8;   // x halo tile (+ conv1 7x7)
constexpr int ENC_ACS = ENC_AH * ENC_AW;        // a1 channel stride in LDS (floats)

// ---------------------------------------------------------------- weight folding / re-layout
// s = bn_w / sqrt(var + eps), t = (conv_b - mean) * s + bn_b       (BN eval, eps = 1e-5)
__global__ void k_fold_weights(smk_encoder_weights w, EncoderDev e) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 64) {
        float s = w.bn1_w[t] / sqrtf(w.bn1_var[t] + 1e-5f);
        e.s1[t] = s;
        e.t1[t] = (w.conv1_b[t] - w.bn1_mean[t]) * s + w.bn1_b[t];
    }
    if (t < 128) {
        float s = w.bn2_w[t] / sqrtf(w.bn2_var[t] + 1e-5f);
        e.s2[t] = s;
        e.t2[t] = (w.conv2_b[t] - w.bn2_mean[t]) * s + w.bn2_b[t];
    }
    if (t < 64 * 49) e.w1[t] = w.conv1_w[t];                       // [c][49]
    if (t < 9 * 64 * 128) {                                         // conv2_w [o][c][3][3] -> w2t [tap][c][o]
        int o = t % 128, c = (t / 128) % 64, tap = t / (128 * 64);
        e.w2t[t] = w.conv2_w[((size_t)o * 64 + c) * 9 + tap];
    }
}

hipError_t launch_fold_weights(const smk_encoder_weights &w, const EncoderDev &e, hipStream_t st) {
    int n = 9 * 64 * 128;
    hipLaunchKernelGGL(k_fold_weights, dim3((n + 255) / 256), dim3(256), 0, st, w, e);
    return hipGetLastError();
}

// conv1 + BN + ReLU for one pixel and NC consecutive channels starting at c0 (wave-uniform), from a 7x7 patch in regs.
template <int NC>
__device__ __forceinline__ void conv1_pixel(const float (&patch)[49], const float *__restrict__ w1,
                                            const float *__restrict__ s1, const float *__restrict__ t1, int c0,
                                            float (&out)[NC]) {
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
        const float *wc = w1 + (c0 + cc) * 49;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 49; ++t) acc = fmaf(patch[t], wc[t], acc);
        float y = fmaf(acc, s1[c0 + cc], t1[c0 + cc]);
        out[cc] = y > 0.f ? y : 0.f;
    }
}

// ---------------------------------------------------------------- parity hook: conv1 activations to HBM
__global__ void k_conv1_only(const float *frames, int64_t fstride, int H, int W, EncoderDev e, float *act) {
    int b = blockIdx.z;
    int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    if (i >= H || j >= W) return;
    const float *x = frames + (size_t)b * fstride;
    float patch[49];
#pragma unroll
    for (int ki = 0; ki < 7; ++ki)
#pragma unroll
        for (int kj = 0; kj < 7; ++kj) {
            int ii = i + ki - 3, jj = j + kj - 3;
            patch[ki * 7 + kj] = (ii >= 0 && ii < H && jj >= 0 && jj < W) ? x[(size_t)ii * W + jj] : 0.f;
        }
    for (int c0 = 0; c0 < 64; c0 += 16) {
        float o[16];
        conv1_pixel<16>(patch, e.w1, e.s1, e.t1, c0, o);
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) act[(((size_t)b * 64 + c0 + cc) * H + i) * W + j] = o[cc];
    }
}

hipError_t launch_conv1_only(const float *frames, int64_t fstride, int B, int H, int W, const EncoderDev &e, float *act,
                             hipStream_t st) {
    dim3 grid(cdiv(W, 64), cdiv(H, 4), B), block(64, 4);
    hipLaunchKernelGGL(k_conv1_only, grid, block, 0, st, frames, fstride, H, W, e, act);
    return hipGetLastError();
}

// ---------------------------------------------------------------- fused encoder, fp32 MFMA
// LDS carve (floats): xs [16][40] | a1s [64][10][34] | w2s 2 x [64][128]  (= 2560 + 87040 + 65536 B = 155,136 B)
// The epilogue reuses the a1s/w2s region as a2s [256 pixels][128+1 channels... see below].
constexpr int LDS_XS = ENC_XH * ENC_XW;                  // 640
constexpr int LDS_A1 = 64 * ENC_ACS;                     // 21760
constexpr int LDS_W2 = 64 * 128;                         // 8192 per buffer
constexpr int LDS_F32_TOTAL = LDS_XS + LDS_A1 + 2 * LDS_W2;
constexpr int A2_PITCH = 129;                            // a2s [pixel][o], +1 pad

template <int PS>
__global__ __launch_bounds__(512) void k_encoder_f32(const float *__restrict__ frames, int64_t fstride, int H, int W,
                                                     EncoderDev e, float *__restrict__ features) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *xs = lds, *a1s = lds + LDS_XS, *w2s = lds + LDS_XS + LDS_A1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, r0 = blockIdx.y * ENC_TH, c0 = blockIdx.x * ENC_TW;
    const float *x = frames + (size_t)b * fstride;

    // ---- stage the x halo tile (zero outside the image = conv1's padding)
    for (int k = tid; k < LDS_XS; k += 512) {
        int ii = r0 - 4 + k / ENC_XW, jj = c0 - 4 + k % ENC_XW;
        xs[k] = (ii >= 0 && ii < H && jj >= 0 && jj < W) ? x[(size_t)ii * W + jj] : 0.f;
    }
    // prefetch tap-0 weights into registers (4 x float4 per thread = 32 KB per workgroup)
    float4 wreg[4];
    {
        const float4 *src = reinterpret_cast<const float4 *>(e.w2t);
#pragma unroll
        for (int q = 0; q < 4; ++q) wreg[q] = src[q * 512 + tid];
    }
    __syncthreads();

    // ---- conv1 + BN + ReLU into a1s: 6 pixel chunks of 64 x 4 channel quarters = 24 wave tasks over 8 waves
    for (int task = wave; task < 24; task += 8) {
        const int chunk = task % 6, cq = task / 6;            // wave-uniform
        const int pix = chunk * 64 + lane;
        if (pix < ENC_ACS) {
            const int ar = pix / ENC_AW, ac = pix % ENC_AW;     // a1 halo coords; image coords (r0-1+ar, c0-1+ac)
            const int ii = r0 - 1 + ar, jj = c0 - 1 + ac;
            float o[16];
            if (ii >= 0 && ii < H && jj >= 0 && jj < W) {
                float patch[49];
#pragma unroll
                for (int ki = 0; ki < 7; ++ki)
#pragma unroll
                    for (int kj = 0; kj < 7; ++kj) patch[ki * 7 + kj] = xs[(ar + ki) * ENC_XW + ac + kj];
                conv1_pixel<16>(patch, e.w1, e.s1, e.t1, cq * 16, o);
            } else {
#pragma unroll
                for (int cc = 0; cc < 16; ++cc) o[cc] = 0.f;      // conv2's zero padding
            }
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) a1s[(cq * 16 + cc) * ENC_ACS + pix] = o[cc];
        }
    }
    // tap-0 weights -> LDS buffer 0
    {
        float4 *dst = reinterpret_cast<float4 *>(w2s);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q * 512 + tid] = wreg[q];
    }
    __syncthreads();

    // ---- conv2 implicit GEMM: wave -> tile row `wave`, 32 pixels x 128 channels = 4 accumulators of 32x32
    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int l31 = lane & 31, hi = lane >> 5;
    for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) {
            const float4 *src = reinterpret_cast<const float4 *>(e.w2t + (size_t)(tap + 1) * LDS_W2);
#pragma unroll
            for (int q = 0; q < 4; ++q) wreg[q] = src[q * 512 + tid];
        }
        const int ki = tap / 3, kj = tap % 3;
        const float *ap = a1s + hi * ENC_ACS + (wave + ki) * ENC_AW + l31 + kj;
        const float *bp = w2s + (tap & 1) * LDS_W2 + hi * 128 + l31;
#pragma unroll 4
        for (int cp = 0; cp < 32; ++cp) {
            float a = ap[cp * 2 * ENC_ACS];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float bv = bp[cp * 2 * 128 + n * 32];
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[n], 0, 0, 0);
            }
        }
        if (tap + 1 < 9) {
            float4 *dst = reinterpret_cast<float4 *>(w2s + ((tap + 1) & 1) * LDS_W2);
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q * 512 + tid] = wreg[q];
        }
        __syncthreads();
    }

    // ---- epilogue: BN2 + ReLU, stage a2 [pixel][o] in LDS (a1s/w2s are dead), block-mean pool, store
    float *a2s = lds;                                        // 256 x 129 floats = 132,096 B
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int o = n * 32 + l31;
        const float s = e.s2[o], t = e.t2[o];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = (r & 3) + 8 * (r >> 2) + 4 * hi;   // C/D map of 32x32 MFMA: row = pixel
            float y = fmaf(acc[n][r], s, t);
            a2s[(wave * ENC_TW + col) * A2_PITCH + o] = y > 0.f ? y : 0.f;
        }
    }
    __syncthreads();
    constexpr int CELLS_R = ENC_TH / PS, CELLS_C = ENC_TW / PS;   // pooled cells in this tile
    const int OW = 32, OHW = 32 * 32;
    for (int k = tid; k < CELLS_R * CELLS_C * 128; k += 512) {
        const int o = k & 127, cell = k >> 7, cr = cell / CELLS_C, cc = cell % CELLS_C;
        float sum = 0.f;
        for (int rr = 0; rr < PS; ++rr)
            for (int q = 0; q < PS; ++q) sum += a2s[((cr * PS + rr) * ENC_TW + cc * PS + q) * A2_PITCH + o];
        const int pi = r0 / PS + cr, pj = c0 / PS + cc;
        features[((size_t)b * 128 + o) * OHW + pi * OW + pj] = sum * (1.0f / (PS * PS));
    }
}

hipError_t launch_encoder_f32(const float *frames, int64_t fstride, int B, int H, int W, const EncoderDev &e,
                              float *features, hipStream_t st) {
    const int PS = H / 32;
    dim3 grid(W / ENC_TW, H / ENC_TH, B), block(512);
    size_t lds_bytes = sizeof(float) * (size_t)(LDS_F32_TOTAL > 256 * A2_PITCH ? LDS_F32_TOTAL : 256 * A2_PITCH);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_encoder_f32<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute((const void *)k_encoder_f32<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute((const void *)k_encoder_f32<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    switch (PS) {
        case 2: hipLaunchKernelGGL(k_encoder_f32<2>, grid, block, lds_bytes, st, frames, fstride, H, W, e, features); break;
        case 4: hipLaunchKernelGGL(k_encoder_f32<4>, grid, block, lds_bytes, st, frames, fstride, H, W, e, features); break;
        case 8: hipLaunchKernelGGL(k_encoder_f32<8>, grid, block, lds_bytes, st, frames, fstride, H, W, e, features); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace smk
