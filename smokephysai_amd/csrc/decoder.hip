// SmokePhysNet.reconstruction_head (smokephys_net.py:57-66,117-118) in eval mode:
//   ConvTranspose2d(64,32,k4,s2,p1) + BN + ReLU -> ConvTranspose2d(32,16,k4,s2,p1) + BN + ReLU -> Conv2d(16,1,3,p1) -> Sigmoid
// as three direct fp32 kernels (exact fp32 FMAs like the reference's; BN folded into weights / shift at handle creation).
// 0.139 GFLOP per frame: not matrix-core work -- the point is to replace ~10 MIOpen / elementwise launches (about 1 ms at
// batch 64, 150 us of launch-bound kernels at batch 1) by three.
//
// Stride-2 transposed conv as four 2x2 sub-pixel convolutions: the thread that owns input position (i, j) produces the output
// quad (2i+py, 2j+px).  y = 2 iy - 1 + ky, so row parity 0 takes (ky 1, iy i) and (ky 3, iy i-1), parity 1 takes (ky 0, iy i+1)
// and (ky 2, iy i); columns likewise: 16 FMAs per (input channel, output channel) on the 3x3 neighbourhood of (i, j).
// Weights are indexed by wave-uniform values only, so they travel through the scalar path (s_load -> SGPR FMA operands).
#include "decoder.h"

#include <stdlib.h>

namespace smk {

__global__ void k_fold_decoder(smk_decoder_weights w, DecoderDev d) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 32) {
        const float s = w.bn1_w[t] / sqrtf(w.bn1_var[t] + 1e-5f);
        d.t1[t] = (w.ct1_b[t] - w.bn1_mean[t]) * s + w.bn1_b[t];
    }
    if (t < 16) {
        const float s = w.bn2_w[t] / sqrtf(w.bn2_var[t] + 1e-5f);
        d.t2[t] = (w.ct2_b[t] - w.bn2_mean[t]) * s + w.bn2_b[t];
    }
    if (t < 64 * 32 * 16) {                                   // ct1_w [c][o][4][4]
        const int o = (t >> 4) & 31;
        d.w1[t] = w.ct1_w[t] * (w.bn1_w[o] / sqrtf(w.bn1_var[o] + 1e-5f));
    }
    if (t < 32 * 16 * 16) {                                   // ct2_w [c][o][4][4]
        const int o = (t >> 4) & 15;
        d.w2[t] = w.ct2_w[t] * (w.bn2_w[o] / sqrtf(w.bn2_var[o] + 1e-5f));
    }
    if (t < 16 * 9) d.w3[t] = w.conv_w[t];
    if (t == 0) d.b3[0] = w.conv_b[0];
}

hipError_t launch_fold_decoder(const smk_decoder_weights &w, const DecoderDev &d, hipStream_t st) {
    hipLaunchKernelGGL(k_fold_decoder, dim3((64 * 32 * 16 + 255) / 256), dim3(256), 0, st, w, d);
    return hipGetLastError();
}

constexpr int DC_T = 16;                                      // input positions per tile side
constexpr int DC_CC = 16;                                     // channels staged per chunk
constexpr int DC_PW = DC_T + 3;                               // LDS row pitch (18 used + 1 pad)

// TOK: input is token-major [B][H*W][CIN] (what output_decoder writes), else [B][CIN][H][W].
template <int CIN, int COUT, int OG, bool TOK>
__global__ __launch_bounds__(256) void k_convt4s2(const float *__restrict__ in, const float *__restrict__ wf, const float *__restrict__ shift,
                                                 float *__restrict__ out, int H, int W) {
    __shared__ float tile[DC_CC][DC_T + 2][DC_PW];
    const int tid = threadIdx.x, tj = tid & 15, ti = tid >> 4;
    const int tiles_x = W / DC_T;
    const int i0 = (blockIdx.x / tiles_x) * DC_T, j0 = (blockIdx.x % tiles_x) * DC_T;
    const int og = blockIdx.y, b = blockIdx.z;
    const float *inb = in + (size_t)b * CIN * H * W;
    float acc[OG][4];
#pragma unroll
    for (int o = 0; o < OG; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[o][q] = 0.f;

    for (int c0 = 0; c0 < CIN; c0 += DC_CC) {
        __syncthreads();
        if (TOK) {   // 18 x 18 positions x 4 float4 (16 channels): position-major reads, channel-major LDS image
            for (int e = tid; e < (DC_T + 2) * (DC_T + 2) * (DC_CC / 4); e += 256) {
                const int c4 = e & 3, p = e >> 2, pr = p / (DC_T + 2), pc = p - pr * (DC_T + 2);
                const int ii = i0 - 1 + pr, jj = j0 - 1 + pc;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ii >= 0 && ii < H && jj >= 0 && jj < W)
                    v = *reinterpret_cast<const float4 *>(inb + ((size_t)ii * W + jj) * CIN + c0 + 4 * c4);
                tile[4 * c4 + 0][pr][pc] = v.x; tile[4 * c4 + 1][pr][pc] = v.y;
                tile[4 * c4 + 2][pr][pc] = v.z; tile[4 * c4 + 3][pr][pc] = v.w;
            }
        } else {
            for (int e = tid; e < DC_CC * (DC_T + 2) * (DC_T + 2); e += 256) {
                const int pc = e % (DC_T + 2), rest = e / (DC_T + 2), pr = rest % (DC_T + 2), c = rest / (DC_T + 2);
                const int ii = i0 - 1 + pr, jj = j0 - 1 + pc;
                tile[c][pr][pc] = (ii >= 0 && ii < H && jj >= 0 && jj < W) ? inb[((size_t)(c0 + c) * H + ii) * W + jj] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int c = 0; c < DC_CC; ++c) {
            float n[3][3];                                    // n[a][d] = in(i - 1 + a, j - 1 + d)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int d = 0; d < 3; ++d) n[a][d] = tile[c][ti + a][tj + d];
            const float *wc = wf + ((size_t)(c0 + c) * COUT + og * OG) * 16;   // wave-uniform: scalar loads
#pragma unroll
            for (int o = 0; o < OG; ++o) {
                const float *k = wc + o * 16;                 // k[ky * 4 + kx]
                // (py, px) = (0,0): rows (ky1, i), (ky3, i-1); cols (kx1, j), (kx3, j-1)
                acc[o][0] = fmaf(n[1][1], k[5], fmaf(n[1][0], k[7], fmaf(n[0][1], k[13], fmaf(n[0][0], k[15], acc[o][0]))));
                // (0,1): cols (kx0, j+1), (kx2, j)
                acc[o][1] = fmaf(n[1][2], k[4], fmaf(n[1][1], k[6], fmaf(n[0][2], k[12], fmaf(n[0][1], k[14], acc[o][1]))));
                // (1,0): rows (ky0, i+1), (ky2, i)
                acc[o][2] = fmaf(n[2][1], k[1], fmaf(n[2][0], k[3], fmaf(n[1][1], k[9], fmaf(n[1][0], k[11], acc[o][2]))));
                // (1,1)
                acc[o][3] = fmaf(n[2][2], k[0], fmaf(n[2][1], k[2], fmaf(n[1][2], k[8], fmaf(n[1][1], k[10], acc[o][3]))));
            }
        }
    }
    const int i = i0 + ti, j = j0 + tj, OH = 2 * H, OW = 2 * W;
#pragma unroll
    for (int o = 0; o < OG; ++o) {
        const int oc = og * OG + o;
        const float t = shift[oc];
        float *op = out + (((size_t)b * COUT + oc) * OH + 2 * i) * OW + 2 * j;
        const float v0 = acc[o][0] + t, v1 = acc[o][1] + t, v2 = acc[o][2] + t, v3 = acc[o][3] + t;
        *reinterpret_cast<float2 *>(op) = make_float2(v0 > 0.f ? v0 : 0.f, v1 > 0.f ? v1 : 0.f);
        *reinterpret_cast<float2 *>(op + OW) = make_float2(v2 > 0.f ? v2 : 0.f, v3 > 0.f ? v3 : 0.f);
    }
}

// Small batches (a few frames: 16-64 workgroups of the kernel above, each a serial chain of 4 staging rounds x 16 channels whose weights
// arrive through scalar loads -- 22 us for 67 MFLOP at batch 1).  Same arithmetic, cut for latency: a workgroup owns an 8 x 8 tile of input
// positions and FOUR output channels (wave = channel, lane = position), stages ALL input channels of its 10 x 10 halo tile and its 4 x CIN x 16
// weights in LDS once -- one barrier -- and reads the weights back as broadcast 16-byte LDS reads.  Per channel the fmaf chain of k_convt4s2,
// channels in the same order: bit-identical outputs.
template <int CIN, int COUT, bool TOK>
__global__ __launch_bounds__(256) void k_convt4s2_small(const float *__restrict__ in, const float *__restrict__ wf, const float *__restrict__ shift,
                                                       float *__restrict__ out, int H, int W) {
    constexpr int T = 8, PW = T + 3;                          // 10 used columns + 1 pad
    __shared__ float tile[CIN][T + 2][PW];
    __shared__ __attribute__((aligned(16))) float wk[4][CIN][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, tj = lane & 7, ti = lane >> 3;
    const int tiles_x = W / T;
    const int i0 = (blockIdx.x / tiles_x) * T, j0 = (blockIdx.x % tiles_x) * T;
    const int og = blockIdx.y, b = blockIdx.z;
    const float *inb = in + (size_t)b * CIN * H * W;
    if (TOK) {   // 100 positions x CIN / 4 float4: position-major reads, channel-major LDS image
        for (int e = tid; e < (T + 2) * (T + 2) * (CIN / 4); e += 256) {
            const int c4 = e % (CIN / 4), p = e / (CIN / 4), pr = p / (T + 2), pc = p - pr * (T + 2);
            const int ii = i0 - 1 + pr, jj = j0 - 1 + pc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ii >= 0 && ii < H && jj >= 0 && jj < W) v = *reinterpret_cast<const float4 *>(inb + ((size_t)ii * W + jj) * CIN + 4 * c4);
            tile[4 * c4 + 0][pr][pc] = v.x; tile[4 * c4 + 1][pr][pc] = v.y;
            tile[4 * c4 + 2][pr][pc] = v.z; tile[4 * c4 + 3][pr][pc] = v.w;
        }
    } else {
        for (int e = tid; e < CIN * (T + 2) * (T + 2); e += 256) {
            const int pc = e % (T + 2), rest = e / (T + 2), pr = rest % (T + 2), c = rest / (T + 2);
            const int ii = i0 - 1 + pr, jj = j0 - 1 + pc;
            tile[c][pr][pc] = (ii >= 0 && ii < H && jj >= 0 && jj < W) ? inb[((size_t)c * H + ii) * W + jj] : 0.f;
        }
    }
    for (int e = tid; e < 4 * CIN * 4; e += 256) {            // weights [c][o][16] -> wk[o - 4 og][c][16], 16 bytes per item
        const int q = e & 3, c = (e >> 2) % CIN, o = e / (4 * CIN);
        *reinterpret_cast<float4 *>(&wk[o][c][4 * q]) = *reinterpret_cast<const float4 *>(wf + ((size_t)c * COUT + og * 4 + o) * 16 + 4 * q);
    }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
    for (int c = 0; c < CIN; ++c) {
        float n[3][3];                                        // n[a][d] = in(i - 1 + a, j - 1 + d)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int d = 0; d < 3; ++d) n[a][d] = tile[c][ti + a][tj + d];
        const float4 k0 = *reinterpret_cast<const float4 *>(&wk[wv][c][0]), k1 = *reinterpret_cast<const float4 *>(&wk[wv][c][4]);
        const float4 k2 = *reinterpret_cast<const float4 *>(&wk[wv][c][8]), k3 = *reinterpret_cast<const float4 *>(&wk[wv][c][12]);
        // k[ky * 4 + kx]: k0 = k[0..3], k1 = k[4..7], k2 = k[8..11], k3 = k[12..15]
        a0 = fmaf(n[1][1], k1.y, fmaf(n[1][0], k1.w, fmaf(n[0][1], k3.y, fmaf(n[0][0], k3.w, a0))));      // k[5], k[7], k[13], k[15]
        a1 = fmaf(n[1][2], k1.x, fmaf(n[1][1], k1.z, fmaf(n[0][2], k3.x, fmaf(n[0][1], k3.z, a1))));      // k[4], k[6], k[12], k[14]
        a2 = fmaf(n[2][1], k0.y, fmaf(n[2][0], k0.w, fmaf(n[1][1], k2.y, fmaf(n[1][0], k2.w, a2))));      // k[1], k[3], k[9], k[11]
        a3 = fmaf(n[2][2], k0.x, fmaf(n[2][1], k0.z, fmaf(n[1][2], k2.x, fmaf(n[1][1], k2.z, a3))));      // k[0], k[2], k[8], k[10]
    }
    const int i = i0 + ti, j = j0 + tj, OH = 2 * H, OW = 2 * W, oc = og * 4 + wv;
    const float t = shift[oc];
    float *op = out + (((size_t)b * COUT + oc) * OH + 2 * i) * OW + 2 * j;
    const float v0 = a0 + t, v1 = a1 + t, v2 = a2 + t, v3 = a3 + t;
    *reinterpret_cast<float2 *>(op) = make_float2(v0 > 0.f ? v0 : 0.f, v1 > 0.f ? v1 : 0.f);
    *reinterpret_cast<float2 *>(op + OW) = make_float2(v2 > 0.f ? v2 : 0.f, v3 > 0.f ? v3 : 0.f);
}

// Conv2d(16, 1, 3, padding 1) + Sigmoid: thread = one output pixel of a TH x TW tile (TH TW = 256).  8 x 32 tiles: the 32 lanes of one LDS
// access group read 32 consecutive floats of ONE row (conflict-free at any pitch); with 16 x 16 tiles a group spans two rows 19 floats apart
// and three banks collide (SQ_LDS_BANK_CONFLICT was half of the kernel's LDS cycles, profiles/r03/inference_b64_pmc_sq.json).
template <int TH, int TW>
__global__ __launch_bounds__(256) void k_conv3_sigmoid(const float *__restrict__ in, const float *__restrict__ w3, const float *__restrict__ b3,
                                                      float *__restrict__ out, int H, int W) {
    static_assert(TH * TW == 256, "one thread per output pixel");
    constexpr int PW = TW + 3;
    __shared__ float tile[16][TH + 2][PW];
    const int tid = threadIdx.x, tj = tid % TW, ti = tid / TW;
    const int tiles_x = W / TW;
    const int i0 = (blockIdx.x / tiles_x) * TH, j0 = (blockIdx.x % tiles_x) * TW, b = blockIdx.z;
    const float *inb = in + (size_t)b * 16 * H * W;
    for (int e = tid; e < 16 * (TH + 2) * (TW + 2); e += 256) {
        const int pc = e % (TW + 2), rest = e / (TW + 2), pr = rest % (TH + 2), c = rest / (TH + 2);
        const int ii = i0 - 1 + pr, jj = j0 - 1 + pc;
        tile[c][pr][pc] = (ii >= 0 && ii < H && jj >= 0 && jj < W) ? inb[((size_t)c * H + ii) * W + jj] : 0.f;
    }
    __syncthreads();
    float acc = b3[0];
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int d = 0; d < 3; ++d) acc = fmaf(tile[c][ti + a][tj + d], w3[c * 9 + a * 3 + d], acc);
    out[((size_t)b * H + i0 + ti) * W + j0 + tj] = 1.0f / (1.0f + expf(-acc));
}

hipError_t launch_decoder(const DecoderDev &d, const float *tokens, int B, int S, float *tmp1, float *tmp2, float *recon,
                          hipStream_t st) {
    const int t1 = (S / DC_T) * (S / DC_T), t2 = (2 * S / DC_T) * (2 * S / DC_T), t3 = (4 * S / DC_T) * (4 * S / DC_T);
    // a few frames: the latency-cut form (8 x 8 tiles, four output channels per workgroup, one staging round; SMK_DECODER_SMALL=0 keeps the
    // tiled form).  Measured interleaved on one box: batch 1 0.607 -> 0.589 ms per forward, batch 4 unchanged (1.130 -> 1.126)
    static const bool small_ok = !(getenv("SMK_DECODER_SMALL") && atoi(getenv("SMK_DECODER_SMALL")) == 0);
    if (small_ok && B <= 8 && S % 8 == 0) {
        const int s1 = (S / 8) * (S / 8), s2 = (2 * S / 8) * (2 * S / 8);
        hipLaunchKernelGGL((k_convt4s2_small<64, 32, true>), dim3(s1, 8, B), dim3(256), 0, st, tokens, d.w1, d.t1, tmp1, S, S);
        hipLaunchKernelGGL((k_convt4s2_small<32, 16, false>), dim3(s2, 4, B), dim3(256), 0, st, tmp1, d.w2, d.t2, tmp2, 2 * S, 2 * S);
    } else if ((long long)t1 * 4 * B < 256) {
        hipLaunchKernelGGL((k_convt4s2<64, 32, 2, true>), dim3(t1, 16, B), dim3(256), 0, st, tokens, d.w1, d.t1, tmp1, S, S);
        hipLaunchKernelGGL((k_convt4s2<32, 16, 2, false>), dim3(t2, 8, B), dim3(256), 0, st, tmp1, d.w2, d.t2, tmp2, 2 * S, 2 * S);
    } else {
        hipLaunchKernelGGL((k_convt4s2<64, 32, 8, true>), dim3(t1, 4, B), dim3(256), 0, st, tokens, d.w1, d.t1, tmp1, S, S);
        hipLaunchKernelGGL((k_convt4s2<32, 16, 8, false>), dim3(t2, 2, B), dim3(256), 0, st, tmp1, d.w2, d.t2, tmp2, 2 * S, 2 * S);
    }
    // (a no-LDS form of the last convolution for small batches -- every thread its 144 taps from L1 / L2 -- was built and measured 20 us
    //  SLOWER per batch-1 forward, interleaved on one box; removed)
    if ((4 * S) % 32 == 0) hipLaunchKernelGGL((k_conv3_sigmoid<8, 32>), dim3(t3, 1, B), dim3(256), 0, st, tmp2, d.w3, d.b3, recon, 4 * S, 4 * S);
    else hipLaunchKernelGGL((k_conv3_sigmoid<DC_T, DC_T>), dim3(t3, 1, B), dim3(256), 0, st, tmp2, d.w3, d.b3, recon, 4 * S, 4 * S);
    return hipGetLastError();
}

}  // namespace smk
