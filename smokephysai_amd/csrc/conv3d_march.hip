// k_conv3d_march: configs[4]'s second convolution -- Conv3d(64 -> 128, 3 x 3 x 3, padding 1) + bias + ReLU on a channels-last volume
// [D][H][W][64] -- fused with the DEPTH half of the token pooling (SPEC_3D.md section 8: the depth axis is pooled to 1), as a march along z
// with the activations held in LDS.  The implicit-GEMM form on the layer kernel (linear.hip, CONV = 1) stages and splits every voxel's 64
// channels once per TAP (27 x); here a workgroup owns a column of 8 x 16 voxels, keeps the three planes z-1, z, z+1 of its 10 x 18 halo tile
// in LDS as swizzled bf16 hi / lo planes (k_encoder_b16's a1 image: encoder.hip), and reads all 27 taps of plane z's outputs from there:
// every voxel is staged and split 1.4 x (the halo) instead of 27 x, and the K loop is the 2-D encoder's tap loop three planes deep.
//   tile     8 x 16 voxels of one plane; 4 waves, wave w = output channels 32w .. 32w+31 (2 N tiles) x all 128 voxels (8 M tiles of one
//            16-voxel row): 64 accumulator registers, unit = 24 MFMAs (4 M tiles x 2 N tiles x 3 products) as in k_encoder_b16.
//   ring     slot of plane p = (p + 1) % 3; 3 x 2 x 23,040 B = 138,240 B of LDS -> one workgroup (one wave per SIMD) per CU.  Plane z+2 is
//            requested from HBM BEFORE output plane z's tap loop (48 registers per thread: one wave per SIMD has 512) and split + stored into
//            the slot of plane z-1 behind it, between two barriers (about 3 % of a z step).
//   weights  the layer handle's pre-split layout [K/16][hi|lo][128][16] (K = 27 x 64, column tap * 64 + c), streamed from L2 through a ring
//            of two 32-k steps exactly as k_linear_b16 does.
//   output   zsum[y][x][o] = sum over z of relu(conv + bias): accumulated in 64 more registers in z order (deterministic), written once per
//            tile; smk_pool3d_accumulate (one "plane") then forms the 32 x 32 token sums.  The activated conv2 output (8.6 GB per 512 x 512 x 64
//            volume) is never written.
// Arithmetic: the split-bf16 x3 form (lo*hi + hi*lo + hi*hi, fp32 accumulation), taps in the order (kz, ky, kx) -- as the layer kernel.
#include "conv3d.h"
#include "linear.h"

namespace smk {

typedef __bf16 m3_bf16x8 __attribute__((ext_vector_type(8)));
typedef float m3_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int m3_u32x4 __attribute__((ext_vector_type(4)));

constexpr int M3_TH = 8, M3_TW = 16, M3_AW = M3_TW + 2, M3_APIX = (M3_TH + 2) * M3_AW;      // halo tile 10 x 18 = 180 voxels
constexpr int M3_PART = M3_APIX * 128;                        // one bf16 plane of a slot: 64 channels x 2 B per voxel = 23,040 B
constexpr int M3_SLOT = 2 * M3_PART, M3_LDS = 3 * M3_SLOT;    // hi | lo; three z planes: 138,240 B
constexpr int M3_ITEMS = M3_APIX * 8, M3_NIT = (M3_ITEMS + 255) / 256;      // staging item = (voxel, group of 8 channels): 1,440 -> 6 per thread
constexpr unsigned M3_OOB = 0x80000000u;                      // an offset past every plane's range: the buffer load returns zeros

__global__ __launch_bounds__(256, 1) void k_conv3d_march(const float *__restrict__ a1, int D, int H, int W, const unsigned short *__restrict__ wq,
                                                         const float *__restrict__ bias, float *__restrict__ zsum, int tiles_x, int ntiles,
                                                         int relu) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, kg = lane >> 4;
    const int o0 = wave * 32 + px;                                            // N tile 0; tile 1 = + 16
    const float b2a = bias[o0], b2b = bias[o0 + 16];

    // ---- weight ring (k_linear_b16's addressing with N = 128): 32-k step ks = tap * 2 + half; four 16-byte loads per lane and step
    const int lane_b = (kg >> 1) * (128 * 64) + o0 * 32 + (kg & 1) * 16;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(wq), 0, 27 * 64 * 128 * 4, 0x00020000);
    uint4 bq[2][2][2];                                                        // [slot][nt][hi | lo]
    auto load_b = [&](int slot, int kn) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const m3_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_b + nt * 512, kn * 16384 + part * 4096, 0);
                bq[slot][nt][part] = make_uint4(v[0], v[1], v[2], v[3]);
            }
    };
    load_b(0, 0);

    const unsigned plane_bytes = (unsigned)H * (unsigned)W * 256u;
    const int c16[2] = {kg << 4, (4 + kg) << 4};

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int r0 = ty * M3_TH, c0 = tx * M3_TW;

        // ---- this thread's staging items: voxel p of the halo tile, channel group g (32 bytes of the voxel's 256)
        unsigned goff[M3_NIT];
        int loff[M3_NIT];
#pragma unroll
        for (int j = 0; j < M3_NIT; ++j) {
            int idx = tid + 256 * j;
            idx = idx < M3_ITEMS ? idx : M3_ITEMS - 1;
            const int g = idx & 7, p = idx >> 3, row = p / M3_AW, pc = p - row * M3_AW;
            const int ii = r0 - 1 + row, jj = c0 - 1 + pc;
            const bool in = ii >= 0 && ii < H && jj >= 0 && jj < W;
            goff[j] = in ? (unsigned)(ii * W + jj) * 256u + (unsigned)g * 32u : M3_OOB;
            loff[j] = p * 128 + ((g ^ (p & 7)) << 4);
        }
        float v[M3_NIT][8];
        auto stage_load = [&](int z) {
            if (z >= 0 && z < D) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a1 + (size_t)z * H * W * 64), 0,
                                                                                     (int)plane_bytes, 0x00020000);
#pragma unroll
                for (int j = 0; j < M3_NIT; ++j) {
                    const m3_u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)goff[j], 0, 0);
                    const m3_u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(goff[j] + 16u), 0, 0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        v[j][c] = __uint_as_float(q0[c]);
                        v[j][4 + c] = __uint_as_float(q1[c]);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < M3_NIT; ++j)
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[j][c] = 0.f;
            }
        };
        auto stage_store = [&](int slot) {
            unsigned char *base = smem + slot * M3_SLOT;
#pragma unroll
            for (int j = 0; j < M3_NIT; ++j)
                if (tid + 256 * j < M3_ITEMS) {
                    m3_bf16x8 vh, vl;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const __bf16 h = (__bf16)v[j][c];
                        vh[c] = h;
                        vl[c] = (__bf16)(v[j][c] - (float)h);
                    }
                    *reinterpret_cast<m3_bf16x8 *>(base + loff[j]) = vh;
                    *reinterpret_cast<m3_bf16x8 *>(base + M3_PART + loff[j]) = vl;
                }
        };
        // planes -1 (zeros), 0, 1 -> slots 0, 1, 2
        stage_load(-1); stage_store(0);
        stage_load(0);  stage_store(1);
        stage_load(1);  stage_store(2);
        __syncthreads();

        m3_f32x4 zs[8][2];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) zs[mt][nt] = m3_f32x4{0.f, 0.f, 0.f, 0.f};

        int s0 = 0;                                                           // slot of plane z - 1
#pragma unroll 1
        for (int z = 0; z < D; ++z) {
            stage_load(z + 2);                                                // lands under the tap loop

            m3_f32x4 acc[8][2];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = m3_f32x4{0.f, 0.f, 0.f, 0.f};

            // A addressing (k_encoder_b16): halo voxel p = (mt + ki) * 18 + px + kj, unit = g ^ (p & 7), p & 7 = (px + kj + 2 (mt + ki)) & 7
            auto tap_consts = [&](int ki, int kj, int (&om)[4]) {
#pragma unroll
                for (int m = 0; m < 4; ++m) om[m] = ((px + kj) << 7) ^ (((px + kj + 2 * (m + ki)) & 7) << 4);
            };
            auto load_a = [&](int sbase, int ki, int half, int hm, const int (&om)[4], m3_bf16x8 (&ah)[4], m3_bf16x8 (&al)[4]) {
                const unsigned char *ph = smem + sbase + ki * (M3_AW * 128);                     // wave-uniform part
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int off = (om[m] ^ c16[half]) + (4 * hm + m) * (M3_AW * 128);
                    ah[m] = *reinterpret_cast<const m3_bf16x8 *>(ph + off);
                    al[m] = *reinterpret_cast<const m3_bf16x8 *>(ph + M3_PART + off);
                }
            };
            m3_bf16x8 ahA[4], alA[4], ahB[4], alB[4];
            int om[4];
            int kz = 0, ki = 0, kj = 0, sbase = s0 * M3_SLOT;
            tap_consts(0, 0, om);
            load_a(sbase, 0, 0, 0, om, ahA, alA);
#pragma unroll 1
            for (int tap = 0; tap < 27; ++tap) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int half = u >> 1, hm = u & 1, slot = half;
                    if (hm == 0) {                             // k-step ks + 1 into the other slot, at the k-step's start (wraps to the next plane's step 0)
                        int kn = tap * 2 + half + 1;
                        kn = kn >= 54 ? kn - 54 : kn;
                        load_b(slot ^ 1, __builtin_amdgcn_readfirstlane(kn));
                    }
                    if (u < 3) {
                        if (u & 1) load_a(sbase, ki, (u + 1) >> 1, (u + 1) & 1, om, ahA, alA);
                        else load_a(sbase, ki, (u + 1) >> 1, (u + 1) & 1, om, ahB, alB);
                    } else if (tap < 26) {                     // first unit of the next tap (u = 3 is odd: set A)
                        kj = kj == 2 ? 0 : kj + 1;
                        if (kj == 0) {
                            ki = ki == 2 ? 0 : ki + 1;
                            if (ki == 0) {
                                ++kz;
                                int s = s0 + kz;
                                s = s >= 3 ? s - 3 : s;
                                sbase = __builtin_amdgcn_readfirstlane(s * M3_SLOT);
                            }
                        }
                        tap_consts(ki, kj, om);
                        load_a(sbase, ki, 0, 0, om, ahA, alA);
                    }
                    // product-major: consecutive MFMAs go to different accumulators; each accumulator sums lo*hi, hi*lo, hi*hi in that order
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt) {
                                const m3_bf16x8 bh = __builtin_bit_cast(m3_bf16x8, bq[slot][nt][0]);
                                const m3_bf16x8 bl = __builtin_bit_cast(m3_bf16x8, bq[slot][nt][1]);
                                m3_f32x4 &c = acc[4 * hm + m][nt];
                                const m3_bf16x8 ah = (u & 1) ? ahB[m] : ahA[m], al = (u & 1) ? alB[m] : alA[m];
                                if (pr == 0) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
                                else if (pr == 1) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
                                else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
                            }
                    // 24 MFMAs: the next unit's 8 fragment reads one per third MFMA, the k-step's 4 weight loads behind its first MFMAs
#pragma unroll
                    for (int i = 0; i < 24; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        else if (hm == 0 && (i == 1 || i == 2 || i == 4 || i == 5)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }

            // ---- this plane's activations into the depth sums.  Lane: channel o0 (nt 0) / o0 + 16 (nt 1), voxels (row mt, columns 4 kg .. 4 kg + 3)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float y = acc[mt][nt][i] + (nt ? b2b : b2a);
                        y = (relu && !(y > 0.f)) ? 0.f : y;
                        zs[mt][nt][i] += y;
                    }

            __syncthreads();                                   // every wave is done with plane z - 1
            stage_store(s0);                                   // plane z + 2 (zeros past the volume) takes its slot
            __syncthreads();
            s0 = s0 == 2 ? 0 : s0 + 1;
        }

#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    zsum[((size_t)(r0 + mt) * W + c0 + 4 * kg + i) * 128 + o0 + 16 * nt] = zs[mt][nt][i];
    }
}

hipError_t launch_conv3d_march(const LinearDev &l, const float *a1, int D, int H, int W, float *zsum, int act, hipStream_t st) {
    if (l.K != 27 * 64 || l.N != 128 || D < 1 || H < M3_TH || W < M3_TW || H % M3_TH || W % M3_TW || (long long)H * W * 256 >= (1LL << 31) ||
        (act != 0 && act != 2))
        return hipErrorInvalidValue;
    once_per_device((const void *)k_conv3d_march, [&] {
        (void)hipFuncSetAttribute((const void *)k_conv3d_march, hipFuncAttributeMaxDynamicSharedMemorySize, M3_LDS);
    });
    const int tiles_x = W / M3_TW, ntiles = tiles_x * (H / M3_TH);
    const int ncu = device_num_cu();
    hipLaunchKernelGGL(k_conv3d_march, dim3(ntiles < ncu ? ntiles : ncu), dim3(256), M3_LDS, st, a1, D, H, W, l.wq, l.bias, zsum, tiles_x, ntiles,
                       act == 2 ? 1 : 0);
    return hipGetLastError();
}

}  // namespace smk
