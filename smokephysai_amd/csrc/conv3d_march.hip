// k_conv3d_march: configs[4]'s second convolution -- Conv3d(64 -> 128, 3 x 3 x 3, padding 1) + bias + ReLU on a channels-last volume
// [D][H][W][64] -- fused with the DEPTH half of the token pooling (SPEC_3D.md section 8: the depth axis is pooled to 1), as a march along z
// with the activations held in LDS.  The implicit-GEMM form on the layer kernel (linear.hip, CONV = 1) stages and splits every voxel's 64
// channels once per TAP (27 x); here a workgroup owns a column of 8 x 16 voxels, keeps the three planes z-1, z, z+1 of its 10 x 18 halo tile
// in LDS as swizzled bf16 hi / lo planes (k_encoder_b16's a1 image: encoder.hip), and reads all 27 taps of plane z's outputs from there:
// every voxel is staged and split 1.4 x (the halo) instead of 27 x, and the K loop is the 2-D encoder's tap loop three planes deep.
//   tile     8 x 16 voxels of one plane; 4 waves, wave w = output channels 32w .. 32w+31 (2 N tiles) x all 128 voxels (8 M tiles of one
//            16-voxel row): 64 accumulator registers, unit = 24 MFMAs (4 M tiles x 2 N tiles x 3 products) as in k_encoder_b16.
//   ring     slot of plane p = (p + 1) % 3; 3 x 2 x 23,040 B = 138,240 B of LDS -> one workgroup (one wave per SIMD) per CU.  Plane z+2 is
//            requested from HBM BEFORE output plane z's taps (48 registers per thread: one wave per SIMD has 512) and its six staging items per
//            thread are split and written into plane z-1's slot UNDER the MFMAs of the (kz = 1, 2) tap bodies, one item per body; the one
//            barrier per plane sits after the kz = 0 taps (the last readers of plane z-1; it also publishes the previous plane's stores).
//   loop     bodies of three taps (kx = 0, 1, 2 of one (kz, ky)): 12 units of 24 MFMAs, 6 32-k steps.  The next unit's 8 fragment reads are
//            issued in the first 16 MFMAs of a unit (one wave per SIMD: nothing else hides LDS latency), lo fragments first.
//   weights  the layer handle's pre-split layout [K/16][hi|lo][128][16] (K = 27 x 64, column tap * 64 + c), streamed from L2 through a ring
//            of six slots (= a body's six steps, so slot indices are compile-time), each fragment requested M3_AHEAD = 3 steps early.
//   output   zsum[y][x][o] = sum over z of relu(conv + bias): accumulators start at the bias, relu(acc) is added into 64 more registers in
//            z order (deterministic), written once per tile; smk_pool3d_accumulate (one "plane") then forms the 32 x 32 token sums.  The
//            activated conv2 output (8.6 GB per 512 x 512 x 64 volume) is never written.
//   measured 48.9 k cycles per plane and workgroup for 41.5 k of MFMA issue (-DSMK_M3_STAMPS); 13.2 ms per volume at 1.7-1.95 GHz: power-limited.
// Arithmetic: the split-bf16 x3 form (lo*hi + hi*lo + hi*hi, fp32 accumulation), taps in the order (kz, ky, kx) -- as the layer kernel.
#include "conv3d.h"
#include "linear.h"
#include <type_traits>

namespace smk {

typedef __bf16 m3_bf16x8 __attribute__((ext_vector_type(8)));
typedef float m3_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int m3_u32x4 __attribute__((ext_vector_type(4)));

constexpr int M3_TH = 8, M3_TW = 16, M3_AW = M3_TW + 2, M3_APIX = (M3_TH + 2) * M3_AW;      // halo tile 10 x 18 = 180 voxels
constexpr int M3_PART = M3_APIX * 128;                        // one bf16 plane of a slot: 64 channels x 2 B per voxel = 23,040 B
constexpr int M3_SLOT = 2 * M3_PART, M3_LDS = 3 * M3_SLOT;    // hi | lo; three z planes: 138,240 B
constexpr int M3_ITEMS = M3_APIX * 8, M3_NIT = (M3_ITEMS + 255) / 256;      // staging item = (voxel, group of 8 channels): 1,440 -> 6 per thread
constexpr unsigned M3_OOB = 0x80000000u;                      // an offset past every plane's range: the buffer load returns zeros

#ifdef SMK_M3_STAMPS      /* diagnostic build only (tools/README.md): s_memtime phase stamps of wave 0 into a debug buffer */
#define M3_STAMP(v) unsigned long long v; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0)
#define M3_RSTAMP(v) unsigned long long v; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0)
#define M3_STAMP_ARG , unsigned long long *stamps
#else
#define M3_STAMP(v)
#define M3_RSTAMP(v)
#define M3_STAMP_ARG
#endif

#ifndef M3_RD_EVERY
#define M3_RD_EVERY 2         /* the next unit's 8 fragment reads: one per this many MFMAs from the unit's start (one wave per SIMD: LDS latency is exposed) */
#endif
#ifndef M3_AHEAD
#define M3_AHEAD 3            /* 32-k steps between a weight fragment's request and its use (ring of 6 slots) */
#endif
__global__ __launch_bounds__(256, 1) void k_conv3d_march(const float *__restrict__ a1, int D, int H, int W, const unsigned short *__restrict__ wq,
                                                         const float *__restrict__ bias, float *__restrict__ zsum, int tiles_x, int ntiles,
                                                         int relu M3_STAMP_ARG) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int AH = M3_AHEAD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, kg = lane >> 4;
    const int o0 = wave * 32 + px;                                            // N tile 0; tile 1 = + 16
    const float b2a = bias[o0], b2b = bias[o0 + 16];

    // ---- weight ring (k_linear_b16's addressing with N = 128): 32-k step ks = tap * 2 + half, 54 per plane; four 16-byte loads per lane and
    //      step.  Six slots = the six k-steps of a body (three taps), so slot indices are compile-time; a fragment is requested AH steps early
    //      (one wave per SIMD: nobody else covers an L2 round trip, and the waits also cover the plane loads issued before them)
    const int lane_b = (kg >> 1) * (128 * 64) + o0 * 32 + (kg & 1) * 16;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(wq), 0, 27 * 64 * 128 * 4, 0x00020000);
    uint4 bq[6][2][2];                                                        // [slot][nt][hi | lo]
    auto load_b = [&](int slot, int kn) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const m3_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_b + nt * 512, kn * 16384 + part * 4096, 0);
                bq[slot][nt][part] = make_uint4(v[0], v[1], v[2], v[3]);
            }
    };
#pragma unroll
    for (int k = 0; k < AH; ++k) load_b(k, k);

    const unsigned plane_bytes = (unsigned)H * (unsigned)W * 256u;
    const int c16[2] = {kg << 4, (4 + kg) << 4};
#ifdef SMK_M3_STAMPS
    unsigned long long sm[6] = {0, 0, 0, 0, 0, 0};
    M3_STAMP(t_begin);
    M3_RSTAMP(r_begin);
#endif

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int r0 = ty * M3_TH, c0 = tx * M3_TW;

        // ---- this thread's staging items: voxel p of the halo tile, channel group g (32 bytes of the voxel's 256)
        unsigned goff[M3_NIT];
        int loff0[M3_NIT];
#pragma unroll
        for (int j = 0; j < M3_NIT; ++j) {
            int idx = tid + 256 * j;
            idx = idx < M3_ITEMS ? idx : M3_ITEMS - 1;
            const int g = idx & 7, p = idx >> 3, row = p / M3_AW, pc = p - row * M3_AW;
            const int ii = r0 - 1 + row, jj = c0 - 1 + pc;
            const bool in = ii >= 0 && ii < H && jj >= 0 && jj < W;
            goff[j] = in ? (unsigned)(ii * W + jj) * 256u + (unsigned)g * 32u : M3_OOB;
            loff0[j] = p * 128 + ((g ^ (p & 7)) << 4);
        }
        const bool last_item = tid + 256 * (M3_NIT - 1) < M3_ITEMS;           // the sixth item exists for 160 of the 256 threads
        float v[M3_NIT][8];
        auto stage_load = [&](int z) {                        // a plane outside the volume: every offset out of range -> zeros, no branch
            const bool inside = z >= 0 && z < D;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a1 + (size_t)(inside ? z : 0) * H * W * 64), 0,
                                                                                 (int)plane_bytes, 0x00020000);
#pragma unroll
            for (int j = 0; j < M3_NIT; ++j) {
                const unsigned off = inside ? goff[j] : M3_OOB;
                const m3_u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
                const m3_u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + 16u), 0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[j][c] = __uint_as_float(q0[c]);
                    v[j][4 + c] = __uint_as_float(q1[c]);
                }
            }
        };
        auto store_item = [&](unsigned char *base, const float (&x)[8], int off, bool ok) {
            m3_bf16x8 vh, vl;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const __bf16 h = (__bf16)x[c];
                vh[c] = h;
                vl[c] = (__bf16)(x[c] - (float)h);
            }
            if (ok) {
                *reinterpret_cast<m3_bf16x8 *>(base + off) = vh;
                *reinterpret_cast<m3_bf16x8 *>(base + M3_PART + off) = vl;
            }
        };
        // planes -1 (zeros), 0, 1 -> slots 0, 1, 2
#pragma unroll
        for (int pz = -1; pz < 2; ++pz) {
            stage_load(pz);
#pragma unroll
            for (int j = 0; j < M3_NIT; ++j) store_item(smem + (pz + 1) * M3_SLOT, v[j], loff0[j], j < M3_NIT - 1 || last_item);
        }
        __syncthreads();

        m3_f32x4 zs[8][2];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) zs[mt][nt] = m3_f32x4{0.f, 0.f, 0.f, 0.f};

        // A addressing (k_encoder_b16): halo voxel p = (mt + ki) * 18 + px + kj, unit = g ^ (p & 7), p & 7 = (px + kj + 2 (mt + ki)) & 7
        auto tap_consts = [&](int ki, int kj, int (&om)[4]) {
#pragma unroll
            for (int m = 0; m < 4; ++m) om[m] = ((px + kj) << 7) ^ (((px + kj + 2 * (m + ki)) & 7) << 4);
        };
        auto load_a = [&](int sbase, int ki, int half, int hm, const int (&om)[4], m3_bf16x8 (&ah)[4], m3_bf16x8 (&al)[4]) {
            const unsigned char *ph = smem + sbase + ki * (M3_AW * 128);                         // wave-uniform part
            // the lo fragments first: a unit's first eight MFMAs are the lo * hi products
#pragma unroll
            for (int m = 0; m < 4; ++m) al[m] = *reinterpret_cast<const m3_bf16x8 *>(ph + M3_PART + (om[m] ^ c16[half]) + (4 * hm + m) * (M3_AW * 128));
#pragma unroll
            for (int m = 0; m < 4; ++m) ah[m] = *reinterpret_cast<const m3_bf16x8 *>(ph + (om[m] ^ c16[half]) + (4 * hm + m) * (M3_AW * 128));
        };
        m3_bf16x8 ahA[4], alA[4], ahB[4], alB[4];
        int om[4];
        int s0 = 0;                                                           // slot of plane z - 1
        tap_consts(0, 0, om);
        load_a(0, 0, 0, 0, om, ahA, alA);                                     // (kz, ki, kj) = (0, 0, 0) of plane 0's outputs

#pragma unroll 1
        for (int z = 0; z < D; ++z) {
            M3_STAMP(t0);
            stage_load(z + 2);                                                // lands under the first taps; stored from (kz = 1, ki = 0) on
            M3_STAMP(t1);

            m3_f32x4 acc[8][2];                                               // starts at the bias
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float bb = nt ? b2b : b2a;
                    acc[mt][nt] = m3_f32x4{bb, bb, bb, bb};
                }
            unsigned char *const wr_base = smem + s0 * M3_SLOT;               // plane z + 2 replaces plane z - 1

            // one body = the three taps kj = 0, 1, 2 of (kz, ki): 12 units of 24 MFMAs, 6 k-steps.  STORE: one staging item of plane z + 2 is
            // split and written under the MFMAs of the body's fifth unit.
            auto body = [&](auto STORE_T, int it, int sbase, int ki, int sbase_n, int ki_n) {
                constexpr int SJ = decltype(STORE_T)::value;                  // staging item stored under this body, or -1
                constexpr bool STORE = SJ >= 0;
                auto unit = [&](auto PU) {
                    constexpr int pu = decltype(PU)::value, kj = pu >> 2, u = pu & 3, half = u >> 1, hm = u & 1, q = 2 * kj + half;
                    if (hm == 0) {                             // k-step ks + AH, at this k-step's start (wraps into the next plane's steps)
                        int kn = it * 6 + q + AH;
                        kn = kn >= 54 ? kn - 54 : kn;
                        load_b((q + AH) % 6, __builtin_amdgcn_readfirstlane(kn));
                    }
                    if (pu < 11) {
                        constexpr int nu = (pu + 1) & 3, nkj = (pu + 1) >> 2;
                        if (nu == 0) tap_consts(ki, nkj, om);
                        if (pu & 1) load_a(sbase, ki, nu >> 1, nu & 1, om, ahA, alA);
                        else load_a(sbase, ki, nu >> 1, nu & 1, om, ahB, alB);
                    } else {                                   // the next body's first unit (pu = 11 is odd: set A); past tap 26: the next plane's
                        tap_consts(ki_n, 0, om);
                        load_a(sbase_n, ki_n, 0, 0, om, ahA, alA);
                    }
                    if constexpr (SJ >= 0 && pu == 4) store_item(wr_base, v[SJ], loff0[SJ], SJ < M3_NIT - 1 || last_item);
                    // product-major: consecutive MFMAs go to different accumulators; each accumulator sums lo*hi, hi*lo, hi*hi in that order
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt) {
                                const m3_bf16x8 bh = __builtin_bit_cast(m3_bf16x8, bq[q][nt][0]);
                                const m3_bf16x8 bl = __builtin_bit_cast(m3_bf16x8, bq[q][nt][1]);
                                m3_f32x4 &c = acc[4 * hm + m][nt];
                                const m3_bf16x8 ah = (pu & 1) ? ahB[m] : ahA[m], al = (pu & 1) ? alB[m] : alA[m];
                                if (pr == 0) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
                                else if (pr == 1) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
                                else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
                            }
                    // 24 MFMAs: the next unit's 8 fragment reads early in the unit, the k-step's 4 weight loads between them; a
                    // staged item's split arithmetic (about 3 vector instructions per value) spread over the gaps, its two writes at the end
#pragma unroll
                    for (int i = 0; i < 24; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i % M3_RD_EVERY == 0 && i / M3_RD_EVERY < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        else if (hm == 0 && (i == 1 || i == 3 || i == 5 || i == 7)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        if (STORE && pu == 4) {
                            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                            if (i == 23) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                unit(std::integral_constant<int, 0>{}); unit(std::integral_constant<int, 1>{}); unit(std::integral_constant<int, 2>{});
                unit(std::integral_constant<int, 3>{}); unit(std::integral_constant<int, 4>{}); unit(std::integral_constant<int, 5>{});
                unit(std::integral_constant<int, 6>{}); unit(std::integral_constant<int, 7>{}); unit(std::integral_constant<int, 8>{});
                unit(std::integral_constant<int, 9>{}); unit(std::integral_constant<int, 10>{}); unit(std::integral_constant<int, 11>{});
            };
            const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;  // slots of planes z, z + 1
            // kz = 0: plane z - 1, still being read -- no stores
#pragma unroll 1
            for (int ki = 0; ki < 3; ++ki) {
                const int ki_n = ki == 2 ? 0 : ki + 1;
                const int sb_n = __builtin_amdgcn_readfirstlane((ki == 2 ? s1 : s0) * M3_SLOT);
                body(std::integral_constant<int, -1>{}, ki, s0 * M3_SLOT, ki, sb_n, ki_n);
            }
            M3_STAMP(t2);
            __syncthreads();                                   // every wave is done with plane z - 1; the stores of plane z + 1 (previous step) are visible
            M3_STAMP(t3);
            // kz = 1, 2: planes z, z + 1; the six items of plane z + 2 go into plane z - 1's slot, one per body (unrolled: the item is a
            // compile-time register set -- a rotating one made hipcc copy the loaded registers, i.e. wait for HBM, right behind the loads)
            const int b1 = s1 * M3_SLOT, b2 = s2 * M3_SLOT;
            body(std::integral_constant<int, 0>{}, 3, b1, 0, b1, 1);
            body(std::integral_constant<int, 1>{}, 4, b1, 1, b1, 2);
            body(std::integral_constant<int, 2>{}, 5, b1, 2, b2, 0);
            body(std::integral_constant<int, 3>{}, 6, b2, 0, b2, 1);
            body(std::integral_constant<int, 4>{}, 7, b2, 1, b2, 2);
            body(std::integral_constant<int, 5>{}, 8, b2, 2, b1, 0);           // after tap 26: plane z, the next output plane's first input
            M3_STAMP(t4);

            // ---- this plane's activations into the depth sums.  Lane: channel o0 (nt 0) / o0 + 16 (nt 1), voxels (row mt, columns 4 kg .. 4 kg + 3)
            // (the activation switch outside the loop: a per-element select on it serialises through vcc -- 2.1 k cycles instead of 0.5 k)
            if (relu) {
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) zs[mt][nt][i] += __builtin_fmaxf(acc[mt][nt][i], 0.f);
            } else {
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) zs[mt][nt][i] += acc[mt][nt][i];
            }
            s0 = s1;
#ifdef SMK_M3_STAMPS
            M3_STAMP(t5);
            sm[0] += t1 - t0; sm[1] += t2 - t1; sm[2] += t3 - t2; sm[3] += t4 - t3; sm[4] += t5 - t4;
#endif
        }

#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    zsum[((size_t)(r0 + mt) * W + c0 + 4 * kg + i) * 128 + o0 + 16 * nt] = zs[mt][nt][i];
        __syncthreads();                                       // the next tile's prologue rewrites every slot
    }
#ifdef SMK_M3_STAMPS
    M3_STAMP(t_end);
    M3_RSTAMP(r_end);
    if (stamps && tid == 0) {
        unsigned long long *rec = stamps + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 6; ++i) rec[i] = sm[i];
        rec[6] = t_end - t_begin; rec[7] = r_end - r_begin;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// k_conv3d_s7_march: configs[4]'s FIRST convolution -- Conv3d(1 -> 64, 7 x 7 x 7, padding 3) + bias + ReLU of a scalar volume [D][H][W] into
// the channels-last activations [D][H][W][64] -- as the same kind of march.  The implicit-GEMM form on the layer kernel (linear.hip, CONV = 2)
// gathers and splits every (voxel, window row) piece from global memory once per output voxel, uses two of its four waves (N = 64), and is
// bound by that staging (6.1 ms per 512 x 512 x 64 volume for 0.74 counted TFLOP).  Here:
//   K layout   8 ky slots x 8 kx slots per kz (7 x 7 used; slot 7 of either carries zero weights): K = 448, a 32-k step = the 4 window rows
//              ky = 4 (s & 1) .. + 3 of kz = s >> 1, 14 steps -- so the input plane of a step is wave-uniform (a scalar slot base).
//   weights    IN REGISTERS for the whole kernel: wave w owns output channels 16 w .. 16 w + 15, i.e. 14 steps x (hi | lo) fragments = 112
//              registers, loaded once from the layer handle's layout (N = 64, K = 448).
//   operand    an MFMA A fragment of output voxel (y, x), window row (kz, ky) is the 8 consecutive inputs x[z + kz - 3][y + ky - 3][x - 3 .. x + 4]:
//              a sliding window.  Each input plane of the 14 x 22 halo tile is expanded ONCE, when it enters the ring, into a table of
//              ready fragments [14 rows][16 positions] x (16 B hi | 16 B lo) = 7 KB; seven planes (z - 3 .. z + 3) = 50 KB of LDS, two
//              workgroups per CU.  A fragment fetch is then one aligned ds_read_b128 per part -- no gather, no split arithmetic in the loop.
//              Rows of a table are 256 B, so the four k groups of a read (four window rows) fall on disjoint 16-lane bank sets.
//   loop       per output plane and wave 14 steps x 8 tile rows x 3 products = 336 MFMAs on 224 fragment reads: the LDS pipe (each wave reads
//              every fragment: 896 KB per plane and workgroup = 7 k cycles at 128 B/clk) bounds it, not the matrix pipe (5.4 k).
// Arithmetic: split-bf16 x3, fp32 accumulation from the bias, window rows in (kz, ky) order.
constexpr int S7_ROWS = M3_TH + 6, S7_PART = S7_ROWS * 16 * 16, S7_PLANE = 2 * S7_PART, S7_LDS = 7 * S7_PLANE;      // 3,584 / 7,168 / 50,176 B
constexpr int S7_STEPS = 14, S7_PD = 4, S7_FENCE = 1000;                        // 32-k steps; fragment pairs in flight ahead of their MFMAs

__global__ __launch_bounds__(256, 2) void k_conv3d_s7_march(const float *__restrict__ x, int D, int H, int W, const unsigned short *__restrict__ wq,
                                                            const float *__restrict__ bias, float *__restrict__ a1, int tiles_x, int ntiles,
                                                            int relu) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, kg = lane >> 4;
    const int o = wave * 16 + px;
    const float bo = bias[o];

    // ---- this wave's weights: layer layout [K/16][hi|lo][64][16] (linear.h), 32-k step s = k16 blocks 2s, 2s+1; lane group kg = 8 k of the step
    m3_bf16x8 wh[S7_STEPS], wl[S7_STEPS];
    {
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(wq), 0, 448 * 64 * 4, 0x00020000);
        const int lane_b = (kg >> 1) * (64 * 64) + o * 32 + (kg & 1) * 16;
#pragma unroll
        for (int s = 0; s < S7_STEPS; ++s) {
            const m3_u32x4 h = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_b, s * 8192, 0);
            const m3_u32x4 l = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_b, s * 8192 + 2048, 0);
            wh[s] = __builtin_bit_cast(m3_bf16x8, h);
            wl[s] = __builtin_bit_cast(m3_bf16x8, l);
        }
    }
    // window row of (step s, group kg): kz = s >> 1, ky = 4 (s & 1) + kg (slot 7: zero weights -- any valid row)
    const int lane_a[2] = {kg * 256 + px * 16, (kg == 3 ? 6 : 4 + kg) * 256 + px * 16};
    const unsigned plane_bytes = (unsigned)H * (unsigned)W * 4u;
    const size_t a1_plane = (size_t)H * W * 64;

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int r0 = ty * M3_TH, c0 = tx * M3_TW;
        // ---- table builder: thread f < 224 = fragment (halo row f >> 4, position f & 15) = inputs (r0 - 3 + row, c0 - 3 + pos + 0 .. 7)
        unsigned goff0, gmask = 0;                                            // byte offset of element 0; bit e = element e lies inside the plane
        {
            const int row = (tid >> 4) < S7_ROWS ? (tid >> 4) : S7_ROWS - 1, pos = tid & 15;
            const int yy = r0 - 3 + row, xx0 = c0 - 3 + pos;
            goff0 = (unsigned)((yy * W + xx0) * 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) gmask |= (yy >= 0 && yy < H && xx0 + e >= 0 && xx0 + e < W) ? 1u << e : 0u;
        }
        float nx[8];
        auto plane_load = [&](int p) {
            if (p >= 0 && p < D) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (size_t)p * H * W), 0, (int)plane_bytes,
                                                                                     0x00020000);
#pragma unroll
                for (int e = 0; e < 8; ++e) nx[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((gmask >> e & 1u) ? goff0 + 4u * e : M3_OOB), 0, 0));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) nx[e] = 0.f;
            }
        };
        auto plane_store = [&](int slot) {
            m3_bf16x8 vh, vl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const __bf16 h = (__bf16)nx[e];
                vh[e] = h;
                vl[e] = (__bf16)(nx[e] - (float)h);
            }
            if (tid < S7_ROWS * 16) {
                *reinterpret_cast<m3_bf16x8 *>(smem + slot * S7_PLANE + tid * 16) = vh;
                *reinterpret_cast<m3_bf16x8 *>(smem + slot * S7_PLANE + S7_PART + tid * 16) = vl;
            }
        };
        // planes -3 .. 3 -> slots 0 .. 6 (slot of plane p = (p + 3) % 7)
#pragma unroll 1
        for (int p = -3; p <= 3; ++p) {
            plane_load(p);
            plane_store(p + 3);
        }
        __syncthreads();

        int zb = 0;                                                           // slot of plane z - 3
#pragma unroll 1
        for (int z = 0; z < D; ++z) {
            plane_load(z + 4);                                                // lands under the MFMAs
            int sbase[7];                                                     // scalar: LDS offset of input plane z - 3 + kz
#pragma unroll
            for (int kz = 0; kz < 7; ++kz) {
                const int sl = zb + kz;
                sbase[kz] = __builtin_amdgcn_readfirstlane((sl >= 7 ? sl - 7 : sl) * S7_PLANE);
            }
            m3_f32x4 acc[8];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) acc[mt] = m3_f32x4{bo, bo, bo, bo};
            // iteration n = (step n >> 3, tile row n & 7): fragment pair read S7_PD iterations ahead
            m3_bf16x8 fh[S7_PD], fl[S7_PD];
            auto frag = [&](int n, m3_bf16x8 &h, m3_bf16x8 &l) {
                const unsigned char *pa = smem + sbase[n >> 4] + lane_a[(n >> 3) & 1] + (n & 7) * 256;
                l = *reinterpret_cast<const m3_bf16x8 *>(pa + S7_PART);
                h = *reinterpret_cast<const m3_bf16x8 *>(pa);
            };
#pragma unroll
            for (int n = 0; n < S7_PD; ++n) frag(n, fh[n], fl[n]);
#pragma unroll
            for (int n = 0; n < S7_STEPS * 8; ++n) {
                const int s = n >> 3, mt = n & 7;
                const m3_bf16x8 h = fh[n % S7_PD], l = fl[n % S7_PD];
                if (n + S7_PD < S7_STEPS * 8) frag(n + S7_PD, fh[n % S7_PD], fl[n % S7_PD]);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l, wh[s], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h, wl[s], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h, wh[s], acc[mt], 0, 0, 0);
                // fenced every few iterations: left alone hipcc hoists a dozen reads to the top and spills lane constants, whose reloads then
                // wait behind the plane loads
                if (n % S7_FENCE == S7_FENCE - 1) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- a1[z][r0 + mt][c0 + 4 kg + i][o] = relu(acc): a scalar row pointer + one lane offset
            const int lane_o = 4 * kg * 64 + o;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                float *prow = a1 + (size_t)z * a1_plane + ((size_t)(r0 + mt) * W + c0) * 64;
#pragma unroll
                for (int i = 0; i < 4; ++i) prow[lane_o + i * 64] = relu ? __builtin_fmaxf(acc[mt][i], 0.f) : acc[mt][i];
            }
            __syncthreads();                                   // every wave is done with plane z - 3
            plane_store(zb);                                   // plane z + 4 takes its slot
            __syncthreads();
            zb = zb == 6 ? 0 : zb + 1;
        }
    }
}

hipError_t launch_conv3d_s7_march(const LinearDev &l, const float *x, int D, int H, int W, float *a1, int act, hipStream_t st) {
    if (l.K != 448 || l.N != 64 || D < 1 || H < M3_TH || W < M3_TW || H % M3_TH || W % M3_TW || (long long)H * W * 256 >= (1LL << 31) ||
        (act != 0 && act != 2))
        return hipErrorInvalidValue;
    once_per_device((const void *)k_conv3d_s7_march, [&] {
        (void)hipFuncSetAttribute((const void *)k_conv3d_s7_march, hipFuncAttributeMaxDynamicSharedMemorySize, S7_LDS);
    });
    const int tiles_x = W / M3_TW, ntiles = tiles_x * (H / M3_TH);
    const int nwg = 2 * device_num_cu();
    hipLaunchKernelGGL(k_conv3d_s7_march, dim3(ntiles < nwg ? ntiles : nwg), dim3(256), S7_LDS, st, x, D, H, W, l.wq, l.bias, a1, tiles_x, ntiles,
                       act == 2 ? 1 : 0);
    return hipGetLastError();
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
#ifdef SMK_M3_STAMPS
    static unsigned long long *stamps = nullptr;
    if (!stamps) (void)hipMalloc(&stamps, 1024 * 8 * sizeof(unsigned long long));
    const int nwg = ntiles < ncu ? ntiles : ncu;
    hipLaunchKernelGGL(k_conv3d_march, dim3(nwg), dim3(256), M3_LDS, st, a1, D, H, W, l.wq, l.bias, zsum, tiles_x, ntiles, act == 2 ? 1 : 0, stamps);
    if (getenv("SMK_M3_STAMPS_PRINT")) {
        static unsigned long long host[1024 * 8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(host, stamps, (size_t)nwg * 64, hipMemcpyDeviceToHost);
        double s[8] = {0};
        for (int w = 0; w < nwg; ++w)
            for (int i = 0; i < 8; ++i) s[i] += (double)host[w * 8 + i] / nwg;
        const double steps = (double)D * ((ntiles + nwg - 1) / nwg);
        fprintf(stderr, "m3 stamps (cycles per z step, wave 0): load-issue %.0f  taps kz0 %.0f  barrier %.0f  taps kz1,2 %.0f  epilogue %.0f  (-) %.0f | kernel %.0f cycles, %.3f ms, clock %.2f GHz\n",
                s[0] / steps, s[1] / steps, s[2] / steps, s[3] / steps, s[4] / steps, s[5] / steps, s[6], s[7] / 1e5, s[6] / (s[7] * 10.0) );
    }
#else
    hipLaunchKernelGGL(k_conv3d_march, dim3(ntiles < ncu ? ntiles : ncu), dim3(256), M3_LDS, st, a1, D, H, W, l.wq, l.bias, zsum, tiles_x, ntiles,
                       act == 2 ? 1 : 0);
#endif
    return hipGetLastError();
}

}  // namespace smk
