// Token-wise linear layers of SmokePhysNet's transformer body on the gfx950 matrix cores:
//     y = act(x W^T + b [+ periodic addend]) [+ residual]        x [M][K] fp32, W [N][K] fp32, y [M][N] fp32
// (feature_proj smokephys_net.py:38,97; q/k/v/out projections chaos_attention.py:25-28,77-79,113; FFN smokephys_net.py:153-158;
//  output_decoder smokephys_net.py:50-54).  The reference runs them as fp32 GEMMs; fp32 accuracy is kept with split-bf16
// operands (v = hi + lo, two bf16; product = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate), i.e.
// three bf16 MFMAs per fp32-equivalent MFMA at 16x the fp32 matrix rate.
//
// Structure (same K loop as k_encoder_bf16, encoder.hip): 256 threads = 4 waves; workgroup tile = 32*MB rows x 128 columns,
// wave w owns columns 32w..32w+31 for all MB row blocks.  A (activations): the fp32 rows are split on the fly and staged
// through LDS as two bf16 planes, 64 k per chunk, row pitch 144 B (conflict-free ds_read_b128 fragments), double buffered
// as ONE continuous chunk stream across the tiles a persistent workgroup walks (no prologue bubble per tile).  B (weights):
// pre-split once (launch_split_linear_weights) into fragment order and streamed L2 -> registers through a 4-deep ring of
// 1 KiB buffer loads; every workgroup keeps one column tile for its whole life, so the ring never restarts.
// One barrier per chunk, placed before the chunk's LAST k-step: the next chunk's first fragments are read under that
// k-step's MFMAs and the matrix pipe never waits on the barrier + LDS latency.
#include "linear.h"

namespace smk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int LN_PITCH = 144;      // bytes per staged row per plane: 64 bf16 + 16 pad (9 x 16 B, odd: 16 rows hit 16 bank groups)
constexpr int LN_RING = 4;         // B fragments in flight: 3 k-steps ahead; 4 k-steps per chunk keep the ring indices static
template <int MB> constexpr int ln_plane_bytes() { return MB * 32 * LN_PITCH; }
template <int MB> constexpr int ln_lds_bytes() { return 2 * 2 * ln_plane_bytes<MB>(); }   // MB = 4: 73,728 B -> 2 workgroups per CU

__global__ __launch_bounds__(256) void k_split_linear_weights(const float *__restrict__ w, const float *__restrict__ bias, LinearDev l) {
    const long long total = (long long)l.N * l.K;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = (int)(i / l.K), k = (int)(i - (long long)n * l.K);
        const float v = w[i];
        const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
        const size_t base = ((size_t)(k >> 4) * 2 * l.N + n) * 16 + (k & 15);
        l.wq[base] = __builtin_bit_cast(unsigned short, hi);
        l.wq[base + (size_t)l.N * 16] = __builtin_bit_cast(unsigned short, lo);
    }
    for (int n = blockIdx.x * 256 + threadIdx.x; n < l.N; n += gridDim.x * 256) l.bias[n] = bias ? bias[n] : 0.f;
}

hipError_t launch_split_linear_weights(const float *w, const float *bias, const LinearDev &l, hipStream_t st) {
    const long long total = (long long)l.N * l.K;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_split_linear_weights, dim3(blocks), dim3(256), 0, st, w, bias, l);
    return hipGetLastError();
}

__device__ __forceinline__ float gelu_erf(float v) {       // nn.GELU() default (smokephys_net.py:155)
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}

struct LinearArgs {
    LinearDev l;
    LinearCall c;
    int tiles_m, tiles_n;
};

template <int MB, int ACT>
__global__ __launch_bounds__(256, 2) void k_linear_x3(const LinearArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TM = MB * 32, PLANE = ln_plane_bytes<MB>();
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.l.K, N = a.l.N, M = a.c.M;
    const int nchunks = K >> 6, nks = K >> 4;
    const int tn = blockIdx.x % a.tiles_n;                   // fixed for the life of the workgroup (gridDim.x % tiles_n == 0)
    const int tm_step = gridDim.x / a.tiles_n;
    int tm = blockIdx.x / a.tiles_n;
    const int n = tn * 128 + wave * 32 + r;                  // this lane's output column
    const bool n_ok = n < N;

    // ---- B ring: 16 bytes per lane at a per-lane constant offset from a wave-uniform (scalar) fragment base
    const int lane_b = n_ok ? (n * 2 + hi) * 16 : 0;
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(a.l.wq), 0, K * N * 4, 0x00020000);
    const int frag_bytes = N * 32;                           // one (k-step, part) plane
    auto load_b = [&](int kn, int part) -> uint4 {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_b, (kn * 2 + part) * frag_bytes, 0);
        return make_uint4(v[0], v[1], v[2], v[3]);
    };
    uint4 bqh[LN_RING], bql[LN_RING];
#pragma unroll
    for (int k = 0; k < LN_RING - 1; ++k) {
        bqh[k] = load_b(k, 0);
        bql[k] = load_b(k, 1);
    }

    // ---- A staging: thread = float4 column sc of rows sr, sr+16, ... of the 32*MB x 64 chunk (a wave reads 4 full 256-B rows)
    const int sc = tid & 15, sr = tid >> 4;
    float4 stage[2 * MB];
    auto stage_load = [&](int tmx, int cx) {
#pragma unroll
        for (int j = 0; j < 2 * MB; ++j) {
            const long long row = (long long)tmx * TM + sr + 16 * j;
            stage[j] = row < M ? *reinterpret_cast<const float4 *>(a.c.x + row * a.c.ldx + cx * 64 + sc * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage_store = [&](int buf) {
        unsigned char *ph = smem + buf * 2 * PLANE + sr * LN_PITCH + sc * 8, *pl = ph + PLANE;
#pragma unroll
        for (int j = 0; j < 2 * MB; ++j) {
            const float v[4] = {stage[j].x, stage[j].y, stage[j].z, stage[j].w};
            bf16x4 vh, vl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const __bf16 h = (__bf16)v[i];
                vh[i] = h;
                vl[i] = (__bf16)(v[i] - (float)h);
            }
            *reinterpret_cast<bf16x4 *>(ph + 16 * j * LN_PITCH) = vh;
            *reinterpret_cast<bf16x4 *>(pl + 16 * j * LN_PITCH) = vl;
        }
    };
    // the chunk stream: (tile row, chunk) pairs in the order this workgroup consumes them; rows past M read as zeros
    int ld_tm = tm, ld_c = 0;
    auto advance = [&]() {
        if (++ld_c == nchunks) { ld_c = 0; ld_tm += tm_step; }
    };
    stage_load(ld_tm, ld_c); advance();
    stage_store(0);
    stage_load(ld_tm, ld_c); advance();
    __syncthreads();

    const int frag_off = r * LN_PITCH + hi * 16;
    auto load_a = [&](int buf, int ks, bf16x8 (&ah)[MB], bf16x8 (&al)[MB]) {
        const unsigned char *p = smem + buf * 2 * PLANE + frag_off + ks * 32;
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
            ah[mi] = *reinterpret_cast<const bf16x8 *>(p + mi * 32 * LN_PITCH);
            al[mi] = *reinterpret_cast<const bf16x8 *>(p + PLANE + mi * 32 * LN_PITCH);
        }
    };
    bf16x8 ahA[MB], alA[MB], ahB[MB], alB[MB];
    int buf = 0;
    load_a(0, 0, ahA, alA);
    const float bias = n_ok ? a.l.bias[n] : 0.f;

    for (; tm < a.tiles_m; tm += tm_step) {
        f32x16 acc[MB];
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[mi][g] = 0.f;

#pragma unroll 1
        for (int c = 0; c < nchunks; ++c) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                {   // refill the ring slot consumed one k-step ago (k index wraps: the next tile uses the same weights)
                    int kn = c * 4 + u + LN_RING - 1;
                    kn = kn >= nks ? kn - nks : kn;
                    kn = __builtin_amdgcn_readfirstlane(kn);
                    bqh[(u + LN_RING - 1) % LN_RING] = load_b(kn, 0);
                    bql[(u + LN_RING - 1) % LN_RING] = load_b(kn, 1);
                }
                if (u == 0) {   // the staged registers hold the chunk after this one: split + write it to the other buffer
                    stage_store(buf ^ 1);        // (last read before the previous chunk's barrier), then refill them
                    stage_load(ld_tm, ld_c);
                    advance();
                }
                if (u == 3) __syncthreads();     // other buffer complete and visible; every read of this buffer has returned
                if (u & 1) load_a(u == 3 ? buf ^ 1 : buf, (u + 1) & 3, ahA, alA);
                else load_a(buf, u + 1, ahB, alB);
                const bf16x8 bh = __builtin_bit_cast(bf16x8, bqh[u]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, bql[u]);
#pragma unroll
                for (int mi = 0; mi < MB; ++mi) {
                    if (u & 1) {
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alB[mi], bh, acc[mi], 0, 0, 0);
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahB[mi], bl, acc[mi], 0, 0, 0);
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahB[mi], bh, acc[mi], 0, 0, 0);
                    } else {
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alA[mi], bh, acc[mi], 0, 0, 0);
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahA[mi], bl, acc[mi], 0, 0, 0);
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahA[mi], bh, acc[mi], 0, 0, 0);
                    }
                }
            }
            buf ^= 1;
        }

        // ---- epilogue.  acc[mi][g]: row (g&3) + 8(g>>2) + 4hi of row block mi, column n
        if (n_ok) {
            const int row0 = tm * TM;
            const float *padd = nullptr;
            int ph0 = 0;
            float inv_period = 0.f;
            if (a.c.padd) {                                  // a tile lies inside one group (launcher: rows_per_group % TM == 0)
                const int grp = row0 / a.c.rows_per_group;
                ph0 = row0 - grp * a.c.rows_per_group;
                padd = a.c.padd + (size_t)grp * a.c.period * N + n;
                inv_period = 1.0f / (float)a.c.period;
            }
#pragma unroll
            for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int lr = mi * 32 + (g & 3) + 8 * (g >> 2) + 4 * hi, row = row0 + lr;
                    if (row < M) {
                        float v = acc[mi][g] + bias;
                        if (padd) {                          // (ph0 + lr) mod period without an integer division
                            const int xx = ph0 + lr;
                            int ph = xx - (int)((float)xx * inv_period) * a.c.period;
                            ph = ph < 0 ? ph + a.c.period : (ph >= a.c.period ? ph - a.c.period : ph);
                            v += padd[(size_t)ph * N];
                        }
                        if (ACT == 1) v = gelu_erf(v);
                        if (a.c.res) v = a.c.res[(long long)row * a.c.ldr + n] + v;
                        a.c.y[(long long)row * a.c.ldy + n] = v;
                    }
                }
        }
    }
}

template <int MB>
static hipError_t launch_mb(const LinearArgs &a, int nwg_max, hipStream_t st) {
    constexpr int lds = ln_lds_bytes<MB>();
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)k_linear_x3<MB, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void *)k_linear_x3<MB, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    const long long tiles = (long long)a.tiles_m * a.tiles_n;
    long long nwg = tiles < nwg_max ? tiles : nwg_max;
    nwg -= nwg % a.tiles_n;                               // every workgroup keeps one column tile
    if (nwg < a.tiles_n) nwg = a.tiles_n;
    if (a.c.act == 1) hipLaunchKernelGGL((k_linear_x3<MB, 1>), dim3((unsigned)nwg), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_linear_x3<MB, 0>), dim3((unsigned)nwg), dim3(256), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_linear_x3(const LinearDev &l, const LinearCall &c, hipStream_t st) {
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipGetLastError();
        num_cu = prop.multiProcessorCount;
    }
    LinearArgs a;
    a.l = l;
    a.c = c;
    a.tiles_n = cdiv(l.N, 128);
    // row-block count per tile: the largest that still gives every CU about two workgroups (small M: finer tiles)
    static int force_mb = -1;
    if (force_mb < 0) { const char *s = getenv("SMK_LINEAR_MB"); force_mb = s ? atoi(s) : 0; }
    int mb = 4;
    while (mb > 1 && (long long)cdiv(c.M, 32 * mb) * a.tiles_n < 2LL * num_cu) mb >>= 1;
    if (force_mb == 1 || force_mb == 2 || force_mb == 4) mb = force_mb;
    while (c.padd && mb > 1 && c.rows_per_group % (32 * mb) != 0) mb >>= 1;
    if (c.padd && c.rows_per_group % (32 * mb) != 0) return hipErrorInvalidValue;   // api.hip checks rows_per_group % 32 == 0
    a.tiles_m = cdiv(c.M, 32 * mb);
    const int nwg_max = 2 * num_cu;
    switch (mb) {
        case 4: return launch_mb<4>(a, nwg_max, st);
        case 2: return launch_mb<2>(a, nwg_max, st);
        default: return launch_mb<1>(a, nwg_max, st);
    }
}

}  // namespace smk
