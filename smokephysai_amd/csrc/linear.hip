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
#include <type_traits>

#include "linear.h"

namespace smk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int LN_PITCH = 144;      // bytes per staged row per plane: 64 bf16 + 16 pad (9 x 16 B, odd: 16 rows hit 16 bank groups)
#ifndef SMK_LINEAR_SCHED
#define SMK_LINEAR_SCHED 1
#endif
#ifndef SMK_LN_RINGFIRST
#define SMK_LN_RINGFIRST 1
#endif
constexpr int LN_RING = 4;         // B fragments in flight: 3 k-steps ahead; 4 k-steps per chunk keep the ring indices static
template <int MB> constexpr int ln_plane_bytes() { return MB * 32 * LN_PITCH; }
template <int MB, int NW> constexpr int ln_lds_bytes() { return 2 * 2 * ln_plane_bytes<MB>() + 2048; }   // + bias tile (+ fused-LayerNorm column sums, row statistics); MB = 4: 75,264 B -> 2 workgroups per CU

// transposed: `w` is [K][N] row-major (the handle then computes x W for a layer whose weight is W [K][N]: its input-gradient GEMM).
// ld: source row pitch in floats; k_valid: k >= k_valid reads as zero (the weight-gradient form pads the token rows to whole segments).
__global__ __launch_bounds__(256) void k_split_linear_weights(const float *__restrict__ w, const float *__restrict__ bias, LinearDev l,
                                                              int transposed, long long ld, int k_valid) {
    const long long total = (long long)l.N * l.K;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int n, k;
        if (transposed) { k = (int)(i / l.N); n = (int)(i - (long long)k * l.N); }     // i walks the source in memory order either way
        else { n = (int)(i / l.K); k = (int)(i - (long long)n * l.K); }
        const float v = k < k_valid ? (transposed ? w[(long long)k * ld + n] : w[(long long)n * ld + k]) : 0.f;
        const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
        const size_t base = ((size_t)(k >> 4) * 2 * l.N + n) * 16 + (k & 15);
        l.wq[base] = __builtin_bit_cast(unsigned short, hi);
        l.wq[base + (size_t)l.N * 16] = __builtin_bit_cast(unsigned short, lo);
    }
    if (l.bias)
        for (int n = blockIdx.x * 256 + threadIdx.x; n < l.N; n += gridDim.x * 256) l.bias[n] = bias ? bias[n] : 0.f;
}

// The transposed form at weight-gradient sizes (the "weights" are the 65,536 x in_features activations): thread = (16-k block, n);
// 16 loads that are coalesced across the wave (consecutive n) and two 32-byte stores that tile the fragment planes contiguously.
__global__ __launch_bounds__(256) void k_split_linear_weights_t16(const float *__restrict__ w, LinearDev l, long long ld, int k_valid) {
    const int n = blockIdx.y * 256 + threadIdx.x;
    if (n >= l.N) return;
    for (int kb = blockIdx.x; kb < l.K / 16; kb += gridDim.x) {
        unsigned short hi[16], lo[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = kb * 16 + j;
            const float v = k < k_valid ? w[(long long)k * ld + n] : 0.f;
            const __bf16 h = (__bf16)v;
            hi[j] = __builtin_bit_cast(unsigned short, h);
            lo[j] = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)h));
        }
        uint4 *dh = reinterpret_cast<uint4 *>(l.wq + ((size_t)kb * 2 * l.N + n) * 16);
        uint4 *dl = reinterpret_cast<uint4 *>(l.wq + ((size_t)kb * 2 * l.N + l.N + n) * 16);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            dh[q] = make_uint4(hi[8 * q] | (unsigned)hi[8 * q + 1] << 16, hi[8 * q + 2] | (unsigned)hi[8 * q + 3] << 16,
                               hi[8 * q + 4] | (unsigned)hi[8 * q + 5] << 16, hi[8 * q + 6] | (unsigned)hi[8 * q + 7] << 16);
            dl[q] = make_uint4(lo[8 * q] | (unsigned)lo[8 * q + 1] << 16, lo[8 * q + 2] | (unsigned)lo[8 * q + 3] << 16,
                               lo[8 * q + 4] | (unsigned)lo[8 * q + 5] << 16, lo[8 * q + 6] | (unsigned)lo[8 * q + 7] << 16);
        }
    }
}

hipError_t launch_split_linear_weights(const float *w, const float *bias, const LinearDev &l, hipStream_t st, int transposed,
                                       long long ld, int k_valid) {
    if (ld <= 0) ld = transposed ? l.N : l.K;
    if (k_valid < 0) k_valid = l.K;
    if (transposed && !l.bias && l.K % 16 == 0 && (long long)l.N * l.K >= (1 << 20)) {
        const int kblocks = l.K / 16;
        hipLaunchKernelGGL(k_split_linear_weights_t16, dim3(kblocks < 4096 ? kblocks : 4096, cdiv(l.N, 256)), dim3(256), 0, st, w, l, ld, k_valid);
        return hipGetLastError();
    }
    const long long total = (long long)l.N * l.K;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_split_linear_weights, dim3(blocks), dim3(256), 0, st, w, bias, l, transposed, ld, k_valid);
    return hipGetLastError();
}

// nn.GELU() default (smokephys_net.py:155): 0.5 v (1 + erf(v / sqrt 2)).  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7,
// i.e. ~1e-7 of |v| on the result -- far inside the 1e-4 parity bar) in ~15 VALU instructions; the library erff costs
// about as much as the tile's MFMAs in the 512 -> 2048 layer.
__device__ __forceinline__ float gelu_erf(float v) {
    const float z = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);
    const float erf_abs = fmaf(-(p * t), e, 1.0f);
    return 0.5f * v * (1.0f + copysignf(erf_abs, v));
}

#ifdef SMK_LN_STAMPS      /* diagnostic build only (tools/README.md): s_memtime phase stamps into a debug buffer */
#define LN_STAMP(v) unsigned long long v; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0)
#define LN_RSTAMP(v) unsigned long long v; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0)
#endif
// timing ablations (SMK_LINEAR_DBG) exist only in diagnostic builds; the product library ignores the variable
#ifdef SMK_LN_DIAG
#define LN_DBG(a, bit) ((a).dbg & (bit))
#else
#define LN_DBG(a, bit) 0
#endif
// 4 fp32 values as the 16 bytes {hi[0..3], lo[0..3]} (bf16 pairs: hi = RNE(v), lo = RNE(v - hi)): LinearCall::split_from
__device__ __forceinline__ float4 split4_inplace(const float (&v)[4]) {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t vh, vl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 h = (__bf16)v[i];
        vh[i] = h;
        vl[i] = (__bf16)(v[i] - (float)h);
    }
    const float2 h2 = __builtin_bit_cast(float2, vh), l2 = __builtin_bit_cast(float2, vl);
    return make_float4(h2.x, h2.y, l2.x, l2.y);
}

struct LinearArgs {
    LinearDev l;
    LinearCall c;
    int tiles_m, tiles_n;
    int swz, stagger, stagger_unit, num_cu;
    // implicit-GEMM Conv3d (k_linear_b16<NW, true>): x is a channels-last slab [cDl][cH][cW][64], output rows are the voxels of planes
    // cz_off .. of that slab in memory order; chunk c of the K range is tap c of the 3 x 3 x 3 window (zero outside the slab)
    int cDl, cH, cW, cz_off;
    unsigned long long *stamps;
    int dbg;              // timing ablations, honoured only by -DSMK_LN_DIAG builds (results are wrong when non-zero): 1 A loads re-read tile 0,
                          // 2 B ring re-reads k-step 0, 4 no epilogue, 8 epilogue stores as whole 128-byte row pieces
};

// AS: the activations arrive already split (SMK_FMT_SPLIT_BF16: per row, per 8 k: 8 hi | 8 lo bf16 -- the same 4 bytes per
// element as fp32, written by the producing kernel's epilogue): staging is then a 16-byte copy, no arithmetic in the K loop.
// KS > 1 (small problems only: fewer tiles than the chip has workgroup slots): KS wave groups of NW waves share one output tile, each
// walking 1/KS of the K range with its own LDS chunk stream; the partial sums are merged through LDS in fixed group order
// (deterministic, no atomics) and group 0 runs the epilogue.
// RING: weight fragments in flight, in 16-k steps (default LN_RING = 4: three steps ahead -- enough when other waves cover an L2 round
// trip).  The one-tile-per-workgroup problems of a single frame (M = 1,024, MB = 1: a k-step is 3 MFMAs = 96 cycles) are bound by exactly
// that round trip -- 32 k-steps x ~0.3 us = the 10 us such a layer took -- so they run RING = 16 (fifteen steps = ~1 us ahead; the chunk loop
// is unrolled RING / 4 times so that the slot indices stay compile-time; needs K % (16 RING) == 0).
// LNF: LayerNorm fused in front (LinearCall::ln_wsum; fp32 input, one wave group, ONE tile per workgroup: the row sums are gathered over
// the workgroup's whole chunk stream).
template <int MB, int NW, bool AS, int KS = 1, int RING = LN_RING, bool LNF = false>
__global__ __launch_bounds__(NW * 64 * KS, KS > 2 ? 1 : 2) void k_linear_x3(const LinearArgs a) {
    static_assert(!LNF || (!AS && KS == 1), "fused LayerNorm: fp32 activations, one wave group");
    constexpr int TN = NW * 32, RP = AS ? NW * 8 : NW * 4;   // tile columns; rows staged per pass (fp32: 16 float4 per row chunk; split: 8 x 32 B)
    constexpr bool sched = SMK_LINEAR_SCHED;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    constexpr int TM = MB * 32, PLANE = ln_plane_bytes<MB>();
    const int grp = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / (NW * 64));
    unsigned char *smem = smem_all + grp * ln_lds_bytes<MB, NW>();
    const int tid = KS == 1 ? (int)threadIdx.x : (int)threadIdx.x % (NW * 64), lane = tid & 63, r = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.l.K, N = a.l.N, M = a.c.M;
    const int nchunks = (K >> 6) / KS, nks = nchunks * 4;      // this wave group's share of the K range
    const int c_off = grp * nchunks, k_off = c_off * 4;
    // Workgroups are dealt to the 8 XCDs round-robin (id % 8).  vid renumbers them so that one XCD holds a contiguous id range:
    // the tiles_n workgroups that share a row block (the same A rows) then run on ONE XCD at the same time and A is fetched
    // from HBM once (L2 hits for the others) instead of once per XCD.  (gridDim.x % (8 * tiles_n) == 0 or swz == 0.)
    const int vid0 = a.swz ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    // K-segmented launches (LinearCall::nseg > 1): a contiguous id range per segment; the segment only offsets the three base pointers
    const int wg_per_seg = (int)gridDim.x / a.c.nseg;
    const int seg = a.c.nseg > 1 ? vid0 / wg_per_seg : 0;
    const int vid = vid0 - seg * wg_per_seg;
    const int tn = vid % a.tiles_n;                          // fixed for the life of the workgroup (wg_per_seg % tiles_n == 0)
    const int tm_step = wg_per_seg / a.tiles_n;
    int tm = vid / a.tiles_n;
    // Two workgroups share a CU and run the same program with the same period: delay every other dispatch round by about
    // half a tile so that one's epilogue / staging stalls overlap the other's MFMA stretch (speed only).
    // Tiles of equal length keep all workgroups of the chip in lock-step: every epilogue is then one chip-wide write burst
    // that drains at HBM write bandwidth while the matrix pipes idle.  Starting the workgroups in `stagger` phase groups
    // spreads the bursts (a one-time cost of up to (stagger-1)/stagger of a tile for the last group).
    if (a.stagger > 1) {
        const int ph = (blockIdx.x >> 3) % a.stagger;          // same XCD, consecutive CUs -> different phases
        for (int i = 0; i < ph * a.stagger_unit; ++i) __builtin_amdgcn_s_sleep(127);
    }
    float *const y_seg = a.c.y + (size_t)seg * M * a.c.ldy;   // segment s writes its own dense [M][N] slab
    const int n = tn * TN + wave * 32 + r;                  // this lane's output column
    const bool n_ok = n < N;

    // ---- B ring: 16 bytes per lane at a per-lane constant offset from a wave-uniform (scalar) fragment base
    const int lane_b = n_ok ? (n * 2 + hi) * 16 : 0;
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(a.l.wq) + (size_t)seg * K * N * 2, 0, K * N * 4, 0x00020000);
    const int frag_bytes = N * 32;                           // one (k-step, part) plane
    auto load_b = [&](int kn, int part) -> uint4 {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_b, ((LN_DBG(a, 2) ? 0 : kn + k_off) * 2 + part) * frag_bytes, 0);
        return make_uint4(v[0], v[1], v[2], v[3]);
    };
    uint4 bqh[RING], bql[RING];
#pragma unroll
    for (int k = 0; k < RING - 1; ++k) {
        bqh[k] = load_b(k, 0);
        bql[k] = load_b(k, 1);
    }

    // ---- A staging.  fp32: thread = float4 column sc of rows sr, sr+RP, ... of the 32*MB x 64 chunk (a wave reads 4 full 256-B
    //      rows); split: thread = k-group sc (8 hi | 8 lo = 32 B) of rows sr, sr+RP, ...
    const int sc = AS ? (tid & 7) : (tid & 15), sr = AS ? (tid >> 3) : (tid >> 4);
    constexpr int NPC_ALL = TM / RP;                         // pieces per chunk per thread
    float4 stage[AS ? 1 : NPC_ALL];
    u32x4 sth[AS ? NPC_ALL : 1], stl[AS ? NPC_ALL : 1];
    // LNF: every staged row piece is taken relative to a PIVOT -- the row's own first element x[row][0], loaded beside the piece -- before
    // it is split into bf16: the GEMM then sees operands of the size of the row's SPREAD, not of its mean, and
    //   LN(x) W'^T = rstd ((x - p) W'^T - (mean - p) wsum),   mean - p = S1 / K,   var = S2 / K - (S1 / K)^2
    // with S1 = sum (x - p), S2 = sum (x - p)^2 loses nothing to cancellation for rows whose mean dwarfs their spread (neither in the
    // variance nor in the epilogue's subtraction).  ln_s / ln_q: this thread's share (4 of every 64 k) of S1 / S2 of rows sr + RP j.
    // The chunk stream runs one chunk ahead of the K loop, so the first chunk of the NEXT tile is staged while this tile still
    // computes: its sums start in the ln_n* set and become the current set after the epilogue.
    float ln_piv[LNF ? NPC_ALL : 1], ln_s[LNF ? NPC_ALL : 1], ln_q[LNF ? NPC_ALL : 1], ln_ns[LNF ? NPC_ALL : 1], ln_nq[LNF ? NPC_ALL : 1];
#pragma unroll
    for (int j = 0; j < (LNF ? NPC_ALL : 1); ++j) { ln_piv[j] = ln_s[j] = ln_q[j] = ln_ns[j] = ln_nq[j] = 0.f; }
    // x through a buffer resource: a row past M (ragged last tile, or the chunk stream running past this workgroup's last
    // tile) is out of range and reads as zero in hardware -- no clamp, no predicate (either would cost VALU issue slots or
    // split the k-step into basic blocks and undo the MFMA / staging interleave below).  One v_add per load: the offset
    // must sit in the VGPR operand to be range-checked.
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.c.x) + (size_t)seg * K, 0, (int)(((long long)(M - 1) * a.c.ldx + K) * 4), 0x00020000);
    const int ldxb = (int)a.c.ldx * 4;                       // row bytes (split rows are dense: ldx = K)
    const int lane_x = sr * ldxb + sc * (AS ? 32 : 16);
    auto stage_load = [&](int tmx, int cx, int j) {
        const unsigned row_u = (unsigned)(LN_DBG(a, 1) ? 0 : tmx) * TM + RP * j;          // wave-uniform part (SALU); tmx <= tiles_m
        const unsigned off = row_u * (unsigned)ldxb + (unsigned)(cx + c_off) * 256u;      // < 2^32: api.hip bounds (rows + 256) * ldx
        if (AS) {
            sth[AS ? j : 0] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(off + (unsigned)lane_x), 0, 0);
            stl[AS ? j : 0] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(off + (unsigned)lane_x + 16u), 0, 0);
        } else {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(off + (unsigned)lane_x), 0, 0);
            // (not __builtin_bit_cast(float, v[i]): hipcc 7.2 then emits a 1-dword load and leaves v[1..3] undefined)
            stage[AS ? 0 : j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            if constexpr (LNF)       // the row's pivot x[row][0] (the 16 threads of a staging row read one address; a row past M reads 0)
                ln_piv[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)(row_u * (unsigned)ldxb + (unsigned)(sr * ldxb)), 0, 0));
        }
    };
    // first_of_next (wave-uniform, LNF only): the piece being stored is chunk 0 of the tile AFTER the one in the K loop
    auto stage_store = [&](int buf, int j, bool first_of_next = false) {
        if (AS) {
            unsigned char *ph = smem + buf * 2 * PLANE + (sr + RP * j) * LN_PITCH + sc * 16;
            *reinterpret_cast<u32x4 *>(ph) = sth[AS ? j : 0];
            *reinterpret_cast<u32x4 *>(ph + PLANE) = stl[AS ? j : 0];
            return;
        }
        unsigned char *ph = smem + buf * 2 * PLANE + (sr + RP * j) * LN_PITCH + sc * 8;
        float v[4] = {stage[AS ? 0 : j].x, stage[AS ? 0 : j].y, stage[AS ? 0 : j].z, stage[AS ? 0 : j].w};
        if constexpr (LNF) {
            const float pv = ln_piv[j];                                      // loaded with this very piece
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = v[i] - pv;
            const float s1 = (v[0] + v[1]) + (v[2] + v[3]), s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            ln_s[j] += first_of_next ? 0.f : s1;                             // selects, not branches: the k-step stays one basic block
            ln_q[j] += first_of_next ? 0.f : s2;
            ln_ns[j] = first_of_next ? s1 : ln_ns[j];
            ln_nq[j] = first_of_next ? s2 : ln_nq[j];
        }
        bf16x4 vh, vl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#ifdef SMK_LN_NOCONV      /* timing-only build: no split arithmetic (wrong results) */
            vh[i] = __builtin_bit_cast(__bf16, (unsigned short)(__builtin_bit_cast(unsigned, v[i]) >> 16));
            vl[i] = vh[i];
#else
            const __bf16 h = (__bf16)v[i];
            vh[i] = h;
            vl[i] = (__bf16)(v[i] - (float)h);
#endif
        }
        *reinterpret_cast<bf16x4 *>(ph) = vh;
        *reinterpret_cast<bf16x4 *>(ph + PLANE) = vl;
    };
    // the chunk stream: (tile row, chunk) pairs in the order this workgroup consumes them; rows past M read as zeros
    int ld_tm = tm, ld_c = 0;
    auto advance = [&]() {
        if (++ld_c == nchunks) { ld_c = 0; ld_tm = ld_tm + tm_step < a.tiles_m ? ld_tm + tm_step : a.tiles_m; }   // past the end: row >= M
    };
#pragma unroll
    for (int j = 0; j < TM / RP; ++j) stage_load(ld_tm, ld_c, j);
    advance();
#pragma unroll
    for (int j = 0; j < TM / RP; ++j) {
        stage_store(0, j, true);
        stage_load(ld_tm, ld_c, j);
    }
    if constexpr (LNF) {                                             // chunk 0 of the FIRST tile: its sums are the current set
#pragma unroll
        for (int j = 0; j < NPC_ALL; ++j) { ln_s[j] = ln_ns[j]; ln_q[j] = ln_nq[j]; }
    }
    advance();
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
    const bool nw_ok = tn * TN + wave * 32 < N;
    float *bias_s = reinterpret_cast<float *>(smem + 4 * PLANE);       // this workgroup's 128 bias values
    if (tid < TN) bias_s[tid] = (a.l.bias && tn * TN + tid < N) ? a.l.bias[tn * TN + tid] : 0.f;   // visible after the first barrier below             // N % 32 == 0: a wave's 32 columns are all inside or all outside
    float *wsum_s = bias_s + TN, *stat_s = wsum_s + TN;                  // LNF: column sums of W', then (mean, rstd) per tile row
    if (LNF && tid < TN) wsum_s[tid] = tn * TN + tid < N ? a.c.ln_wsum[tn * TN + tid] : 0.f;

#ifdef SMK_LN_STAMPS
    unsigned long long sum_k = 0, sum_e = 0, ntl = 0, sum_u[5] = {0, 0, 0, 0, 0}, t_prev = 0;
    LN_STAMP(t_begin);
    LN_RSTAMP(r_begin);
#endif
    for (; tm < a.tiles_m; tm += tm_step) {
#ifdef SMK_LN_STAMPS
        LN_STAMP(t_k0);
#endif
        f32x16 acc[MB];
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[mi][g] = 0.f;

#pragma unroll 1
        for (int c0 = 0; c0 < nchunks; c0 += RING / 4) {
            auto kstep = [&](auto U) {
                constexpr int pu = decltype(U)::value, u = pu & 3;        // pu: k-step within the unrolled group of RING / 4 chunks
                const int c = c0 + (pu >> 2);
#ifdef SMK_LN_STAMPS
                LN_STAMP(t_u);
                if (u > 0) sum_u[u - 1] += t_u - t_prev;
                else if (c > 0) sum_u[3] += t_u - t_prev;
                t_prev = t_u;
#endif
                {   // refill the ring slot consumed one k-step ago (k index wraps: the next tile uses the same weights)
                    int kn = c * 4 + u + RING - 1;
                    kn = kn >= nks ? kn - nks : kn;
                    kn = __builtin_amdgcn_readfirstlane(kn);
                    bqh[(pu + RING - 1) % RING] = load_b(kn, 0);
                    bql[(pu + RING - 1) % RING] = load_b(kn, 1);
                }
                // The staged registers hold the chunk after this one: k-steps 0..2 each split + write a share of its 2*MB
                // pieces to the other buffer (last read before the previous chunk's barrier; complete before this chunk's) and
                // re-issue each piece's load at once for the chunk after that (a whole chunk of MFMAs to land).
                constexpr int NP = TM / RP, P0 = (NP * 3 + 7) / 8, P1 = (NP * 6 + 7) / 8;      // MB = 4: pieces 0-2 | 3-5 | 6-7
                constexpr int pbeg = u == 0 ? 0 : u == 1 ? P0 : u == 2 ? P1 : NP, pend = u == 0 ? P0 : u == 1 ? P1 : NP;
#pragma unroll
                for (int j = pbeg; j < pend; ++j) {
                    stage_store(buf ^ 1, j, LNF && c == nchunks - 1);
                    stage_load(ld_tm, ld_c, j);
                }
                if (u == 2) advance();
#ifdef SMK_LN_STAMPS
                if (u == 3) { LN_STAMP(t_b0); __syncthreads(); LN_STAMP(t_b1); sum_u[4] += t_b1 - t_b0; }
#else
#ifndef SMK_LN_NOBAR
                if (u == 3) __syncthreads();     // other buffer complete and visible; every read of this buffer has returned
#endif
#endif
                if (u & 1) load_a(u == 3 ? buf ^ 1 : buf, (u + 1) & 3, ahA, alA);
                else load_a(buf, u + 1, ahB, alB);
                const bf16x8 bh = __builtin_bit_cast(bf16x8, bqh[pu % RING]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, bql[pu % RING]);
                // product-major emission: consecutive MFMAs go to different accumulators; each still sums lo*hi, hi*lo, hi*hi in order
#pragma unroll
                for (int mi = 0; mi < MB; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, (u & 1) ? alB[mi] : alA[mi], acc[mi], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < MB; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, (u & 1) ? ahB[mi] : ahA[mi], acc[mi], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < MB; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, (u & 1) ? ahB[mi] : ahA[mi], acc[mi], 0, 0, 0);
                // Schedule of one k-step: next k-step's fragment reads, the ring refill and (k-steps 0,1) the staging work are
                // issued INSIDE the gaps of this k-step's MFMAs (left alone, hipcc sinks every ds_read to just before its
                // consumer and waits on it there).
                if (sched) {
                    // per SIMD an MFMA gap hides about five other vector-issue slots (MI355X_MICROARCH.md, constants table), shared
                    // by the two resident waves: the split arithmetic is spread at ~12 VALU per piece over the k-step's gaps
                    constexpr int NMF = 3 * MB, NDS = 2 * MB, NPC = pend - pbeg;
                    constexpr int VPER = NPC ? ((AS ? 3 : (LNF ? 26 : 14)) * NPC + NMF - 1) / NMF : 0;
#pragma unroll
                    for (int i = 0; i < NMF; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // 1 MFMA
#if SMK_LN_RINGFIRST
                        if (i < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);               // ring refill first (L2 latency)
                        if (i < NDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             // 1 DS read (next k-step's fragment)
#else
                        if (i < NDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             // 1 DS read (next k-step's fragment)
#endif
                        if (VPER) __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);             // split arithmetic
                        if (i >= NMF - NPC) {
                            __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);                      // a finished piece: 2 DS writes
                            __builtin_amdgcn_sched_group_barrier(0x020, (AS || LNF) ? 2 : 1, 0);    //   + its re-issued load(s) (LNF: + the pivot)
                        }
#if !SMK_LN_RINGFIRST
                        if (i >= NMF - 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // ring refill
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            auto chunk = [&](auto J) {
                constexpr int j = decltype(J)::value;
                kstep(std::integral_constant<int, 4 * j>{});
                kstep(std::integral_constant<int, 4 * j + 1>{});
                kstep(std::integral_constant<int, 4 * j + 2>{});
                kstep(std::integral_constant<int, 4 * j + 3>{});
                buf ^= 1;
            };
            chunk(std::integral_constant<int, 0>{});
            if constexpr (RING >= 8) chunk(std::integral_constant<int, 1>{});
            if constexpr (RING >= 16) {
                chunk(std::integral_constant<int, 2>{});
                chunk(std::integral_constant<int, 3>{});
            }
        }

#ifdef SMK_LN_STAMPS
        LN_STAMP(t_k1);
#endif
        if (KS > 1) {   // merge the wave groups' partial sums: groups 1.. park theirs in LDS, group 0 adds them in group order
            float *xch = reinterpret_cast<float *>(smem_all + KS * ln_lds_bytes<MB, NW>());
            if (grp > 0) {
                float *dstp = xch + ((size_t)(grp - 1) * (NW * 64) + tid) * (MB * 16);
#pragma unroll
                for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4 *>(dstp + mi * 16 + 4 * q) =
                            make_float4(acc[mi][4 * q], acc[mi][4 * q + 1], acc[mi][4 * q + 2], acc[mi][4 * q + 3]);
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int gsrc = 1; gsrc < KS; ++gsrc) {
                    const float *srcp = xch + ((size_t)(gsrc - 1) * (NW * 64) + tid) * (MB * 16);
#pragma unroll
                    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 o = *reinterpret_cast<const float4 *>(srcp + mi * 16 + 4 * q);
                            acc[mi][4 * q] += o.x; acc[mi][4 * q + 1] += o.y; acc[mi][4 * q + 2] += o.z; acc[mi][4 * q + 3] += o.w;
                        }
                }
            }
            // (the next write to xch by groups 1.. lies behind at least one more workgroup barrier -- the next tile's first chunk --
            //  which group 0 reaches only after these reads)
        }
        if constexpr (LNF) {   // row statistics: the 16 threads sc = 0 .. 15 of a staging row hold its 64 k of every chunk between them
#pragma unroll
            for (int j = 0; j < NPC_ALL; ++j) {
                float s1 = ln_s[j], s2 = ln_q[j];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
                if (sc == 0) {
                    const float dm = s1 / (float)K;                       // mean - pivot
                    float var = s2 / (float)K - dm * dm;
                    var = var > 0.f ? var : 0.f;
                    stat_s[2 * (sr + RP * j)] = dm;
                    stat_s[2 * (sr + RP * j) + 1] = 1.0f / sqrtf(var + a.c.ln_eps);
                }
                ln_s[j] = ln_ns[j]; ln_q[j] = ln_nq[j];                   // the next tile's first chunk is already in
            }
            __syncthreads();
        }
        // ---- epilogue.  The weights are the MFMA's row operand, so acc[mi][4q + i] = output row mi*32 + r (this lane's token),
        //      column 8q + 4hi + i of the wave's 32: four consecutive columns per lane -> 16-byte loads and stores.
        if (grp == 0 && nw_ok && !LN_DBG(a, 4)) {
            const int row0 = tm * TM, ncol = tn * TN + wave * 32 + 4 * hi;
            const float *bias_w = bias_s + wave * 32 + 4 * hi;
            // Global loads (residual, periodic addend) of row block mi+1 are issued BEFORE the stores of block mi: vmcnt retires in
            // order, so a load issued after a store could only be consumed once that store had been acknowledged by memory.
            float4 ex[2][4];                                 // per column group: residual (+ addend) of the block being finished
            const bool any_ex = a.c.res || a.c.padd;
            auto fetch_extra = [&](int mi, float4 (&e)[4]) {
                const int row = row0 + mi * 32 + r;
#pragma unroll
                for (int q = 0; q < 4; ++q) e[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row >= M) return;
                if (a.c.padd) {                              // a tile lies inside one group (launcher: rows_per_group % TM == 0)
                    const int grp = row0 / a.c.rows_per_group;
                    const int xx = row - grp * a.c.rows_per_group;                      // (row in group) mod period, no integer division
                    int ph = xx - (int)((float)xx * (1.0f / (float)a.c.period)) * a.c.period;
                    ph = ph < 0 ? ph + a.c.period : (ph >= a.c.period ? ph - a.c.period : ph);
                    const float *pp = a.c.padd + ((size_t)grp * a.c.period + ph) * N + ncol;
#pragma unroll
                    for (int q = 0; q < 4; ++q) e[q] = *reinterpret_cast<const float4 *>(pp + 8 * q);
                } else {
                    const float *rp = a.c.res + (long long)row * a.c.ldr + ncol;
#pragma unroll
                    for (int q = 0; q < 4; ++q) e[q] = *reinterpret_cast<const float4 *>(rp + 8 * q);
                }
            };
            // activation + store of 4 consecutive columns of one row (fp32 or split-bf16)
            auto finish = [&](float (&v)[4], int row, int q) {
                if (a.c.act == 1) {                       // wave-uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
                } else if (a.c.act == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
                }
            };
            auto store4 = [&](const float (&v)[4], int row, int q) {
                if (row >= M) return;
                if (a.c.y_split) {   // SMK_FMT_SPLIT_BF16: group (ncol + 8q) / 8 of the row, this lane's half (4 hi | 4 lo)
                    bf16x4 vh, vl;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const __bf16 h = (__bf16)v[i];
                        vh[i] = h;
                        vl[i] = (__bf16)(v[i] - (float)h);
                    }
                    __bf16 *ys = reinterpret_cast<__bf16 *>(a.c.y) + (size_t)row * (2 * N) + ((ncol >> 3) + q) * 16 + 4 * hi;
                    *reinterpret_cast<bf16x4 *>(ys) = vh;
                    *reinterpret_cast<bf16x4 *>(ys + 8) = vl;
                } else if (LN_DBG(a, 8)) {   // timing ablation: the same bytes as 8 rows x 128 contiguous bytes per store instruction (values misplaced)
                    const int L = r + 32 * hi, rr = row - r + 8 * q + (L >> 3);
                    if (rr < M) *reinterpret_cast<float4 *>(y_seg + (long long)rr * a.c.ldy + (ncol - 4 * hi) + (L & 7) * 4) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    const bool sp = a.c.split_from >= 0 && ncol + 8 * q >= a.c.split_from;        // wave-uniform (split_from % 32 == 0)
                    *reinterpret_cast<float4 *>(y_seg + (long long)row * a.c.ldy + ncol + 8 * q) = sp ? split4_inplace(v) : make_float4(v[0], v[1], v[2], v[3]);
                }
            };
            // fused LayerNorm: v = rstd ((x - p) W'^T - (mean - p) wsum) + b' for row mi*32 + r, columns 8q + 4hi .. + 3 of the wave's 32
            auto ln_finish = [&](float (&v)[4], int mi, int q, const float4 &bq) {
                const float mean = stat_s[2 * (mi * 32 + r)], rstd = stat_s[2 * (mi * 32 + r) + 1];      // ("mean" = mean - pivot)
                const float4 wq4 = *reinterpret_cast<const float4 *>(wsum_s + wave * 32 + 4 * hi + 8 * q);
                v[0] = rstd * (acc[mi][4 * q] - mean * wq4.x) + bq.x;
                v[1] = rstd * (acc[mi][4 * q + 1] - mean * wq4.y) + bq.y;
                v[2] = rstd * (acc[mi][4 * q + 2] - mean * wq4.z) + bq.z;
                v[3] = rstd * (acc[mi][4 * q + 3] - mean * wq4.w) + bq.w;
            };
            if (!any_ex) {
                // plain path: no global loads at all, so nothing ever waits on vmcnt -- which on CDNA4 also counts the stores
                // (a wait here would drain every store to memory before the next one is issued)
#pragma unroll
                for (int mi = 0; mi < MB; ++mi) {
                    const int row = row0 + mi * 32 + r;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 bq = *reinterpret_cast<const float4 *>(bias_w + 8 * q);
                        float v[4] = {acc[mi][4 * q] + bq.x, acc[mi][4 * q + 1] + bq.y, acc[mi][4 * q + 2] + bq.z, acc[mi][4 * q + 3] + bq.w};
                        if constexpr (LNF) ln_finish(v, mi, q, bq);
                        finish(v, row, q);
                        store4(v, row, q);
                    }
                }
            } else {
                fetch_extra(0, ex[0]);
#pragma unroll
                for (int mi = 0; mi < MB; ++mi) {
                    if (mi + 1 < MB) fetch_extra(mi + 1, ex[(mi + 1) & 1]);
                    const int row = row0 + mi * 32 + r;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 bq = *reinterpret_cast<const float4 *>(bias_w + 8 * q);
                        const float4 eq = ex[mi & 1][q];
                        float v[4] = {acc[mi][4 * q] + bq.x, acc[mi][4 * q + 1] + bq.y, acc[mi][4 * q + 2] + bq.z, acc[mi][4 * q + 3] + bq.w};
                        if constexpr (LNF) ln_finish(v, mi, q, bq);
                        if (a.c.padd) { v[0] += eq.x; v[1] += eq.y; v[2] += eq.z; v[3] += eq.w; }
                        finish(v, row, q);
                        if (!a.c.padd) { v[0] = eq.x + v[0]; v[1] = eq.y + v[1]; v[2] = eq.z + v[2]; v[3] = eq.w + v[3]; }
                        store4(v, row, q);
                    }
                }
            }
        }
#ifdef SMK_LN_STAMPS
        LN_STAMP(t_e);
        sum_k += t_k1 - t_k0; sum_e += t_e - t_k1; ++ntl;
#endif
    }
#ifdef SMK_LN_STAMPS
    LN_STAMP(t_end);
    LN_RSTAMP(r_end);
    if (a.stamps && lane == 0) {
        unsigned long long *rec = a.stamps + (((size_t)blockIdx.x * NW + wave) & 4095) * 8;
        rec[0] = sum_k; rec[1] = sum_e; rec[2] = t_end - t_begin; rec[3] = r_end - r_begin; rec[4] = ntl; rec[5] = sum_u[0] | (sum_u[1] << 32); rec[6] = sum_u[2] | (sum_u[3] << 32); rec[7] = sum_u[4];
    }
#endif
}

template <int MB, int NW, bool AS, int KS = 1, int RING = LN_RING, bool LNF = false>
static hipError_t launch_mb(const LinearArgs &a, hipStream_t st) {
    constexpr int lds = KS * ln_lds_bytes<MB, NW>() + (KS - 1) * NW * 64 * MB * 16 * 4;
    once_per_device((const void *)k_linear_x3<MB, NW, AS, KS, RING, LNF>, [&] {
        (void)hipFuncSetAttribute((const void *)k_linear_x3<MB, NW, AS, KS, RING, LNF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    const int nseg = a.c.nseg;
    const int nwg_max = (NW == 8 || KS > 1 ? 1 : 2) * a.num_cu / nseg;     // 8 waves per CU either way (split-K: one workgroup per CU)
    const long long tiles = (long long)a.tiles_m * a.tiles_n;
    long long nwg = tiles < nwg_max ? tiles : nwg_max;
    nwg -= nwg % a.tiles_n;                               // every workgroup keeps one column tile
    if (nwg < a.tiles_n) nwg = a.tiles_n;
    nwg *= nseg;                                          // per segment: the same walk over its own [M][N] slab
    LinearArgs b = a;
    static int swz_env = -1, stg_env = -1;
    if (swz_env < 0) { const char *s = getenv("SMK_LINEAR_SWZ"); swz_env = s ? atoi(s) : 1; }
    if (stg_env < 0) { const char *s = getenv("SMK_LINEAR_STAGGER"); stg_env = s ? atoi(s) : 0; }
    b.swz = swz_env && nwg % (8 * a.tiles_n) == 0;
    b.stagger = stg_env;
    // one tile ~ (K/64) chunks x ~4.2 K cycles + ~9 K epilogue; s_sleep(127) ~ 8 K cycles
    b.stagger_unit = stg_env > 1 ? (int)(((a.l.K / 64) * 4200 + 9000) / 8128 / stg_env) : 0;
    if (b.stagger_unit < 1) b.stagger = 0;
    hipLaunchKernelGGL((k_linear_x3<MB, NW, AS, KS, RING, LNF>), dim3((unsigned)nwg), dim3(NW * 64 * KS), lds, st, b);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// k_linear_b16: the same layer on the 16x16x32 MFMA shape (v_mfma_f32_16x16x32_bf16) -- the form the full-size calls take (128-row
// tiles, fp32 activations in and out).  Cycles per flop are those of the 32x32x16 shape; the point is the power limit: with the whole
// chip in this kernel the shader clock falls to 1.3-1.7 GHz (stamps above), and at equal cycles the 16x16x32 shape sustains a higher
// clock (MI355X_MICROARCH.md "Shape": 1.12-1.15 x the FLOP/s; k_encoder_b16 gained 20 % from the same change).
//   tile 128 tokens x 32 NW columns, wave w owns columns 32w .. 32w+31 as 2 column blocks x 8 token blocks of 16 x 16 (64 accumulator
//   registers, as before); the WEIGHTS are the MFMA's row operand, so a lane ends with 4 consecutive output columns of one token row
//   per accumulator -> 16-byte stores.  The unit of the loop is half a 32-k step: 4 token blocks x 2 column blocks x 3 products = 24
//   MFMAs = 384 matrix-pipe cycles (the old k-step), so the skeleton carries over: four units per 64-k chunk, the next unit's 8 token
//   fragments read under this unit's MFMAs, staging of the chunk after next spread over units 0-2, one barrier per chunk before unit 3's
//   reads of the other buffer.  Weight fragments come from the SAME pre-split layout ([K/16][hi|lo][N][16]: lane group g = lane >> 4
//   takes k = 8g .. 8g+7 of the 32-k step, i.e. half of k16 block (g >> 1)) through a ring two 32-k steps deep, all four loads of step
//   s+1 issued behind the first MFMAs of step s.
//   Token image in LDS: [128 rows][64 k] bf16 per plane, UNPADDED 128-byte rows with the 16-byte unit u of row r stored at u ^ (r & 7):
//   conflict-free for this shape's ds_read_b128 (lanes {0-3, 12-15, 20-27} together: 8 rows of one k-group with 8 rows of the next) and
//   for the 8-byte staging stores, found by search over pitches x swizzles; 65 KB per workgroup instead of 74.
//   CONV: the implicit-GEMM form of a 3 x 3 x 3, 64-channel Conv3d on a channels-last volume (SPEC_3D.md section 8): output row = voxel,
//   chunk c = tap c, and the chunk's 64 k of a row are the 64 contiguous channels of the voxel at (z + dz, y + dy, x + dx) -- the same
//   256-byte row piece a linear layer stages, at a shifted address, or zeros outside the volume (the offset is then moved past the buffer's
//   range, where the hardware returns 0: no branch in the staging path).  No patch matrix ever exists in memory.
//   CONV = 2: the 7 x 7 x 7 convolution of a SCALAR volume [Dl][H][W] (SPEC_3D.md section 8, conv1) the same way: the K range is laid out as
//   56 (kz, ky) window rows x 8 kx slots (row r = kz * 7 + ky; rows >= 49 and slot 7 carry zero weights), chunk c = window rows 8c .. 8c+7,
//   and a staged 16-byte piece = 4 consecutive kx taps = 4 consecutive x of the input (one unaligned dwordx4 load); elements past the
//   row's ends are zeroed by selects, window rows outside the volume by the out-of-range offset.  Waves whose 32 output columns lie
//   beyond N (N = 64: two of four) stage their share of the tile and skip the MFMAs.
//   R: depth of the weight ring in 32-k steps.  R = 2 requests step s+1 at the start of step s (768 matrix-pipe cycles ahead for the two
//   waves of a SIMD: about an L2 round trip under load); R = 4 requests step s+3 (the chunk loop is unrolled twice so that the slots stay
//   compile-time indices; needs an even number of 64-k chunks).
// LNF: LayerNorm fused in front, as in k_linear_x3 (pivot-shifted rows, statistics gathered while the rows are staged and carried across the
// tiles of the chunk stream, rstd ((x - p) W'^T - (mean - p) wsum) + b' in the epilogue); plain layers only (CONV = 0).
template <int NW, int CONV = 0, int R = 2, bool LNF = false>
__global__ __launch_bounds__(NW * 64, 2) void k_linear_b16(const LinearArgs a) {
    static_assert(!LNF || CONV == 0, "fused LayerNorm: plain layers");
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int TM = 128, TN = NW * 32, RP = NW * 4, NP = TM / RP, PLANE = TM * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.l.K, N = a.l.N, M = a.c.M;
    const int nchunks = K >> 6, nks = nchunks * 2;           // 32-k steps
    const int vid = a.swz ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;    // XCD-contiguous ids (see k_linear_x3)
    const int tn = vid % a.tiles_n, tm_step = (int)gridDim.x / a.tiles_n;
    int tm = vid / a.tiles_n;
    const int ncol0 = tn * TN + wave * 32;
    const bool nw_ok = ncol0 < N;                            // N % 32 == 0: a wave's 32 columns are all inside or all outside

    // ---- weight ring: per 32-k step four 16-byte loads per lane (column block 0 / 1 x hi / lo) at a lane-constant offset from a scalar base
    const int lane_b = nw_ok ? (g >> 1) * (N * 64) + (ncol0 + l16) * 32 + (g & 1) * 16 : 0;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(a.l.wq), 0, K * N * 4, 0x00020000);
    const int step_bytes = N * 128, part_bytes = N * 32;     // one 32-k step = two k16 blocks x (hi | lo) planes of N x 32 bytes
    uint4 bq[R][4];                                          // [ring slot = 32-k step mod R][nb * 2 + part]
    auto load_b = [&](int slot, int kn) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_b + nb * 512, kn * step_bytes + part * part_bytes, 0);
                bq[slot][nb * 2 + part] = make_uint4(v[0], v[1], v[2], v[3]);
            }
    };
#pragma unroll
    for (int k = 0; k < R - 1; ++k) load_b(k, k < nks ? k : k - nks);

    // ---- token staging (as k_linear_x3, fp32 source): thread = float4 column sc of rows sr, sr + RP, ...
    const int sc = tid & 15, sr = tid >> 4;
    float4 stage[NP];
    // (see k_linear_x3; here the current tile's sums are closed at the START of its last chunk -- every piece of the tile has been stored by
    // then -- and the same registers then gather the next tile's first chunk: three registers per staged row instead of five)
    float ln_piv[LNF ? NP : 1], ln_s[LNF ? NP : 1], ln_q[LNF ? NP : 1];
#pragma unroll
    for (int j = 0; j < (LNF ? NP : 1); ++j) { ln_piv[j] = ln_s[j] = ln_q[j] = 0.f; }
    const unsigned xbytes = CONV == 1 ? (unsigned)((long long)a.cDl * a.cH * a.cW * 256)
                          : CONV == 2 ? (unsigned)((long long)a.cDl * a.cH * a.cW * 4)
                                      : (unsigned)(((long long)(M - 1) * a.c.ldx + K) * 4);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.c.x), 0, (int)xbytes, 0x00020000);
    const int ldxb = CONV ? 256 : (int)a.c.ldx * 4;
    const int lane_x = sr * ldxb + sc * 16;
    // CONV: (x, y, z) of this thread's NP staging rows of the tile being loaded, packed x | y << 10 | z << 20 (recomputed when the chunk
    // stream moves to the next tile)
    unsigned vxyz[CONV ? NP : 1];
    auto conv_coords = [&](int tmx) {
        if constexpr (CONV != 0) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int m = tmx * TM + RP * j + sr;
                const int plane = a.cH * a.cW, zr = m / plane, rem = m - zr * plane, yy = rem / a.cW;
                const int zz = zr + a.cz_off;
                vxyz[j] = (unsigned)(rem - yy * a.cW) | (unsigned)yy << 10 | (unsigned)(zz > 1023 ? 1023 : zz) << 20;     // (z past the slab: any invalid value)
            }
        }
    };
    auto stage_load = [&](int tmx, int cx, int j) {
        if constexpr (CONV == 1) {
            const int dz = cx / 9 - 1, dy = (cx / 3) % 3 - 1, dx = cx % 3 - 1;                // wave-uniform (scalar)
            const int x = (int)(vxyz[j] & 1023u) + dx, y = (int)((vxyz[j] >> 10) & 1023u) + dy, z = (int)(vxyz[j] >> 20) + dz;
            const bool ok = (unsigned)x < (unsigned)a.cW && (unsigned)y < (unsigned)a.cH && (unsigned)z < (unsigned)a.cDl;
            const unsigned lin = (unsigned)((z * a.cH + y) * a.cW + x);
            const unsigned off = ok ? lin * 256u + (unsigned)sc * 16u : 0xfffffff0u;          // past the buffer: reads as zero in hardware
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0);
            stage[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        } else if constexpr (CONV == 2) {
            const int r = cx * 8 + (sc >> 1);                                                 // window row kz * 7 + ky (>= 49: zero weights)
            const int kz = r / 7, dz = kz - 3, dy = r - kz * 7 - 3, dx0 = (sc & 1) ? 1 : -3;
            const int x = (int)(vxyz[j] & 1023u) + dx0, y = (int)((vxyz[j] >> 10) & 1023u) + dy, z = (int)(vxyz[j] >> 20) + dz;
            const bool ok = r < 49 && (unsigned)y < (unsigned)a.cH && (unsigned)z < (unsigned)a.cDl;
            const int lin = (z * a.cH + y) * a.cW + x;                                        // may point before the row's start: masked below
            // a piece that starts before the buffer (lin < 0: only with y = z = 0) would wrap: those elements are masked anyway, so read from 0
            const unsigned off = ok ? (unsigned)(lin < 0 ? 0 : lin) * 4u : 0xfffffff0u;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0);
            const int sh = lin < 0 ? -lin : 0;                                                // elements shifted by the clamp to offset 0
            float e[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
            if (sh) {                                                                         // (x + i < 0 for i < sh <= 3: take element i - sh; masked below anyway)
                const float t0 = e[0], t1 = e[1], t2 = e[2];
                e[3] = sh == 1 ? t2 : (sh == 2 ? t1 : t0); e[2] = sh == 1 ? t1 : t0; e[1] = t0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i] = (unsigned)(x + i) < (unsigned)a.cW ? e[i] : 0.f;
            stage[j] = make_float4(e[0], e[1], e[2], e[3]);
        } else {
            const unsigned off = ((unsigned)tmx * TM + RP * j) * (unsigned)ldxb + (unsigned)cx * 256u;    // rows past M read as zero in hardware
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(off + (unsigned)lane_x), 0, 0);
            stage[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            if constexpr (LNF)       // the row's pivot x[row][0], beside the piece (one address per staging row)
                ln_piv[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)(((unsigned)tmx * TM + RP * j + (unsigned)sr) * (unsigned)ldxb), 0, 0));
        }
    };
    // row sr + RP j: RP is a multiple of 8, so (row & 7) = sr & 7 for every piece
    const int st_off = sr * 128 + ((((sc >> 1) ^ (sr & 7))) << 4) + (sc & 1) * 8;
    auto stage_store = [&](int buf, int j) {
        unsigned char *ph = smem + buf * 2 * PLANE + st_off + RP * j * 128;
        float v[4] = {stage[j].x, stage[j].y, stage[j].z, stage[j].w};
        if constexpr (LNF) {
            const float pv = ln_piv[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = v[i] - pv;
            ln_s[j] += (v[0] + v[1]) + (v[2] + v[3]);
            ln_q[j] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        bf16x4 vh, vl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __bf16 h = (__bf16)v[i];
            vh[i] = h;
            vl[i] = (__bf16)(v[i] - (float)h);
        }
        *reinterpret_cast<bf16x4 *>(ph) = vh;
        *reinterpret_cast<bf16x4 *>(ph + PLANE) = vl;
    };
    int ld_tm = tm, ld_c = 0;
    auto advance = [&]() {
        if (++ld_c == nchunks) {
            ld_c = 0;
            ld_tm = ld_tm + tm_step < a.tiles_m ? ld_tm + tm_step : a.tiles_m;
            conv_coords(ld_tm);                              // (past the last tile: rows >= M, whose plane index is outside the slab -> zeros)
        }
    };
    conv_coords(ld_tm);
#pragma unroll
    for (int j = 0; j < NP; ++j) stage_load(ld_tm, ld_c, j);
    advance();
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        stage_store(0, j);
        stage_load(ld_tm, ld_c, j);
    }
    advance();
    float *bias_s = reinterpret_cast<float *>(smem + 4 * PLANE);
    if (tid < TN) bias_s[tid] = (a.l.bias && tn * TN + tid < N) ? a.l.bias[tn * TN + tid] : 0.f;
    float *wsum_s = bias_s + 256, *stat_s = wsum_s + 256;    // LNF: column sums of W', then (mean - pivot, rstd) per tile row
    if (LNF && tid < TN) wsum_s[tid] = tn * TN + tid < N ? a.c.ln_wsum[tn * TN + tid] : 0.f;
    __syncthreads();

    // token fragments of unit u (32-k step u >> 1, token blocks 4 (u & 1) .. + 3): lane = (token l16 of the block, k-group g)
    const int fa0 = l16 * 128 + ((g ^ (lane & 7)) << 4);      // 32-k step 0 of the chunk; step 1 = the unit index ^ 4, i.e. offset ^ 64
    auto load_a = [&](int buf, int u, bf16x8 (&ah)[4], bf16x8 (&al)[4]) {
        const unsigned char *p = smem + buf * 2 * PLANE + ((u & 2) ? (fa0 ^ 64) : fa0) + (u & 1) * (4 * 2048);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ah[t] = *reinterpret_cast<const bf16x8 *>(p + t * 2048);
            al[t] = *reinterpret_cast<const bf16x8 *>(p + PLANE + t * 2048);
        }
    };
    bf16x8 ahA[4], alA[4], ahB[4], alB[4];
    int buf = 0;
    load_a(0, 0, ahA, alA);

#ifdef SMK_LN_STAMPS
    unsigned long long sum_k = 0, sum_e = 0, ntl = 0;
    LN_STAMP(t_begin);
    LN_RSTAMP(r_begin);
#endif
    // LNF: at the start of a tile's LAST chunk every piece of the tile has gone through stage_store: close its statistics (the 16 threads
    // sc = 0 .. 15 of a staging row hold its 64 k of every chunk between them) and hand the registers to the next tile, whose first
    // chunk is staged during this last chunk.  stat_s is read by this tile's epilogue; the previous tile's epilogue lies behind at
    // least one chunk barrier.
    auto ln_close = [&](int cc) {
        if constexpr (LNF) {
            if (cc != nchunks - 1) return;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                float s1 = ln_s[j], s2 = ln_q[j];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
                if (sc == 0) {
                    const float dm = s1 / (float)K;                       // mean - pivot
                    float var = s2 / (float)K - dm * dm;
                    var = var > 0.f ? var : 0.f;
                    stat_s[2 * (sr + RP * j)] = dm;
                    stat_s[2 * (sr + RP * j) + 1] = 1.0f / sqrtf(var + a.c.ln_eps);
                }
                ln_s[j] = 0.f; ln_q[j] = 0.f;
            }
        }
    };
    for (; tm < a.tiles_m; tm += tm_step) {
#ifdef SMK_LN_STAMPS
        LN_STAMP(t_k0);
#endif
        f32x4 acc[2][8];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[nb][t] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
        for (int c = 0; c < nchunks; c += R / 2) {
            // PU = 4 * (chunk of this iteration: 0, or 1 with R = 4) + unit; cc = that chunk's index
            auto unit = [&](auto PU, int cc) {
                constexpr int pu = decltype(PU)::value, u = pu & 3, ks = u >> 1, h = u & 1, slot = (2 * (pu >> 2) + ks) % R;
                if (h == 0) {                               // all four weight loads of 32-k step s + R - 1, behind this step's first MFMAs
                    int kn = cc * 2 + ks + R - 1;
                    kn = kn >= nks ? kn - nks : kn;          // wraps: the next tile uses the same weights
                    load_b((slot + R - 1) % R, __builtin_amdgcn_readfirstlane(kn));
                }
                constexpr int P0 = (NP * 3 + 7) / 8, P1 = (NP * 6 + 7) / 8;
                constexpr int pbeg = u == 0 ? 0 : u == 1 ? P0 : u == 2 ? P1 : NP, pend = u == 0 ? P0 : u == 1 ? P1 : NP;
#pragma unroll
                for (int j = pbeg; j < pend; ++j) {
                    stage_store(buf ^ 1, j);
                    stage_load(ld_tm, ld_c, j);
                }
                if (u == 2) advance();
                if (u == 3) __syncthreads();                // other buffer complete and visible; every read of this buffer has returned
                if (u & 1) load_a(u == 3 ? buf ^ 1 : buf, (u + 1) & 3, ahA, alA);
                else load_a(buf, u + 1, ahB, alB);
                bf16x8 wh[2], wl[2];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    wh[nb] = __builtin_bit_cast(bf16x8, bq[slot][nb * 2]);
                    wl[nb] = __builtin_bit_cast(bf16x8, bq[slot][nb * 2 + 1]);
                }
                // product-major: consecutive MFMAs go to different accumulators; each accumulator still sums hi*lo, lo*hi, hi*hi in order
                if (CONV == 2 && !nw_ok) return;             // (a wave with no output columns: staging only)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[nb][4 * h + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[nb], (u & 1) ? alB[t] : alA[t], acc[nb][4 * h + t], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[nb][4 * h + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[nb], (u & 1) ? ahB[t] : ahA[t], acc[nb][4 * h + t], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[nb][4 * h + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[nb], (u & 1) ? ahB[t] : ahA[t], acc[nb][4 * h + t], 0, 0, 0);
                {
                    // one fragment read per third MFMA, the weight loads right behind the first MFMAs (k_encoder_b16's placement), the
                    // split arithmetic spread over the gaps (an MFMA of this shape leaves 8 of its 16 cycles to other vector issue)
                    constexpr int NMF = 24, NPC = pend - pbeg, VPER = NPC ? ((LNF ? 26 : 14) * NPC + NMF - 1) / NMF : 0;
#pragma unroll
                    for (int i = 0; i < NMF; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // 1 MFMA
                        if (h == 0 && i < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // weight ring (L2 latency)
                        if (i % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // 1 DS read (next unit's fragment)
                        if (VPER) __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);             // split arithmetic
                        if (i >= NMF - NPC) {
                            __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);                      // a finished piece: 2 DS writes
                            __builtin_amdgcn_sched_group_barrier(0x020, LNF ? 2 : 1, 0);            //   + its re-issued load (LNF: + the pivot)
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if constexpr (LNF) ln_close(c);
            unit(std::integral_constant<int, 0>{}, c);
            unit(std::integral_constant<int, 1>{}, c);
            unit(std::integral_constant<int, 2>{}, c);
            unit(std::integral_constant<int, 3>{}, c);
            buf ^= 1;
            if constexpr (R == 4) {
                if constexpr (LNF) ln_close(c + 1);
                unit(std::integral_constant<int, 4>{}, c + 1);
                unit(std::integral_constant<int, 5>{}, c + 1);
                unit(std::integral_constant<int, 6>{}, c + 1);
                unit(std::integral_constant<int, 7>{}, c + 1);
                buf ^= 1;
            }
        }

#ifdef SMK_LN_STAMPS
        LN_STAMP(t_k1);
#endif
        // ---- epilogue: acc[nb][t][i] = token row t*16 + l16, output column ncol0 + nb*16 + 4g + i
        if (nw_ok) {
            const int row0 = tm * TM, ncol = ncol0 + 4 * g;
            const float *bias_w = bias_s + wave * 32 + 4 * g;
            const float4 b0 = *reinterpret_cast<const float4 *>(bias_w), b1 = *reinterpret_cast<const float4 *>(bias_w + 16);
            const bool any_ex = a.c.res || a.c.padd;
            float4 ex[2][2];
            auto fetch_extra = [&](int t, float4 (&e)[2]) {
                const int row = row0 + t * 16 + l16;
                e[0] = e[1] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row >= M) return;
                if (a.c.padd) {                              // a tile lies inside one group (launcher: rows_per_group % 128 == 0)
                    const int grp = row0 / a.c.rows_per_group;
                    const int xx = row - grp * a.c.rows_per_group;
                    int ph = xx - (int)((float)xx * (1.0f / (float)a.c.period)) * a.c.period;
                    ph = ph < 0 ? ph + a.c.period : (ph >= a.c.period ? ph - a.c.period : ph);
                    const float *pp = a.c.padd + ((size_t)grp * a.c.period + ph) * N + ncol;
                    e[0] = *reinterpret_cast<const float4 *>(pp);
                    e[1] = *reinterpret_cast<const float4 *>(pp + 16);
                } else {
                    const float *rp = a.c.res + (long long)row * a.c.ldr + ncol;
                    e[0] = *reinterpret_cast<const float4 *>(rp);
                    e[1] = *reinterpret_cast<const float4 *>(rp + 16);
                }
            };
            auto finish = [&](float (&v)[4]) {
                if (a.c.act == 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
                } else if (a.c.act == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
                }
            };
            if (any_ex) fetch_extra(0, ex[0]);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                // loads of block t+1 before the stores of block t: vmcnt retires stores in order with loads
                if (any_ex && t + 1 < 8) fetch_extra(t + 1, ex[(t + 1) & 1]);
                const int row = row0 + t * 16 + l16;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const float4 bb = nb ? b1 : b0;
                    float v[4] = {acc[nb][t][0] + bb.x, acc[nb][t][1] + bb.y, acc[nb][t][2] + bb.z, acc[nb][t][3] + bb.w};
                    if constexpr (LNF) {   // rstd ((x - p) W'^T - (mean - p) wsum) + b' for row t*16 + l16, columns nb*16 + 4g .. + 3 of the wave's 32
                        const float dm = stat_s[2 * (t * 16 + l16)], rstd = stat_s[2 * (t * 16 + l16) + 1];
                        const float4 wq4 = *reinterpret_cast<const float4 *>(wsum_s + wave * 32 + 4 * g + 16 * nb);
                        v[0] = rstd * (acc[nb][t][0] - dm * wq4.x) + bb.x;
                        v[1] = rstd * (acc[nb][t][1] - dm * wq4.y) + bb.y;
                        v[2] = rstd * (acc[nb][t][2] - dm * wq4.z) + bb.z;
                        v[3] = rstd * (acc[nb][t][3] - dm * wq4.w) + bb.w;
                    }
                    if (any_ex) {
                        const float4 eq = ex[t & 1][nb];
                        if (a.c.padd) { v[0] += eq.x; v[1] += eq.y; v[2] += eq.z; v[3] += eq.w; }
                        finish(v);
                        if (!a.c.padd) { v[0] = eq.x + v[0]; v[1] = eq.y + v[1]; v[2] = eq.z + v[2]; v[3] = eq.w + v[3]; }
                    } else {
                        finish(v);
                    }
                    if (row < M) {
                        const bool sp = a.c.split_from >= 0 && ncol + 16 * nb >= a.c.split_from;  // wave-uniform (split_from % 32 == 0)
                        *reinterpret_cast<float4 *>(a.c.y + (long long)row * a.c.ldy + ncol + 16 * nb) = sp ? split4_inplace(v) : make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
        }
#ifdef SMK_LN_STAMPS
        LN_STAMP(t_e);
        sum_k += t_k1 - t_k0; sum_e += t_e - t_k1; ++ntl;
#endif
    }
#ifdef SMK_LN_STAMPS
    LN_STAMP(t_end);
    LN_RSTAMP(r_end);
    if (a.stamps && lane == 0) {
        unsigned long long *rec = a.stamps + (((size_t)blockIdx.x * NW + wave) & 4095) * 8;
        rec[0] = sum_k; rec[1] = sum_e; rec[2] = t_end - t_begin; rec[3] = r_end - r_begin; rec[4] = ntl; rec[5] = 0; rec[6] = 0; rec[7] = 0;
    }
#endif
}

template <int NW, int CONV = 0, int R = 2, bool LNF = false>
static hipError_t launch_b16(const LinearArgs &a, hipStream_t st) {
    constexpr int lds = 4 * 128 * 128 + 1024 + (LNF ? 2048 : 0);          // + bias tile (+ LNF: wsum tile, row statistics)
    once_per_device((const void *)k_linear_b16<NW, CONV, R, LNF>, [&] {
        (void)hipFuncSetAttribute((const void *)k_linear_b16<NW, CONV, R, LNF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    const int nwg_max = (NW == 8 ? 1 : 2) * a.num_cu;        // 8 waves per CU either way
    const long long tiles = (long long)a.tiles_m * a.tiles_n;
    long long nwg = tiles < nwg_max ? tiles : nwg_max;
    nwg -= nwg % a.tiles_n;                                  // every workgroup keeps one column tile
    if (nwg < a.tiles_n) nwg = a.tiles_n;
    LinearArgs b = a;
    static int swz_env = -1;
    if (swz_env < 0) { const char *s = getenv("SMK_LINEAR_SWZ"); swz_env = s ? atoi(s) : 1; }
    b.swz = swz_env && nwg % (8 * a.tiles_n) == 0;
    hipLaunchKernelGGL((k_linear_b16<NW, CONV, R, LNF>), dim3((unsigned)nwg), dim3(NW * 64), lds, st, b);
    return hipGetLastError();
}

// 3 x 3 x 3 Conv3d of a channels-last 64-channel slab as an implicit GEMM on the layer kernel (K = 27 x 64; weights in the layer layout,
// column tap * 64 + c): rows = the voxels of planes z_off .. z_off + nz - 1 of the slab [Dl][H][W][64]; y [nz H W][N] + bias, activation
hipError_t launch_conv3d_cl_b16(const LinearDev &l, const float *slab, int Dl, int H, int W, int z_off, int nz, float *y, long long ldy, int act,
                                hipStream_t st) {
    if (l.K != 27 * 64 || H > 1023 || W > 1023 || Dl > 1022 || (long long)Dl * H * W * 256 >= (1LL << 32) - 256) return hipErrorInvalidValue;
    LinearArgs a{};
    a.l = l;
    a.c.x = slab; a.c.ldx = 64; a.c.y = y; a.c.ldy = ldy; a.c.res = nullptr; a.c.ldr = 0; a.c.padd = nullptr; a.c.rows_per_group = 1; a.c.period = 1;
    a.c.M = nz * H * W; a.c.act = act; a.c.x_split = 0; a.c.y_split = 0; a.c.nseg = 1;
    a.cDl = Dl; a.cH = H; a.cW = W; a.cz_off = z_off;
    a.num_cu = device_num_cu();
    a.stamps = nullptr; a.dbg = 0; a.swz = 0; a.stagger = 0; a.stagger_unit = 0;
    const int nw = (l.N >= 256 && (long long)cdiv(a.c.M, 128) * cdiv(l.N, 256) >= a.num_cu) ? 8 : 4;
    a.tiles_n = cdiv(l.N, nw * 32);
    a.tiles_m = cdiv(a.c.M, 128);
    return nw == 8 ? launch_b16<8, 1>(a, st) : launch_b16<4, 1>(a, st);
}

// 7 x 7 x 7 Conv3d of a scalar slab [Dl][H][W] -> N channels as an implicit GEMM (k_linear_b16<4, 2>): weights = a layer handle with
// K = 448, column (kz * 7 + ky) * 8 + kx (slot kx = 7 and rows >= 49 zero); rows = the voxels of planes z_off .. z_off + nz - 1
hipError_t launch_conv3d_s7_b16(const LinearDev &l, const float *slab, int Dl, int H, int W, int z_off, int nz, float *y, long long ldy, int act,
                                hipStream_t st) {
    if (l.K != 448 || H > 1023 || W > 1023 || Dl > 1022 || (long long)Dl * H * W * 4 >= (1LL << 32) - 256) return hipErrorInvalidValue;
    LinearArgs a{};
    a.l = l;
    a.c.x = slab; a.c.ldx = 64; a.c.y = y; a.c.ldy = ldy; a.c.res = nullptr; a.c.ldr = 0; a.c.padd = nullptr; a.c.rows_per_group = 1; a.c.period = 1;
    a.c.M = nz * H * W; a.c.act = act; a.c.x_split = 0; a.c.y_split = 0; a.c.nseg = 1;
    a.cDl = Dl; a.cH = H; a.cW = W; a.cz_off = z_off;
    a.num_cu = device_num_cu();
    a.stamps = nullptr; a.dbg = 0; a.swz = 0; a.stagger = 0; a.stagger_unit = 0;
    a.tiles_n = cdiv(l.N, 128);
    a.tiles_m = cdiv(a.c.M, 128);
    return launch_b16<4, 2>(a, st);
}

hipError_t launch_linear_x3(const LinearDev &l, const LinearCall &c, hipStream_t st) {
    const int num_cu = device_num_cu();
    LinearArgs a;
    a.l = l;
    a.c = c;
    // 8-wave workgroups (128 x 256 tile) halve the per-MFMA staging work (split arithmetic, LDS writes) and the re-reads of A;
    // 4-wave ones (x 128) serve narrow layers and small problems (more workgroups)
    static int force_mb = -1, force_nw = -1;
    if (force_mb < 0) { const char *s = getenv("SMK_LINEAR_MB"); force_mb = s ? atoi(s) : 0; }
    if (force_nw < 0) { const char *s = getenv("SMK_LINEAR_NW"); force_nw = s ? atoi(s) : 0; }
    static int dbg = -1;
#ifdef SMK_LN_DIAG
    if (dbg < 0) { const char *s = getenv("SMK_LINEAR_DBG"); dbg = s ? atoi(s) : 0; }
#else
    dbg = 0;
#endif
    a.dbg = dbg;
    a.num_cu = num_cu;
    a.swz = 0;
    a.stagger = 0;
    a.stamps = nullptr;
    if (c.ln_wsum) {   // LayerNorm fused in front (k_linear_x3<.., LNF>): the tile shapes of the plain layer
        if (c.x_split || c.y_split || c.nseg != 1 || c.res || (c.padd && c.rows_per_group % 32 != 0)) return hipErrorInvalidValue;
        // 128 x 256 tiles (k_linear_b16<8>) from three quarters of a round on: the q | k | v layer at batch 4 is 192 such tiles -- ONE round on 75 %
        // of the CUs, 34 us -- against three rounds of 64 x 128 tiles on k_linear_x3, 40 us (measured: 1.160 -> 1.111 ms per batch-4 forward)
        int nwl = (l.N >= 256 && (long long)cdiv(c.M, 128) * cdiv(l.N, 256) * 4 >= 3LL * num_cu) ? 8 : 4;
        if (force_nw == 4 || force_nw == 8) nwl = force_nw;
        int mbl = 4;
        // (the 8-wave form is built for 128-row tiles: chosen above, it keeps them)
        while (nwl != 8 && mbl > 1 && (long long)cdiv(c.M, 32 * mbl) * cdiv(l.N, nwl * 32) < 2LL * num_cu) mbl >>= 1;
        if (force_mb == 1 || force_mb == 2 || force_mb == 4) mbl = force_mb;
        if (nwl == 8 && (mbl != 4 || (c.padd && c.rows_per_group % 128 != 0))) { nwl = 4; }
        while (c.padd && mbl > 1 && c.rows_per_group % (32 * mbl) != 0) mbl >>= 1;
        a.tiles_n = cdiv(l.N, nwl * 32);
        a.tiles_m = cdiv(c.M, 32 * mbl);
        // full 128-row tiles: the 16x16x32-shape kernel, as for the plain layer (SMK_LINEAR_SHAPE=32 keeps the 32x32x16 one)
        static int shape_ln = -1;
        if (shape_ln < 0) { const char *sv = getenv("SMK_LINEAR_SHAPE"); shape_ln = sv ? atoi(sv) : 16; }
        if (shape_ln != 32 && mbl == 4 && (!c.padd || c.rows_per_group % 128 == 0) && (l.K / 64) % 2 == 0 && l.K >= 128)
            return nwl == 8 ? launch_b16<8, 0, 2, true>(a, st) : launch_b16<4, 0, 2, true>(a, st);
        if (nwl == 8) return launch_mb<4, 8, false, 1, LN_RING, true>(a, st);
        if (mbl == 4) return launch_mb<4, 4, false, 1, LN_RING, true>(a, st);
        if (mbl == 2) return launch_mb<2, 4, false, 1, LN_RING, true>(a, st);
        // one 32-row tile per workgroup (a single frame): the deep weight ring when K allows it
        const bool one = (long long)a.tiles_m * a.tiles_n <= 2LL * num_cu;
        return (one && (l.K / 64) % 4 == 0) ? launch_mb<1, 4, false, 1, 16, true>(a, st) : launch_mb<1, 4, false, 1, LN_RING, true>(a, st);
    }
    int nw = (l.N >= 256 && (long long)cdiv(c.M, 128) * cdiv(l.N, 256) * c.nseg >= num_cu) ? 8 : 4;
    if (force_nw == 4 || force_nw == 8) nw = force_nw;
    a.tiles_n = cdiv(l.N, nw * 32);
    // row-block count per tile: the largest that still gives every CU its share of workgroups (small M: finer tiles)
    static int want_env = -1;
    // workgroups per CU the tile choice aims at: 1 (measured round 4: batch 4 0.284 -> 0.276 ms per frame, batch 2 0.382 -> 0.375, larger
    // batches unchanged -- 64-row tiles at one workgroup per CU beat 32-row tiles at two); SMK_LINEAR_WANT=2 restores round 3's rule
    if (want_env < 0) { const char *sv = getenv("SMK_LINEAR_WANT"); want_env = sv ? atoi(sv) : 1; }
    const long long want = (nw == 8 ? 1LL : (long long)want_env) * num_cu;
    int mb = 4;
    while (mb > 1 && (long long)cdiv(c.M, 32 * mb) * a.tiles_n * c.nseg < want) mb >>= 1;
    if (force_mb == 1 || force_mb == 2 || force_mb == 4) mb = force_mb;
    if (nw == 8) mb = 4;                                   // the 8-wave form is built for full 128-row tiles only
    while (c.padd && mb > 1 && c.rows_per_group % (32 * mb) != 0) mb >>= 1;
    if (c.padd && c.rows_per_group % (32 * mb) != 0) return hipErrorInvalidValue;   // api.hip checks rows_per_group % 32 == 0
    if (c.padd && nw == 8 && mb != 4) { nw = 4; a.tiles_n = cdiv(l.N, 128); }
    a.tiles_m = cdiv(c.M, 32 * mb);
#ifdef SMK_LN_STAMPS
    static unsigned long long *stamp_buf = nullptr;
    if (!stamp_buf) (void)hipMalloc((void **)&stamp_buf, 8 * 8 * 4096);
    (void)hipMemsetAsync(stamp_buf, 0, 8 * 8 * 4096, st);
    a.stamps = stamp_buf;
#endif
    hipError_t e;
    // full 128-row tiles of fp32 activations: the 16x16x32-shape kernel (SMK_LINEAR_SHAPE=32 keeps the 32x32x16 one for A/B runs)
    static int shape_env = -1;
    if (shape_env < 0) { const char *s = getenv("SMK_LINEAR_SHAPE"); shape_env = s ? atoi(s) : 16; }
    if (shape_env != 32 && mb == 4 && !c.x_split && !c.y_split && c.nseg == 1 && !dbg && (!c.padd || c.rows_per_group % 128 == 0)) {
        static int ring_env = -1;
        if (ring_env < 0) { const char *s = getenv("SMK_LINEAR_RING"); ring_env = s ? atoi(s) : 4; }
        if (ring_env == 4 && (l.K / 64) % 2 == 0 && l.K >= 128) e = nw == 8 ? launch_b16<8, 0, 4>(a, st) : launch_b16<4, 0, 4>(a, st);
        else e = nw == 8 ? launch_b16<8>(a, st) : launch_b16<4>(a, st);
    } else if (c.x_split) {
        if (nw == 8) e = launch_mb<4, 8, true>(a, st);
        else if (mb == 4) e = launch_mb<4, 4, true>(a, st);
        else if (mb == 2) e = launch_mb<2, 4, true>(a, st);
        else e = launch_mb<1, 4, true>(a, st);
    } else {
        static int ks2_env = -1;
        if (ks2_env < 0) { const char *sv = getenv("SMK_LINEAR_KS2"); ks2_env = sv ? atoi(sv) : 1; }
        if (nw == 8) e = launch_mb<4, 8, false>(a, st);
        else if (mb == 4) e = launch_mb<4, 4, false>(a, st);
        // 64-row tiles that leave every CU's second workgroup slot empty (batch 4: 256 tiles): two wave groups per workgroup split the K
        // range instead (8 waves per CU either way; the serial K loop, which is what such a launch waits for, is half as long)
        // (measured at M = 4,096: 2048 -> 512 33.3 -> 30.8 us; 512 -> 512 12.6 -> 13.5 us, where the merge outweighs four chunks less: K >= 2,048 only)
        else if (mb == 2 && ks2_env && (long long)a.tiles_m * a.tiles_n * c.nseg <= num_cu && (l.K / 64) % 2 == 0 && l.K >= 2048)
            e = launch_mb<2, 4, false, 2>(a, st);
        else if (mb == 2) e = launch_mb<2, 4, false>(a, st);
        else {
            // few 32 x 128 tiles (batch 1: 128 for the 2048 -> 512 layer): split K over 2 or 4 wave groups per workgroup
            static int force_ks = -1;
            if (force_ks < 0) { const char *sv = getenv("SMK_LINEAR_KS"); force_ks = sv ? atoi(sv) : 0; }
            const long long tiles = (long long)a.tiles_m * a.tiles_n;
            const int nch = l.K / 64;
            int ks = 1;
            // measured at M = 1024: 2048 -> 512 (128 tiles, 32 chunks) 24.4 -> 19.3 us with 4 groups; neutral at 8 chunks; with
            // 512 tiles already on the chip a split only adds the merge (+15 %)
            if (tiles * c.nseg <= num_cu / 2 && nch % 4 == 0 && nch >= 16) ks = 4;
            if (force_ks == 1 || ((force_ks == 2 || force_ks == 4) && nch % force_ks == 0 && nch / force_ks >= 1)) ks = force_ks;
            // one tile per workgroup (a single frame's layers): the deep weight ring (see k_linear_x3) when K allows it
            static int deep = -1;
            if (deep < 0) { const char *sv = getenv("SMK_LINEAR_DEEP"); deep = sv ? atoi(sv) : 1; }
            const bool dr = deep && (nch / ks) % 4 == 0 && tiles * c.nseg <= 2 * num_cu;
            // (not with four wave groups: 1,024 threads leave 128 registers per wave, and the 16-slot ring then spills)
            if (ks == 4) e = launch_mb<1, 4, false, 4>(a, st);
            else if (ks == 2) e = launch_mb<1, 4, false, 2>(a, st);
            else e = dr ? launch_mb<1, 4, false, 1, 16>(a, st) : launch_mb<1, 4, false>(a, st);
        }
    }
#ifdef SMK_LN_STAMPS
    if (getenv("SMK_LN_STAMPS_PRINT")) {   // diagnostic: wait, print the per-wave averages (cycles per tile; clock = core cycles / 100 MHz ticks)
        static unsigned long long h[8 * 4096];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, a.stamps, sizeof(h), hipMemcpyDeviceToHost);
        double k = 0, ep = 0, tot = 0, real = 0, nt = 0, us[5] = {0, 0, 0, 0, 0}; int n = 0;
        for (int w = 0; w < 4096; ++w) if (h[w * 8 + 4]) { ++n; k += h[w*8]; ep += h[w*8+1]; tot += h[w*8+2]; real += h[w*8+3]; nt += h[w*8+4];
            us[0] += h[w*8+5] & 0xffffffffULL; us[1] += h[w*8+5] >> 32; us[2] += h[w*8+6] & 0xffffffffULL; us[3] += h[w*8+6] >> 32; us[4] += h[w*8+7]; }
        const double nch = nt * (l.K / 64);
        if (n) fprintf(stderr, "LN_KSTEPS per chunk: k0 %.0f k1 %.0f k2 %.0f k3(incl barrier) %.0f barrier %.0f\n", us[0] / nch, us[1] / nch, us[2] / nch, us[3] / nch, us[4] / nch);
        if (n) fprintf(stderr, "LN_STAMPS M=%d K=%d N=%d mb=%d nw=%d split=%d waves=%d tiles/wave=%.1f | per tile: kloop %.0f epilogue %.0f | wave total %.0f cyc = %.1f us, clock %.0f MHz\n",
                       c.M, l.K, l.N, mb, nw, c.x_split, n, nt / n, k / nt, ep / nt, tot / n, real / n / 100.0, tot / real * 100.0);
    }
#endif
    return e;
}

// ---- weight gradient of a linear layer: dW [out][in] = dY^T X, a reduction over the token rows (autograd's LinearBackward0, third
// GEMM).  On the layer kernel it is "x' = dY^T [out][rows], w' = X^T [in][rows]": dY is transposed (zero-padded to whole segments) and
// X is split straight into the weight layout with its k index = the token row; the row range is cut into nseg K-segments that run as
// one launch (LinearCall::nseg) and the nseg partial [out][in] slabs are added in segment order (deterministic, no atomics).
// colpart (may be NULL): [rows_pad / 32][cols] partial column sums of the source over this block's 32 rows -- the bias gradient's first
// stage, taken while the tile sits in LDS (k_col_finish adds the row blocks in a fixed order)
__global__ __launch_bounds__(256) void k_transpose_pad(const float *__restrict__ src, long long ld, int rows, int cols, float *__restrict__ dst,
                                                       int rows_pad, float *__restrict__ colpart) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                 // 32 x 8
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        tile[ty + 8 * j][tx] = (r < rows && c < cols) ? src[(long long)r * ld + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + 8 * j, r = r0 + tx;
        if (c < cols && r < rows_pad) dst[(long long)c * rows_pad + r] = tile[tx][ty + 8 * j];
    }
    if (colpart && ty == 0 && c0 + tx < cols) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += tile[k][tx];
        colpart[(size_t)blockIdx.x * cols + c0 + tx] = s;
    }
}

// db[c] = sum over the row blocks of colpart[.][c]: 16 columns x 16 lanes per workgroup, lane kl adds blocks kl, kl + 16, ...; then lane order
__global__ __launch_bounds__(256) void k_col_finish(const float *__restrict__ colpart, int nblocks, int cols, float *__restrict__ db) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float acc = 0.f;
    if (c < cols)
        for (int k = kl; k < nblocks; k += 16) acc += colpart[(size_t)k * cols + c];
    red[kl][cl] = acc;
    __syncthreads();
    if (kl == 0 && c < cols) {
        float s = red[0][cl];
#pragma unroll
        for (int j = 1; j < 16; ++j) s += red[j][cl];
        db[c] = s;
    }
}

__global__ __launch_bounds__(256) void k_sum_segments(const float *__restrict__ part, int nseg, long long n4, float *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 acc = reinterpret_cast<const float4 *>(part)[i];
        for (int s = 1; s < nseg; ++s) {
            const float4 v = reinterpret_cast<const float4 *>(part)[(long long)s * n4 + i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4 *>(out)[i] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_linear_wgrad_tr: dW[out][in] = sum over the token rows of dY[row][out] * X[row][in] WITHOUT the transposed copy of dY and the
// re-laid-out copy of X (k_transpose_pad + k_split_linear_weights_t16: 6 ms of a 64 ms step).  The token rows are the MFMA k
// dimension, and both operands are feature-contiguous in memory, i.e. k-STRIDED: they are staged row-major as they lie (coalesced
// 512-byte row pieces, split into bf16 hi / lo planes on the way into LDS) and read back TRANSPOSED by gfx950's ds_read_b64_tr_b16
// (a 16-lane group reads a 4-row x 16-column block and lane i receives column i of the four rows: four consecutive k of one feature).
//   workgroup: a 128 x 128 tile of dW for one segment of the rows; wave (wm, wn): 64 x 64 (4 x 4 MFMA tiles, 64 accumulators).
//   chunk:     32 rows = one k-step; LDS image per operand and plane [32 rows][128 features] bf16, row pitch 288 B: with group g reading rows
//              4g .. 4g+3 and 16+4g .. 16+4g+3 (the k order inside a k-step is free as long as both operands share it) the 32 lanes of a
//              half touch 64 distinct banks.  Double buffered: the next chunk's global loads are in flight under this chunk's MFMAs.
//   output:    partial [segment][out][in]; k_sum_segments adds the segments in order; the bias gradient's partial column sums of dY come
//              from the staging registers of the workgroups with in-tile 0 (k_col_finish adds them in order).
constexpr int WT_PITCH = 288;                                 // bytes per staged row (128 bf16 + 32 pad)
constexpr int WT_PLANE = 32 * WT_PITCH;                       // 9,216
constexpr int WT_LDS = 2 * 4 * WT_PLANE;                      // two buffers x (dY hi, dY lo, X hi, X lo) = 73,728

struct WgradTrArgs {
    const float *dy, *x;
    long long ld_dy, ldx, rows, rows_per_seg;
    int out_f, in_f, tiles_n, ntiles, nseg;
    float *part, *dbpart;
};

__global__ __launch_bounds__(256, 2) void k_linear_wgrad_tr(WgradTrArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave & 1, wn = wave >> 1;
    // blocks i and i + 8 share an XCD: consecutive slots of one XCD take the tiles of ONE segment (they read the same rows of dY and X)
    int seg, tile;
    if ((a.nseg & 7) == 0) {
        const int slot = blockIdx.x >> 3;
        tile = slot % a.ntiles;
        seg = (slot / a.ntiles) * 8 + (blockIdx.x & 7);
    } else {
        tile = blockIdx.x % a.ntiles;
        seg = blockIdx.x / a.ntiles;
    }
    const int tm = tile / a.tiles_n, tn = tile - tm * a.tiles_n;
    const long long row0 = (long long)seg * a.rows_per_seg;
    long long row1 = row0 + a.rows_per_seg;
    row1 = row1 < a.rows ? row1 : a.rows;
    const float *dyb = a.dy + (size_t)tm * 128, *xb = a.x + (size_t)tn * 128;
    // staging map: thread -> rows (tid >> 5) + 8 j (j = 0..3) of the chunk, features 4 (tid & 31) .. + 3.  Addresses = a wave-uniform chunk base
    // (scalar registers, advanced per chunk on the scalar unit) + a per-thread 32-bit byte offset computed once: no vector address arithmetic
    // in the loop (it was ~70 of the loop's ~300 vector instructions).
    const int srow = tid >> 5, sq = tid & 31;
    unsigned voy[4], vox[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        voy[j] = (unsigned)(((size_t)(srow + 8 * j) * a.ld_dy + 4 * sq) * sizeof(float));
        vox[j] = (unsigned)(((size_t)(srow + 8 * j) * a.ldx + 4 * sq) * sizeof(float));
    }
    float4 ry[4], rx[4];
    float dbs[4] = {0.f, 0.f, 0.f, 0.f};
    // FULL chunks (all 32 rows inside the segment) load and split unconditionally; only a segment's last, partial chunk clamps its addresses and
    // zeroes the rows past the end (at the split: overwriting a loaded register under a condition makes the compiler wait for that load on the spot)
    auto fetch = [&](long long r0, bool partial) {
        const unsigned char *cy = reinterpret_cast<const unsigned char *>(dyb) + (size_t)r0 * a.ld_dy * sizeof(float);
        const unsigned char *cx = reinterpret_cast<const unsigned char *>(xb) + (size_t)r0 * a.ldx * sizeof(float);
        if (!partial) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ry[j] = *reinterpret_cast<const float4 *>(cy + voy[j]);
                rx[j] = *reinterpret_cast<const float4 *>(cx + vox[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = r0 + srow + 8 * j < row1;
                ry[j] = *reinterpret_cast<const float4 *>(cy + (ok ? voy[j] : voy[0]));      // (row r0 itself is inside)
                rx[j] = *reinterpret_cast<const float4 *>(cx + (ok ? vox[j] : vox[0]));
            }
        }
    };
    auto stash = [&](int buf, long long r0, bool partial) {
        unsigned char *base = smem + buf * 4 * WT_PLANE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = (srow + 8 * j) * WT_PITCH + 8 * sq;
            const float keep = (!partial || (r0 + srow + 8 * j) < row1) ? 1.f : 0.f;
            float fy[4] = {ry[j].x, ry[j].y, ry[j].z, ry[j].w}, fx[4] = {rx[j].x, rx[j].y, rx[j].z, rx[j].w};
            if (partial) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { fy[c] *= keep; fx[c] *= keep; }
            }
            bf16x4 yh, yl, xh, xl;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const __bf16 h1 = (__bf16)fy[c], h2 = (__bf16)fx[c];
                yh[c] = h1; yl[c] = (__bf16)(fy[c] - (float)h1);
                xh[c] = h2; xl[c] = (__bf16)(fx[c] - (float)h2);
                dbs[c] += fy[c];
            }
            *reinterpret_cast<bf16x4 *>(base + off) = yh;
            *reinterpret_cast<bf16x4 *>(base + WT_PLANE + off) = yl;
            *reinterpret_cast<bf16x4 *>(base + 2 * WT_PLANE + off) = xh;
            *reinterpret_cast<bf16x4 *>(base + 3 * WT_PLANE + off) = xl;
        }
    };
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    f32x4v acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[m][n][g] = 0.f;
    // transposed-read addresses: 16-lane group g = lane >> 4; lane 4q + p of it supplies row (4g + q [+ 16]), features 16 t + 4p .. + 3
    const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int tr_off = (4 * g16 + q) * WT_PITCH + 8 * pp;
    auto frag = [&](const unsigned char *plane, int col0) -> bf16x8 {      // 8 k of feature col0 + (lane & 15): rows 4g..4g+3, 16+4g..16+4g+3
        const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3))) *)(plane + tr_off + 2 * col0));
        const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3))) *)(plane + tr_off + 16 * WT_PITCH + 2 * col0));
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) { r[e] = lo4[e]; r[4 + e] = hi4[e]; }
        return r;
    };
    const long long nchunks = (row1 - row0 + 31) / 32, nfull = (row1 - row0) / 32;      // nchunks - nfull = 0 or 1
    if (nchunks > 0) {
        fetch(row0, nfull == 0);
        stash(0, row0, nfull == 0);
    }
    __syncthreads();
    for (long long c = 0; c < nchunks; ++c) {
        const int buf = (int)(c & 1);
        const bool next_partial = c + 1 >= nfull;             // wave-uniform
        if (c + 1 < nchunks) {
            if (next_partial) fetch(row0 + 32 * (c + 1), true);
            else fetch(row0 + 32 * (c + 1), false);           // in flight under this chunk's MFMAs
        }
        const unsigned char *base = smem + buf * 4 * WT_PLANE;
        bf16x8 ah[4], al[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            ah[m] = frag(base, 64 * wm + 16 * m);
            al[m] = frag(base + WT_PLANE, 64 * wm + 16 * m);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const bf16x8 bh = frag(base + 2 * WT_PLANE, 64 * wn + 16 * n), bl = frag(base + 3 * WT_PLANE, 64 * wn + 16 * n);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4v &cc = acc[m][n];
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bh, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bl, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bh, cc, 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) {
            if (next_partial) stash(buf ^ 1, row0 + 32 * (c + 1), true);
            else stash(buf ^ 1, row0 + 32 * (c + 1), false);
        }
        __syncthreads();
    }
    // partial [seg][out][in]: lane (n15 = lane & 15, kg = lane >> 4) of tile (m, n) holds rows 4kg + r (out) of column n15 (in)
    float *dst = a.part + ((size_t)seg * a.out_f + (size_t)tm * 128 + 64 * wm) * a.in_f + (size_t)tn * 128 + 64 * wn;
    const int n15 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(size_t)(16 * m + 4 * kg + r) * a.in_f + 16 * n + n15] = acc[m][n][r];
    if (a.dbpart && tn == 0) {                                // column sums of this segment's dY rows: 8 staging threads per feature quad
        float *ex = reinterpret_cast<float *>(smem);           // [8][128]
#pragma unroll
        for (int c = 0; c < 4; ++c) ex[srow * 128 + 4 * sq + c] = dbs[c];
        __syncthreads();
        if (tid < 128) {
            float sdb = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) sdb += ex[k * 128 + tid];
            a.dbpart[(size_t)seg * a.out_f + (size_t)tm * 128 + tid] = sdb;
        }
    }
}

struct WgradTrPlan { bool ok; int nseg, tiles_m, tiles_n; long long rows_per_seg; size_t bytes; };
static WgradTrPlan plan_wgrad_tr(long long rows, int out_f, int in_f, long long ld_dy, long long ldx) {
    WgradTrPlan p{};
    static const bool off = [] { const char *e = getenv("SMK_LINEAR_WGRAD_TR"); return e && e[0] == '0'; }();
    p.ok = !off && out_f % 128 == 0 && in_f % 128 == 0 && ld_dy % 4 == 0 && ldx % 4 == 0 && rows >= 32;
    if (!p.ok) return p;
    p.tiles_m = out_f / 128; p.tiles_n = in_f / 128;
    const int tiles = p.tiles_m * p.tiles_n;
    int nseg = 2 * device_num_cu() / tiles;
    if (nseg >= 8) nseg &= ~7;
    const long long max_seg = rows / 256 > 0 ? rows / 256 : 1;      // segments of at least 256 rows
    if (nseg > max_seg) nseg = (int)max_seg;
    if (nseg < 1) nseg = 1;
    if (nseg >= 8) nseg &= ~7;
    p.nseg = nseg;
    p.rows_per_seg = ((rows + nseg - 1) / nseg + 31) / 32 * 32;
    p.bytes = ((size_t)nseg * out_f * in_f + (size_t)nseg * out_f) * sizeof(float);
    return p;
}

WgradPlan plan_linear_wgrad(long long rows, int out_f, int in_f) {
    WgradPlan p;
    const long long tiles = (long long)cdiv(out_f, 128) * cdiv(in_f, 128);
    int nseg = 1;
    while (nseg < 64 && tiles * nseg < 512 && rows / (2 * nseg) >= 1024) nseg *= 2;     // fill ~512 workgroup slots; segments of >= 1024 rows
    p.nseg = nseg;
    const long long unit = 64LL * nseg;
    p.rows_pad = (rows + unit - 1) / unit * unit;
    p.off_wq = (size_t)out_f * p.rows_pad * sizeof(float);                                // after dY^T
    p.off_part = p.off_wq + (size_t)p.rows_pad * in_f * 2 * sizeof(unsigned short);
    p.off_col = p.off_part + (nseg > 1 ? (size_t)nseg * out_f * in_f * sizeof(float) : 0);      // bias-gradient partials [rows_pad / 32][out]
    p.bytes = p.off_col + (size_t)(p.rows_pad / 32) * out_f * sizeof(float);
    const WgradTrPlan t = plan_wgrad_tr(rows, out_f, in_f, 4, 4);      // the transposed-read form needs only its partial sums
    if (t.ok && t.bytes > p.bytes) p.bytes = t.bytes;
    return p;
}

hipError_t launch_linear_wgrad(const float *dy, long long ld_dy, const float *x, long long ldx, long long rows, int out_f, int in_f,
                               float *dw, float *db, void *workspace, hipStream_t st) {
    const WgradTrPlan t = plan_wgrad_tr(rows, out_f, in_f, ld_dy, ldx);
    if (t.ok && (reinterpret_cast<size_t>(dy) & 15) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0) {
        WgradTrArgs a;
        a.dy = dy; a.x = x; a.ld_dy = ld_dy; a.ldx = ldx; a.rows = rows; a.rows_per_seg = t.rows_per_seg;
        a.out_f = out_f; a.in_f = in_f; a.tiles_n = t.tiles_n; a.ntiles = t.tiles_m * t.tiles_n; a.nseg = t.nseg;
        a.part = t.nseg > 1 ? reinterpret_cast<float *>(workspace) : dw;
        a.dbpart = db ? reinterpret_cast<float *>(workspace) + (size_t)t.nseg * out_f * in_f : nullptr;
        once_per_device((const void *)k_linear_wgrad_tr, [&] {
            (void)hipFuncSetAttribute((const void *)k_linear_wgrad_tr, hipFuncAttributeMaxDynamicSharedMemorySize, WT_LDS);
        });
        hipLaunchKernelGGL(k_linear_wgrad_tr, dim3(a.ntiles * t.nseg), dim3(256), WT_LDS, st, a);
        if (t.nseg > 1) {
            const long long n4 = (long long)out_f * in_f / 4;
            const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
            hipLaunchKernelGGL(k_sum_segments, dim3(blocks), dim3(256), 0, st, a.part, t.nseg, n4, dw);
        }
        if (db) hipLaunchKernelGGL(k_col_finish, dim3(cdiv(out_f, 16)), dim3(256), 0, st, a.dbpart, t.nseg, out_f, db);
        return hipGetLastError();
    }
    const WgradPlan p = plan_linear_wgrad(rows, out_f, in_f);
    float *dyt = reinterpret_cast<float *>(workspace);
    LinearDev l;
    l.wq = reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(workspace) + p.off_wq);
    l.bias = nullptr;
    l.N = in_f;
    l.K = (int)p.rows_pad;
    float *part = p.nseg > 1 ? reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(workspace) + p.off_part) : dw;
    float *colpart = db ? reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(workspace) + p.off_col) : nullptr;
    hipLaunchKernelGGL(k_transpose_pad, dim3((unsigned)(p.rows_pad / 32), cdiv(out_f, 32)), dim3(256), 0, st, dy, ld_dy, (int)rows, out_f, dyt,
                       (int)p.rows_pad, colpart);
    if (db) hipLaunchKernelGGL(k_col_finish, dim3(cdiv(out_f, 16)), dim3(256), 0, st, colpart, (int)(p.rows_pad / 32), out_f, db);   // the bias gradient
    hipError_t e = launch_split_linear_weights(x, nullptr, l, st, 1, ldx, (int)rows);
    if (e != hipSuccess) return e;
    l.K = (int)(p.rows_pad / p.nseg);                       // segment length
    LinearCall c;
    c.x = dyt; c.ldx = p.rows_pad;
    c.y = part; c.ldy = in_f;
    c.res = nullptr; c.ldr = 0; c.padd = nullptr; c.rows_per_group = 1; c.period = 1;
    c.M = out_f; c.act = 0; c.x_split = 0; c.y_split = 0;
    c.nseg = p.nseg;
    e = launch_linear_x3(l, c, st);
    if (e != hipSuccess) return e;
    if (p.nseg > 1) {
        const long long n4 = (long long)out_f * in_f / 4;
        const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_sum_segments, dim3(blocks), dim3(256), 0, st, part, p.nseg, n4, dw);
    }
    return hipGetLastError();
}

}  // namespace smk
