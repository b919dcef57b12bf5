// Small kernels of SmokePhysNet's transformer body that are not GEMMs (the GEMMs: linear.hip).
#include "transformer.h"

namespace smk {

// The chaos term of one ChaosAttention layer, folded into Q (chaos_attention.py:39-66 lorenz_system / generate_chaos_field,
// :85-100 chaos_proj, chaos_gate): per batch element three N(0,1) draws * 0.1 seed five explicit-Euler Lorenz steps; each
// state s_t gives C_t = chaos_proj(s_t) [D], g_t = sigmoid(chaos_gate(C_t)) and the addend strength * g_t * C_t that row
// l = t (mod 5) of the sequence adds to its query.  In the reference this is ~90 one-element-per-batch elementwise
// launches per layer; here one workgroup per batch element.  fp32 with the reference's operation order (no contraction).
__device__ __forceinline__ void chaos_addend_body(const ChaosAddendArgs &a, const int b) {
    __shared__ float red[5][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
            a.addend[((size_t)b * 5 + t) * a.ld + d] = g[t] * (((sx[t] * w0 + sy[t] * w1) + sz[t] * w2) + pb);
    }
}

__global__ __launch_bounds__(256) void k_chaos_addend(const ChaosAddendArgs a) { chaos_addend_body(a, blockIdx.x); }
// every layer's addend in one launch (the layers' noise is drawn up front: smokephys_net.py, _body_hip): blockIdx.y = layer
__global__ __launch_bounds__(256) void k_chaos_addend_batch(const ChaosAddendBatch args) { chaos_addend_body(args.layer[blockIdx.y], blockIdx.x); }

// ---- the model's tail: token mean + physics head (see transformer.h)
constexpr int PH_CHUNKS = 32;
__global__ __launch_bounds__(256) void k_token_chunk_sums(const PooledHeadArgs a) {
    const int b = blockIdx.x, ch = blockIdx.y;
    const int per = (a.L + PH_CHUNKS - 1) / PH_CHUNKS, l0 = ch * per, l1 = l0 + per < a.L ? l0 + per : a.L;
    for (int d = threadIdx.x; d < a.D; d += 256) {
        float s = 0.f;
#pragma unroll 8
        for (int l = l0; l < l1; ++l) s += a.x[((size_t)b * a.L + l) * a.ldx + d];
        a.ws[((size_t)b * PH_CHUNKS + ch) * a.D + d] = s;
    }
}
// hidden layer, 32 units per workgroup (grid (B, ceil(H1 / 32)): one workgroup for the whole 512 KB matrix is a chain of load -> reduce
// round trips, 33 us at batch 1): every workgroup forms the pooled vector from the chunk sums (in chunk order), workgroup y = 0 stores it;
// a wave takes 8 units at once, lanes along the input (coalesced 16-byte weight reads); hidden -> ws behind the chunk sums
__global__ __launch_bounds__(256) void k_pooled_hidden(const PooledHeadArgs a) {
    extern __shared__ float ph_smem[];                       // pooled [D]
    float *pooled = ph_smem;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int d = tid; d < a.D; d += 256) {
        float s = 0.f;
#pragma unroll 8
        for (int ch = 0; ch < PH_CHUNKS; ++ch) s += a.ws[((size_t)b * PH_CHUNKS + ch) * a.D + d];
        s = s / (float)a.L;
        pooled[d] = s;
        if (blockIdx.y == 0) a.pooled[(size_t)b * a.D + d] = s;
    }
    __syncthreads();
    float *hid = a.ws + (size_t)a.B * PH_CHUNKS * a.D + (size_t)b * a.H1;
    const int dq = a.D >> 2;                                 // float4 per row (D % 4 == 0: api.hip)
    const int j0 = blockIdx.y * 32 + wave * 8;
    if (j0 >= a.H1) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = lane; q < dq; q += 64) {
        const float4 pv = *reinterpret_cast<const float4 *>(pooled + 4 * q);
        float4 wv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int j = j0 + r < a.H1 ? j0 + r : a.H1 - 1;
            wv[r] = *reinterpret_cast<const float4 *>(a.w1 + (size_t)j * a.D + 4 * q);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) s[r] += ((wv[r].x * pv.x + wv[r].y * pv.y) + wv[r].z * pv.z) + wv[r].w * pv.w;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float v = s[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0 && j0 + r < a.H1) {
            v += a.b1[j0 + r];
            hid[j0 + r] = v > 0.f ? v : 0.f;
        }
    }
}
// output layer: a wave per output unit over the hidden vector
__global__ __launch_bounds__(256) void k_pooled_out(const PooledHeadArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *hid = a.ws + (size_t)a.B * PH_CHUNKS * a.D + (size_t)b * a.H1;
    for (int k = wave; k < a.H2; k += 4) {
        float s = 0.f;
        for (int j = lane; j < a.H1; j += 64) s += a.w2[(size_t)k * a.H1 + j] * hid[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) a.out[(size_t)b * a.H2 + k] = s + a.b2[k];
    }
}
hipError_t launch_pooled_head(const PooledHeadArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(k_token_chunk_sums, dim3(a.B, PH_CHUNKS), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_pooled_hidden, dim3(a.B, (a.H1 + 31) / 32), dim3(256), (size_t)a.D * sizeof(float), st, a);
    hipLaunchKernelGGL(k_pooled_out, dim3(a.B), dim3(256), 0, st, a);
    return hipGetLastError();
}

// The five Lorenz states alone ([B][5][3]; chaos_attention.py:39-59): the part of the chaos term that carries no gradient, for the
// training path (chaos_proj / chaos_gate then run under autograd on a [B, 5, 3] tensor instead of behind ~75 one-element launches).
__global__ __launch_bounds__(64) void k_lorenz_states(const float *__restrict__ noise, int B, float sigma, float rho, float beta, float dt,
                                                     float *__restrict__ states) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    float x = noise[b] * 0.1f, y = noise[B + b] * 0.1f, z = noise[2 * B + b] * 0.1f;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const float dx = sigma * (y - x);
        const float dy = x * (rho - z) - y;
        const float dz = x * y - beta * z;
        x = x + dt * dx;
        y = y + dt * dy;
        z = z + dt * dz;
        float *o = states + ((size_t)b * 5 + t) * 3;
        o[0] = x; o[1] = y; o[2] = z;
    }
}

hipError_t launch_lorenz_states(const float *noise, int B, float sigma, float rho, float beta, float dt, float *states, hipStream_t st) {
    hipLaunchKernelGGL(k_lorenz_states, dim3((B + 63) / 64), dim3(64), 0, st, noise, B, sigma, rho, beta, dt, states);
    return hipGetLastError();
}

hipError_t launch_chaos_addend(const ChaosAddendArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(k_chaos_addend, dim3(a.B), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_chaos_addend_batch(const ChaosAddendBatch &a, hipStream_t st) {
    hipLaunchKernelGGL(k_chaos_addend_batch, dim3(a.layer[0].B, a.NL), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Softmax attention of ChaosAttention (chaos_attention.py:102-112: scores / sqrt(d) / temperature, softmax over keys, @ V,
// heads merged back to [B, L, D]) with the chaos term already folded into Q -- flash style (no L x L score tensor), head
// dim 64, fp32 accuracy on the bf16 matrix cores through split operands (x = hi + lo; hi*hi + hi*lo + lo*hi).
//
// Workgroup = 256 threads = one (batch, head, 128-query block); wave w owns queries 32w..32w+31 with Q (pre-scaled by
// scale*log2 e, split) in registers.  K/V tiles of 64 keys are split on the fly and staged through LDS, double buffered:
// K row-major [key][d] (pitch 144 B), V row-major too ([key][d], unpadded 128-byte rows whose 64-byte halves are swapped on keys with
// bit 1 set) and read back TRANSPOSED by ds_read_b64_tr_b16 -- both images are written with conflict-free 8-byte row stores (round 3's
// [d][key] image of V cost a 4-way bank conflict on every store, 27 % of the kernel's LDS cycles, and 4 x 4 register transposes).
//   S^T = mfma(K, Q): the query is the lane (column), 16 keys per 32-key block sit in the lane's accumulator registers, the
//         other 16 in lane ^ 32 -> row max / sum are lane-local plus one cross-half exchange; alpha is a per-lane scalar.
//   O^T = mfma(V^T, P): exp2'd accumulator registers 8s..8s+7 ARE the B fragment of k-step s (cdna_hip_programming.md, "An
//         accumulator tile as the next MFMA's operand"); their fixed k-permutation -- element j of lane half h is key
//         16s + 8(j>>2) + 4h + (j&3) -- is matched on the A side by two transposed reads of V, at keys 16s+4h..+3 and 16s+8+4h..+3.
// O^T leaves each lane with 4 consecutive d of its query per register quad -> 16-byte stores into [B][L][H*64].
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#if defined(SMK_ATT_ABLATE) && (SMK_ATT_ABLATE & 4)      // diagnostic: the MFMAs replaced by one dependent vector op each (operands stay live)
#define AT_MFMA(A, B, C) ([&] { auto c_ = (C); c_[0] += __builtin_bit_cast(f32x4_t, (A))[0] * __builtin_bit_cast(f32x4_t, (B))[0]; return c_; }())
#else
#define AT_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16((A), (B), (C), 0, 0, 0)
#endif

constexpr int AT_QB = 128, AT_KV = 64;
constexpr int AT_KPITCH = 144, AT_VPITCH = 128;
constexpr int AT_KPLANE = 64 * AT_KPITCH, AT_VPLANE = 64 * AT_VPITCH;          // 9216, 8192
constexpr int AT_BUF = 2 * AT_KPLANE + 2 * AT_VPLANE;                          // 34,816 B per tile (K hi|lo, V hi|lo)
constexpr int AT_LDS = 2 * AT_BUF;                                             // 69,632 B -> 2 workgroups per CU

// presplit: the 16 bytes already hold {hi[0..3], lo[0..3]} (the q | k | v layer's epilogue wrote k and v that way: AttnArgs::kv_split)
__device__ __forceinline__ void at_split4(const float4 &v, bf16x4 &h, bf16x4 &l, bool presplit = false) {
#if defined(SMK_ATT_ABLATE) && (SMK_ATT_ABLATE & 2)      // diagnostic build (tools/att_ablate.sh): no split arithmetic, same LDS stores
    presplit = true;
#endif
    if (presplit) {
        h = __builtin_bit_cast(bf16x4, make_float2(v.x, v.y)); l = __builtin_bit_cast(bf16x4, make_float2(v.z, v.w));
        return;
    }
    const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 t = (__bf16)f[i];
        h[i] = t;
        l[i] = (__bf16)(f[i] - (float)t);
    }
}

// KS = 2 (small grids: at most one workgroup per CU): 512 threads; wave group wave >> 2 takes half of the key tiles for the
// same 128 queries, with its own LDS double buffer; the two partial (m, l, O) are merged through LDS at the end.
// SB (KS = 1 only): one LDS tile buffer instead of two (35,840 B): three workgroups per CU = 3 waves per SIMD (168 registers), two
// barriers per tile; the next tile's global loads are still in flight during the MFMAs.  With 2 waves per SIMD the S -> softmax -> PV
// chain of a wave is exposed (SQ counters: matrix pipe 43 % busy, VALU issue 49 %); a third wave fills part of it.
// KVS: k and v arrive as in-place split-bf16 (AttnArgs::kv_split): the staging moves the two halves of a 16-byte piece as they are
template <int KS, bool SB = false, bool KVS = false>
__global__ __launch_bounds__(256 * KS, SB ? 3 : 2) void k_attention_x3(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int grp = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    unsigned char *smem = smem_all + grp * AT_LDS;
    const int tid = threadIdx.x & 255, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the L/128 query blocks of one (batch, head) read the same K/V: give them consecutive ids on ONE XCD (ids are dealt to the
    // 8 XCDs round-robin; needs gridDim.x % 8 == 0, else the plain order)
    const int vid = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int nqb = a.L / AT_QB;
    const int split = vid % a.nsplit, vq = vid / a.nsplit;   // (nsplit > 1: the splits of one query block are neighbours on one XCD)
    const int qb = vq % nqb, bh = vq / nqb, head = bh % a.H, b = bh / a.H;
    const int NT = a.L / AT_KV / KS / a.nsplit, T0 = (split * KS + grp) * NT;          // this wave group's key tiles: T0 .. T0 + NT - 1

    // ---- Q fragments (B operand: column = query r, k = d 16s + 8hh + j), pre-scaled, split
    bf16x8 qh[4], ql[4];
    {
        const float *qp = a.q + ((size_t)b * a.L + qb * AT_QB + wave * 32 + r) * a.ldq + head * 64 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 v0 = *reinterpret_cast<const float4 *>(qp + 16 * s), v1 = *reinterpret_cast<const float4 *>(qp + 16 * s + 4);
            const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = f[j] * a.scale_log2e;
                const __bf16 t = (__bf16)x;
                qh[s][j] = t;
                ql[s][j] = (__bf16)(x - (float)t);
            }
        }
    }

    // ---- K / V staging through buffer resources (one v_add per load; offsets < 2^31: api.hip)
    const int c4 = tid & 15, rq = tid >> 4;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.k), 0, (int)((size_t)a.B * a.L * a.ldk * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.v), 0, (int)((size_t)a.B * a.L * a.ldv * 4), 0x00020000);
    const int lane_k = (rq * a.ldk + head * 64 + c4 * 4) * 4;                  // K: rows rq + 16j of the tile
    const int lane_v = (rq * a.ldv + head * 64 + c4 * 4) * 4;                  // V: rows rq + 16j, as K
    float4 kst[4], vst[4];
    auto stage_load = [&](int t) {
        const int tt = T0 + (t < NT ? t : NT - 1);                             // past the end: a harmless re-read
        const unsigned row0 = (unsigned)b * a.L + tt * AT_KV;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 kv = __builtin_amdgcn_raw_buffer_load_b128(krs, (int)((row0 + 16 * j) * (unsigned)a.ldk * 4u) + lane_k, 0, 0);
            kst[j] = make_float4(__uint_as_float(kv.x), __uint_as_float(kv.y), __uint_as_float(kv.z), __uint_as_float(kv.w));
            const u32x4 vv = __builtin_amdgcn_raw_buffer_load_b128(vrs, (int)((row0 + 16 * j) * (unsigned)a.ldv * 4u) + lane_v, 0, 0);
            vst[j] = make_float4(__uint_as_float(vv.x), __uint_as_float(vv.y), __uint_as_float(vv.z), __uint_as_float(vv.w));
        }
    };
    constexpr bool kvs = KVS;
    auto stage_store = [&](int buf) {
        unsigned char *base = smem + buf * AT_BUF;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                         // K[key rq + 16j][d 4c4..]
            bf16x4 h, l;
            at_split4(kst[j], h, l, kvs);
            unsigned char *p = base + (rq + 16 * j) * AT_KPITCH + c4 * 8;
            *reinterpret_cast<bf16x4 *>(p) = h;
            *reinterpret_cast<bf16x4 *>(p + AT_KPLANE) = l;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                         // V[key rq + 16j][d 4c4..] (64-byte halves swapped on keys with bit 1)
            bf16x4 h, l;
            at_split4(vst[j], h, l, kvs);
            unsigned char *p = base + 2 * AT_KPLANE + (rq + 16 * j) * AT_VPITCH + ((c4 * 8) ^ (((rq >> 1) & 1) << 6));
            *reinterpret_cast<bf16x4 *>(p) = h;
            *reinterpret_cast<bf16x4 *>(p + AT_VPLANE) = l;
        }
    };

    // V fragments by transposed reads (ds_read_b64_tr_b16): 16-lane group g16 = lane >> 4 = (d half g16 & 1, key half hh); lane 4q + pp of it
    // supplies the address of key row 4hh + q, d 16 (g16 & 1) + 4pp .. + 3, and receives d (lane & 15) of the group's four key rows.
    // Rows q = 0 .. 3 of one read lie 128 B apart with the halves of rows 2, 3 swapped: four different 64-byte bank segments.
    const int vrow = (lane >> 2) & 3, vtr_swz = (vrow >> 1) << 6;
    const int vtr_off = (4 * hh + vrow) * AT_VPITCH + (((lane >> 4) & 1) << 5) + ((lane & 3) << 3);
    auto at_tr = [](const unsigned char *p) -> bf16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3))) *)p);
    };

    stage_load(0);
    if (!SB) {
        stage_store(0);
        stage_load(1);
        __syncthreads();
    }

    f32x16 oacc[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 16; ++g) oacc[db][g] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
        if (SB) {                                              // tile t -> the single buffer (every read of tile t-1 ended at the barrier below)
            stage_store(0);
            stage_load(t + 1);
            __syncthreads();
        }
        const unsigned char *kb_h = smem + (SB ? 0 : (t & 1)) * AT_BUF + r * AT_KPITCH + hh * 16;
        const unsigned char *vt_h = smem + (SB ? 0 : (t & 1)) * AT_BUF + 2 * AT_KPLANE + vtr_off;
        // ---- S^T = K Q^T (log2 units): sacc[kb][g] = score(query r, key 32kb + (g&3) + 8(g>>2) + 4hh)
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int g = 0; g < 16; ++g) sacc[kb][g] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 kh[2], kl[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                kh[kb] = *reinterpret_cast<const bf16x8 *>(kb_h + kb * 32 * AT_KPITCH + s * 32);
                kl[kb] = *reinterpret_cast<const bf16x8 *>(kb_h + AT_KPLANE + kb * 32 * AT_KPITCH + s * 32);
            }
            // product-major: consecutive MFMAs alternate between the two accumulators (each still sums lo*hi, hi*lo, hi*hi in order)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) sacc[kb] = AT_MFMA(kl[kb], qh[s], sacc[kb]);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) sacc[kb] = AT_MFMA(kh[kb], ql[s], sacc[kb]);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) sacc[kb] = AT_MFMA(kh[kb], qh[s], sacc[kb]);
        }
        // the staged registers hold tile t+1: split + write it to the other buffer (its last reads ended before the barrier
        // that closed tile t-1), then re-issue the loads for tile t+2
        if (!SB) {
            if (t + 1 < NT) stage_store((t + 1) & 1);
            stage_load(t + 2);
        }

#if defined(SMK_ATT_ABLATE) && (SMK_ATT_ABLATE & 1)      // diagnostic: no softmax arithmetic (P = the scores' bits), same MFMAs and LDS traffic
        bf16x8 ph[4], pl[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            ph[s4] = __builtin_bit_cast(bf16x8, f32x4_t{sacc[s4 >> 1][8 * (s4 & 1)], sacc[s4 >> 1][8 * (s4 & 1) + 1], sacc[s4 >> 1][8 * (s4 & 1) + 2], sacc[s4 >> 1][8 * (s4 & 1) + 3]});
            pl[s4] = __builtin_bit_cast(bf16x8, f32x4_t{sacc[s4 >> 1][8 * (s4 & 1) + 4], sacc[s4 >> 1][8 * (s4 & 1) + 5], sacc[s4 >> 1][8 * (s4 & 1) + 6], sacc[s4 >> 1][8 * (s4 & 1) + 7]});
        }
        l_run = 1.f;
#else
        // ---- online softmax (per lane: its query's 32 of the tile's 64 keys; lane ^ 32 holds the other 32)
        float mx = sacc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int g = 0; g < 16; ++g) mx = fmaxf(mx, sacc[kb][g]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);             // first tile: exp2(-inf) = 0
        m_run = m_new;
        float psum = 0.f;
        bf16x8 ph[4], pl[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float p = __builtin_amdgcn_exp2f(sacc[s4 >> 1][8 * (s4 & 1) + j] - m_new);
                psum += p;
                const __bf16 t16 = (__bf16)p;
                ph[s4][j] = t16;
                pl[s4][j] = (__bf16)(p - (float)t16);
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 16; ++g) oacc[db][g] *= alpha;
#endif

        // ---- O^T += V^T P: oacc[db][g] = O(query r, d 32db + (g&3) + 8(g>>2) + 4hh)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            bf16x8 vh[2], vl[2];
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                // keys 16 s4 + 4hh + q (and + 8), d 32 db + 16 (g16 & 1) + 4pp .. + 3 -> this lane: d 32 db + r, the four keys in order
                const unsigned char *p0 = vt_h + s4 * 16 * AT_VPITCH + ((db << 6) ^ vtr_swz);
                const bf16x4 h0 = at_tr(p0), h1 = at_tr(p0 + 8 * AT_VPITCH);
                const bf16x4 l0 = at_tr(p0 + AT_VPLANE), l1 = at_tr(p0 + AT_VPLANE + 8 * AT_VPITCH);
                vh[db] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                vl[db] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int db = 0; db < 2; ++db) oacc[db] = AT_MFMA(vl[db], ph[s4], oacc[db]);
#pragma unroll
            for (int db = 0; db < 2; ++db) oacc[db] = AT_MFMA(vh[db], pl[s4], oacc[db]);
#pragma unroll
            for (int db = 0; db < 2; ++db) oacc[db] = AT_MFMA(vh[db], ph[s4], oacc[db]);
        }
        __syncthreads();                                       // tile t+1 visible; every read of tile t's buffer has returned
    }

    if (KS == 2) {   // merge the two wave groups' partial softmax states (same lane layout in both)
        float *xch = reinterpret_cast<float *>(smem_all + AT_LDS) + (size_t)tid * 36;   // group 1's buffers are free after the last barrier
        if (grp == 1) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    *reinterpret_cast<float4 *>(xch + 16 * db + 4 * q4) =
                        make_float4(oacc[db][4 * q4], oacc[db][4 * q4 + 1], oacc[db][4 * q4 + 2], oacc[db][4 * q4 + 3]);
            xch[32] = m_run;
            xch[33] = l_run;
        }
        __syncthreads();
        if (grp == 1) return;
        const float m2 = xch[32], l2 = xch[33];
        const float m_new = fmaxf(m_run, m2);
        const float a1 = __builtin_amdgcn_exp2f(m_run - m_new), a2 = __builtin_amdgcn_exp2f(m2 - m_new);
        l_run = l_run * a1 + l2 * a2;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 o2 = *reinterpret_cast<const float4 *>(xch + 16 * db + 4 * q4);
                oacc[db][4 * q4] = oacc[db][4 * q4] * a1 + o2.x * a2;
                oacc[db][4 * q4 + 1] = oacc[db][4 * q4 + 1] * a1 + o2.y * a2;
                oacc[db][4 * q4 + 2] = oacc[db][4 * q4 + 2] * a1 + o2.z * a2;
                oacc[db][4 * q4 + 3] = oacc[db][4 * q4 + 3] * a1 + o2.w * a2;
            }
    }
    // ---- normalise and store: 4 consecutive d per register quad
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const size_t orow = (size_t)b * a.L + qb * AT_QB + wave * 32 + r;
    if (a.nsplit > 1) {   // this split's partial state: O un-normalised, (max, sum) in log2 units
        const size_t rows = (size_t)a.B * a.L, prow = (size_t)split * rows + orow;
        float *wp = a.ws + prow * ((size_t)a.H * 64) + head * 64 + 4 * hh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                *reinterpret_cast<float4 *>(wp + 32 * db + 8 * q4) =
                    make_float4(oacc[db][4 * q4], oacc[db][4 * q4 + 1], oacc[db][4 * q4 + 2], oacc[db][4 * q4 + 3]);
        if (hh == 0) {
            float *ml = a.ws + (size_t)a.nsplit * rows * ((size_t)a.H * 64) + (prow * a.H + head) * 2;
            ml[0] = m_run;
            ml[1] = l_tot;
        }
        return;
    }
    if (a.lse && hh == 0) a.lse[orow * a.H + head] = m_run + __builtin_amdgcn_logf(l_tot);      // v_log_f32 is log2
    if (a.o_split) {   // SMK_FMT_SPLIT_BF16: feature group (64 head + 32 db + 8 q4) / 8, this lane's half (4 hi | 4 lo)
        __bf16 *os = reinterpret_cast<__bf16 *>(a.o) + orow * (2 * (size_t)a.ldo) + (head * 8) * 16 + 4 * hh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                bf16x4 vh, vl;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = oacc[db][4 * q4 + i] * inv;
                    const __bf16 t16 = (__bf16)v;
                    vh[i] = t16;
                    vl[i] = (__bf16)(v - (float)t16);
                }
                *reinterpret_cast<bf16x4 *>(os + (4 * db + q4) * 16) = vh;
                *reinterpret_cast<bf16x4 *>(os + (4 * db + q4) * 16 + 8) = vl;
            }
        return;
    }
    float *op = a.o + orow * a.ldo + head * 64 + 4 * hh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            *reinterpret_cast<float4 *>(op + 32 * db + 8 * q4) =
                make_float4(oacc[db][4 * q4] * inv, oacc[db][4 * q4 + 1] * inv, oacc[db][4 * q4 + 2] * inv, oacc[db][4 * q4 + 3] * inv);
}

// merge of the nsplit partial states of a split-key launch, in split order (deterministic): thread = (row, head, 4 consecutive d)
__global__ __launch_bounds__(256) void k_attention_combine(const float *__restrict__ ws, float *__restrict__ o, float *__restrict__ lse,
                                                           long long rows, int H, int ldo, int nsplit) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * H * 16) return;
    const int q4 = (int)(idx & 15), head = (int)((idx >> 4) % H);
    const long long row = (idx >> 4) / H;
    const float *ml = ws + (size_t)nsplit * rows * ((size_t)H * 64);
    float m = -INFINITY;
    for (int s = 0; s < nsplit; ++s) m = fmaxf(m, ml[(((size_t)s * rows + row) * H + head) * 2]);
    float l = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < nsplit; ++s) {
        const size_t prow = (size_t)s * rows + row;
        const float w = __builtin_amdgcn_exp2f(ml[(prow * H + head) * 2] - m);
        l += ml[(prow * H + head) * 2 + 1] * w;
        const float4 p = *reinterpret_cast<const float4 *>(ws + prow * ((size_t)H * 64) + head * 64 + 4 * q4);
        acc.x += p.x * w; acc.y += p.y * w; acc.z += p.z * w; acc.w += p.w * w;
    }
    const float inv = 1.0f / l;
    *reinterpret_cast<float4 *>(o + row * ldo + head * 64 + 4 * q4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    if (lse && q4 == 0) lse[row * H + head] = m + __builtin_amdgcn_logf(l);
}

// how many workgroups share the keys of one query block: grows while the grid still fits one 512-thread workgroup per CU and every wave
// group keeps at least one whole key tile
static int attention_nsplit(int B, int L, int H, int num_cu) {
    const int nwg = B * H * (L / AT_QB);
    int n = 1;
    while (n < 8 && nwg * n * 2 <= num_cu && (L / AT_KV) % (n * 2 * 2) == 0) n *= 2;
    static int force = -1;
    if (force < 0) { const char *sv = getenv("SMK_ATTN_SPLIT"); force = sv ? atoi(sv) : 0; }       // diagnostic: 1 = never split
    return force == 1 ? 1 : n;
}
size_t attention_workspace_bytes(int B, int L, int H) {
    const int n = attention_nsplit(B, L, H, device_num_cu());
    return n > 1 ? (size_t)n * B * L * H * (64 + 2) * sizeof(float) : 0;
}

template <bool KVS>
static hipError_t launch_attention_x3_t(const AttnArgs &a0, hipStream_t st) {
    const int num_cu = device_num_cu();
    AttnArgs a = a0;
    a.nsplit = (a.ws && !a.o_split) ? attention_nsplit(a.B, a.L, a.H, num_cu) : 1;
    once_per_device((const void *)k_attention_x3<1, false, KVS>, [&] {
        (void)hipFuncSetAttribute((const void *)k_attention_x3<1, false, KVS>, hipFuncAttributeMaxDynamicSharedMemorySize, AT_LDS);
        (void)hipFuncSetAttribute((const void *)k_attention_x3<2, false, KVS>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * AT_LDS);
    });
    if (a.nsplit > 1) {
        const long long rows = (long long)a.B * a.L;
        hipLaunchKernelGGL((k_attention_x3<2, false, KVS>), dim3(a.B * a.H * (a.L / AT_QB) * a.nsplit), dim3(512), 2 * AT_LDS, st, a);
        hipLaunchKernelGGL(k_attention_combine, dim3((unsigned)((rows * a.H * 16 + 255) / 256)), dim3(256), 0, st, a.ws, a.o, a.lse, rows, a.H,
                           a.ldo, a.nsplit);
        return hipGetLastError();
    }
    const int nwg = a.B * a.H * (a.L / AT_QB);
    static int force_ks = -1;
    if (force_ks < 0) { const char *sv = getenv("SMK_ATTN_KS"); force_ks = sv ? atoi(sv) : 0; }
    // a grid that leaves the second workgroup slot of every CU empty runs the split-KV form instead (8 waves per CU either way)
    int ks = (nwg <= num_cu && (a.L / AT_KV) % 2 == 0) ? 2 : 1;
    if ((force_ks == 1 || force_ks == 2) && (a.L / AT_KV) % force_ks == 0) ks = force_ks;
    static int sb = -1;
    if (sb < 0) { const char *sv = getenv("SMK_ATTN_SB"); sb = sv ? atoi(sv) : 1; }     // measured at B = 64: 553 -> 535 us (interleaved A/B)
    if (ks == 2) hipLaunchKernelGGL((k_attention_x3<2, false, KVS>), dim3(nwg), dim3(512), 2 * AT_LDS, st, a);
    else if (sb && nwg > 2 * num_cu) hipLaunchKernelGGL((k_attention_x3<1, true, KVS>), dim3(nwg), dim3(256), AT_BUF, st, a);
    else hipLaunchKernelGGL((k_attention_x3<1, false, KVS>), dim3(nwg), dim3(256), AT_LDS, st, a);
    return hipGetLastError();
}

hipError_t launch_attention_x3(const AttnArgs &a, hipStream_t st) {
    return a.kv_split ? launch_attention_x3_t<true>(a, st) : launch_attention_x3_t<false>(a, st);
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of the attention above, same split-bf16 arithmetic (every product hi*hi + hi*lo + lo*hi, fp32 accumulate).
// With s_ij = scale q_i.k_j, P = softmax(s) (recomputed from the saved log-sum-exp) and delta_i = sum_d dO_id O_id:
//     dP_ij = dO_i.v_j      dS_ij = P_ij (dP_ij - delta_i)      dq_i = scale sum_j dS_ij k_j
//     dk_j = scale sum_i dS_ij q_i                              dv_j = sum_i P_ij dO_i
// ONE kernel body serves both passes: a workgroup owns 128 "outer" rows (wave w: 32 of them, one per lane and lane half) whose
// operands X (for the scores) and U (for dP) stay in registers as MFMA B fragments, and walks 64-row "inner" tiles Y, W staged
// through LDS:   S^T = mfma(Y, X)   T^T = mfma(W, U)   -> lane = outer row, accumulator registers = 32 of the tile's inner rows,
// exactly the forward's S^T layout; the elementwise results are used IN PLACE as the B fragments of out1^T = mfma(Y^T, dS)
// (and out2^T = mfma(W^T, P)) against transposed LDS images, like the forward's O^T = mfma(V^T, P).
//     DKV = false (dq pass): outer = queries  X = q * scale * log2e   U = dO   inner: Y = k   W = v    out1 = dq; lse / delta per lane
//     DKV = true  (dk, dv):  outer = keys     X = k * scale * log2e   U = v    inner: Y = q   W = dO   out1 = dk, out2 = dv; lse / delta
//                            of the tile's 64 queries from LDS (they run along the accumulator registers)
// Single-buffered LDS (54 KB / 72 KB -> 2 workgroups per CU): the next tile's global loads are in flight during the MFMAs.
constexpr int AB_RPITCH = 144, AB_TPITCH = 136;
constexpr int AB_RPLANE = 64 * AB_RPITCH, AB_TPLANE = 64 * AB_TPITCH;            // 9216, 8704
constexpr int AB_YR = 0, AB_YT = 2 * AB_RPLANE, AB_WR = AB_YT + 2 * AB_TPLANE, AB_WT = AB_WR + 2 * AB_RPLANE;
constexpr int AB_LDS_DQ = AB_WT;                                                   // 54,272 B
constexpr int AB_SCAL = AB_WT + 2 * AB_TPLANE, AB_LDS_DKV = AB_SCAL + 2 * 64 * 4;   // 72,192 B

template <bool DKV>
__global__ __launch_bounds__(256, 2) void k_attention_bwd_x3(const AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int vid = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int nob = a.L / 128;
    const int ob = vid % nob, bh = vid / nob, head = bh % a.H, b = bh / a.H;
    const int NT = a.L / 64;
    const float *Xp = DKV ? a.k : a.q, *Up = DKV ? a.v : a.dout, *Yp = DKV ? a.q : a.k, *Wp = DKV ? a.dout : a.v;
    const int ldX = DKV ? a.ldk : a.ldq, ldU = DKV ? a.ldv : a.ldo, ldY = DKV ? a.ldq : a.ldk, ldW = DKV ? a.ldo : a.ldv;
    const size_t orow = (size_t)b * a.L + ob * 128 + wave * 32 + r;          // this lane's outer row

    // ---- outer operands as B fragments (column = outer row r, k = d 16s + 8hh + j), split; X pre-scaled
    bf16x8 xh[4], xl[4], uh[4], ul[4];
    {
        const float *xp = Xp + orow * ldX + head * 64 + 8 * hh, *up = Up + orow * ldU + head * 64 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 x0 = *reinterpret_cast<const float4 *>(xp + 16 * s), x1 = *reinterpret_cast<const float4 *>(xp + 16 * s + 4);
            const float4 u0 = *reinterpret_cast<const float4 *>(up + 16 * s), u1 = *reinterpret_cast<const float4 *>(up + 16 * s + 4);
            const float fx[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            const float fu[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = fx[j] * a.scale_log2e;
                const __bf16 tx = (__bf16)x, tu = (__bf16)fu[j];
                xh[s][j] = tx; xl[s][j] = (__bf16)(x - (float)tx);
                uh[s][j] = tu; ul[s][j] = (__bf16)(fu[j] - (float)tu);
            }
        }
    }
    float lse_o = 0.f, del_o = 0.f;
    if (!DKV) { lse_o = a.lse[orow * a.H + head]; del_o = a.delta[orow * a.H + head]; }

    // ---- inner tiles: thread = 4 x 4 block (rows 4rq + i, columns 4c4 .. 4c4+3) of Y and of W
    const int c4 = tid & 15, rq = tid >> 4;
    float4 yst[4], wst[4];
    float sc_st = 0.f;
    auto stage_load = [&](int t) {
        const size_t row0 = (size_t)b * a.L + t * 64 + 4 * rq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            yst[i] = *reinterpret_cast<const float4 *>(Yp + (row0 + i) * ldY + head * 64 + c4 * 4);
            wst[i] = *reinterpret_cast<const float4 *>(Wp + (row0 + i) * ldW + head * 64 + c4 * 4);
        }
        if (DKV && tid < 128) {
            const size_t qrow = (size_t)b * a.L + t * 64 + (tid & 63);
            sc_st = (tid < 64 ? a.lse : a.delta)[qrow * a.H + head];
        }
    };
    auto store_image = [&](const float4 (&st)[4], int off_r, int off_t, bool with_t) {
        bf16x4 h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) at_split4(st[i], h[i], l[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                          // row-major [row 4rq + i][d 4c4 ..]
            unsigned char *p = smem + off_r + (4 * rq + i) * AB_RPITCH + c4 * 8;
            *reinterpret_cast<bf16x4 *>(p) = h[i];
            *reinterpret_cast<bf16x4 *>(p + AB_RPLANE) = l[i];
        }
        if (with_t) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {                                      // transposed [d 4c4 + e][rows 4rq .. 4rq+3]
                bf16x4 th, tl;
#pragma unroll
                for (int i = 0; i < 4; ++i) { th[i] = h[i][e]; tl[i] = l[i][e]; }
                unsigned char *p = smem + off_t + (c4 * 4 + e) * AB_TPITCH + rq * 8;
                *reinterpret_cast<bf16x4 *>(p) = th;
                *reinterpret_cast<bf16x4 *>(p + AB_TPLANE) = tl;
            }
        }
    };

    f32x16 o1[2], o2[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 16; ++g) { o1[db][g] = 0.f; o2[db][g] = 0.f; }

    const unsigned char *yr_h = smem + AB_YR + r * AB_RPITCH + hh * 16, *wr_h = smem + AB_WR + r * AB_RPITCH + hh * 16;
    const unsigned char *yt_h = smem + AB_YT + r * AB_TPITCH + hh * 8, *wt_h = smem + AB_WT + r * AB_TPITCH + hh * 8;
    const float *scal = reinterpret_cast<const float *>(smem + AB_SCAL);

    stage_load(0);
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
        store_image(yst, AB_YR, AB_YT, true);
        store_image(wst, AB_WR, AB_WT, DKV);
        if (DKV && tid < 128) reinterpret_cast<float *>(smem + AB_SCAL)[tid] = sc_st;
        if (t + 1 < NT) stage_load(t + 1);
        __syncthreads();

        // The 64 inner rows go in two halves of 32 (kb): scores, elementwise, output MFMAs -- one half's accumulators and fragments
        // live at a time (the whole tile at once needs ~290 registers in the dk/dv pass).
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            // ---- S^T and T^T: acc[g] = value(outer r, inner 32kb + (g&3) + 8(g>>2) + 4hh)
            // the accumulators start at -lse and -delta, so the MFMAs leave S - lse and dP - delta (no subtraction per element)
            f32x16 sacc, tacc;
            if (DKV) {      // inner rows 32kb + 8 (g>>2) + 4hh + (g&3): the tile's lse / delta values from LDS
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 l4 = *reinterpret_cast<const float4 *>(scal + 32 * kb + 8 * q + 4 * hh);
                    const float4 d4 = *reinterpret_cast<const float4 *>(scal + 64 + 32 * kb + 8 * q + 4 * hh);
                    sacc[4 * q] = -l4.x; sacc[4 * q + 1] = -l4.y; sacc[4 * q + 2] = -l4.z; sacc[4 * q + 3] = -l4.w;
                    tacc[4 * q] = -d4.x; tacc[4 * q + 1] = -d4.y; tacc[4 * q + 2] = -d4.z; tacc[4 * q + 3] = -d4.w;
                }
            } else {
#pragma unroll
                for (int g = 0; g < 16; ++g) { sacc[g] = -lse_o; tacc[g] = -del_o; }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 yh = *reinterpret_cast<const bf16x8 *>(yr_h + kb * 32 * AB_RPITCH + s * 32);
                const bf16x8 yl = *reinterpret_cast<const bf16x8 *>(yr_h + AB_RPLANE + kb * 32 * AB_RPITCH + s * 32);
                const bf16x8 wh = *reinterpret_cast<const bf16x8 *>(wr_h + kb * 32 * AB_RPITCH + s * 32);
                const bf16x8 wl = *reinterpret_cast<const bf16x8 *>(wr_h + AB_RPLANE + kb * 32 * AB_RPITCH + s * 32);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yl, xh[s], sacc, 0, 0, 0);
                tacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, uh[s], tacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh, xl[s], sacc, 0, 0, 0);
                tacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ul[s], tacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh, xh[s], sacc, 0, 0, 0);
                tacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, uh[s], tacc, 0, 0, 0);
            }

            // ---- P = exp2(S - lse), dS = P * (dP - delta) (x scale at the end); both end up as B fragments (k = inner row, column = outer row r)
            bf16x8 dsh[2], dsl[2], ph[2], pl[2];
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int g = 8 * sl + j;
                    const float p = __builtin_amdgcn_exp2f(sacc[g]);
                    const float ds = p * tacc[g];                      // the softmax scale is applied once, to the finished out1 accumulators
                    const __bf16 dt = (__bf16)ds;
                    dsh[sl][j] = dt;
                    dsl[sl][j] = (__bf16)(ds - (float)dt);
                    if (DKV) {
                        const __bf16 pt = (__bf16)p;
                        ph[sl][j] = pt;
                        pl[sl][j] = (__bf16)(p - (float)pt);
                    }
                }
            }

            // ---- out1^T += Y^T dS (and out2^T += W^T P): o[db][g] = out(outer r, d 32db + (g&3) + 8(g>>2) + 4hh)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int s4 = 2 * kb + sl;
                bf16x8 th[2], tl[2];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const unsigned char *p0 = yt_h + db * 32 * AB_TPITCH + s4 * 32;          // inner rows 16 s4 + 4hh .. +3 | +8
                    const bf16x4 h0 = *reinterpret_cast<const bf16x4 *>(p0), h1 = *reinterpret_cast<const bf16x4 *>(p0 + 16);
                    const bf16x4 l0 = *reinterpret_cast<const bf16x4 *>(p0 + AB_TPLANE), l1 = *reinterpret_cast<const bf16x4 *>(p0 + AB_TPLANE + 16);
                    th[db] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                    tl[db] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int db = 0; db < 2; ++db) o1[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tl[db], dsh[sl], o1[db], 0, 0, 0);
#pragma unroll
                for (int db = 0; db < 2; ++db) o1[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th[db], dsl[sl], o1[db], 0, 0, 0);
#pragma unroll
                for (int db = 0; db < 2; ++db) o1[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th[db], dsh[sl], o1[db], 0, 0, 0);
                if (DKV) {
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const unsigned char *p0 = wt_h + db * 32 * AB_TPITCH + s4 * 32;
                        const bf16x4 h0 = *reinterpret_cast<const bf16x4 *>(p0), h1 = *reinterpret_cast<const bf16x4 *>(p0 + 16);
                        const bf16x4 l0 = *reinterpret_cast<const bf16x4 *>(p0 + AB_TPLANE), l1 = *reinterpret_cast<const bf16x4 *>(p0 + AB_TPLANE + 16);
                        th[db] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                        tl[db] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#pragma unroll
                    for (int db = 0; db < 2; ++db) o2[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tl[db], ph[sl], o2[db], 0, 0, 0);
#pragma unroll
                    for (int db = 0; db < 2; ++db) o2[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th[db], pl[sl], o2[db], 0, 0, 0);
#pragma unroll
                    for (int db = 0; db < 2; ++db) o2[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th[db], ph[sl], o2[db], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                       // every read of this tile's images has returned
    }

    // ---- store: 4 consecutive d per register quad
    float *p1 = (DKV ? a.dk : a.dq) + orow * (DKV ? a.lddk : a.lddq) + head * 64 + 4 * hh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            *reinterpret_cast<float4 *>(p1 + 32 * db + 8 * q4) = make_float4(a.scale * o1[db][4 * q4], a.scale * o1[db][4 * q4 + 1],
                                                                             a.scale * o1[db][4 * q4 + 2], a.scale * o1[db][4 * q4 + 3]);
    if (DKV) {
        float *p2 = a.dv + orow * a.lddv + head * 64 + 4 * hh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                *reinterpret_cast<float4 *>(p2 + 32 * db + 8 * q4) = make_float4(o2[db][4 * q4], o2[db][4 * q4 + 1], o2[db][4 * q4 + 2], o2[db][4 * q4 + 3]);
    }
}

hipError_t launch_attention_bwd_x3(const AttnBwdArgs &a, hipStream_t st) {
    once_per_device((const void *)k_attention_bwd_x3<false>, [&] {
        (void)hipFuncSetAttribute((const void *)k_attention_bwd_x3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, AB_LDS_DQ);
        (void)hipFuncSetAttribute((const void *)k_attention_bwd_x3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, AB_LDS_DKV);
    });
    const int nwg = a.B * a.H * (a.L / 128);
    hipLaunchKernelGGL(k_attention_bwd_x3<true>, dim3(nwg), dim3(256), AB_LDS_DKV, st, a);
    hipLaunchKernelGGL(k_attention_bwd_x3<false>, dim3(nwg), dim3(256), AB_LDS_DQ, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// nn.LayerNorm over the last dimension (smokephys_net.py:149-150,161,165; eps 1e-5, biased variance): one wave per token row,
// the row held in registers (D <= 2048: up to 8 float4 per lane), mean and centred second moment by wave reductions, one
// read and one write of the row -- an HBM-bound stream.
template <int NV>   // float4 per lane
__global__ __launch_bounds__(256) void k_layernorm(const LayerNormArgs a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const float *xp = a.x + (long long)row * a.ldx;
    float4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < a.D ? *reinterpret_cast<const float4 *>(xp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)a.D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < a.D) {
            const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
            sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)a.D + a.eps);
    float *yp = a.y + (long long)row * a.ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < a.D) {
            const float4 w = *reinterpret_cast<const float4 *>(a.w + c), b = *reinterpret_cast<const float4 *>(a.b + c);
            const float o[4] = {(v[i].x - mean) * rstd * w.x + b.x, (v[i].y - mean) * rstd * w.y + b.y,
                                (v[i].z - mean) * rstd * w.z + b.z, (v[i].w - mean) * rstd * w.w + b.w};
            if (a.y_split) {   // SMK_FMT_SPLIT_BF16: group c / 8, half (c / 4) & 1
                bf16x4 vh, vl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const __bf16 t16 = (__bf16)o[j];
                    vh[j] = t16;
                    vl[j] = (__bf16)(o[j] - (float)t16);
                }
                __bf16 *ys = reinterpret_cast<__bf16 *>(a.y) + (size_t)row * (2 * (size_t)a.D) + (c >> 3) * 16 + (c & 4);
                *reinterpret_cast<bf16x4 *>(ys) = vh;
                *reinterpret_cast<bf16x4 *>(ys + 8) = vl;
            } else {
                *reinterpret_cast<float4 *>(yp + c) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

// Backward: a wave walks rows (row in registers, statistics recomputed), writes dx = rstd (g - mean(g) - xhat mean(g xhat)) with g = dy w and
// keeps per-lane column partials of dy xhat and dy; the 4 waves of a workgroup merge theirs through LDS into one [2][D] partial, and
// k_layernorm_bwd_finish adds the LN_BWD_WGS partials per column in a fixed order (deterministic, no atomics).
template <int NV>
__global__ __launch_bounds__(256) void k_layernorm_bwd(const LayerNormBwdArgs a) {
    __shared__ float red[3][2][NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 wv[NV], dwp[NV], dbp[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        wv[i] = c < a.D ? *reinterpret_cast<const float4 *>(a.w + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        dwp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        dbp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float invD = 1.0f / (float)a.D;
    for (int row = blockIdx.x * 4 + wave; row < a.rows; row += gridDim.x * 4) {
        const float *xp = a.x + (long long)row * a.ldx, *gp = a.dy + (long long)row * a.lddy;
        float4 v[NV], g[NV];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            const bool ok = c < a.D;
            v[i] = ok ? *reinterpret_cast<const float4 *>(xp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            g[i] = ok ? *reinterpret_cast<const float4 *>(gp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum * invD;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < a.D) {
                const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
                sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        const float rstd = 1.0f / sqrtf(sq * invD + a.eps);
        float s1 = 0.f, s2 = 0.f;                              // sum(g w), sum(g w xhat)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < a.D) {
                const float xh[4] = {(v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd};
                const float gy[4] = {g[i].x, g[i].y, g[i].z, g[i].w}, ww[4] = {wv[i].x, wv[i].y, wv[i].z, wv[i].w};
                dwp[i].x += gy[0] * xh[0]; dwp[i].y += gy[1] * xh[1]; dwp[i].z += gy[2] * xh[2]; dwp[i].w += gy[3] * xh[3];
                dbp[i].x += gy[0]; dbp[i].y += gy[1]; dbp[i].z += gy[2]; dbp[i].w += gy[3];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gw = gy[j] * ww[j];
                    s1 += gw;
                    s2 += gw * xh[j];
                }
                v[i] = make_float4(xh[0], xh[1], xh[2], xh[3]);                 // keep xhat
                g[i] = make_float4(gy[0] * ww[0], gy[1] * ww[1], gy[2] * ww[2], gy[3] * ww[3]);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        const float c1 = s1 * invD, c2 = s2 * invD;
        float *dp = a.dx + (long long)row * a.lddx;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < a.D)
                *reinterpret_cast<float4 *>(dp + c) = make_float4(rstd * (g[i].x - c1 - v[i].x * c2), rstd * (g[i].y - c1 - v[i].y * c2),
                                                                  rstd * (g[i].z - c1 - v[i].z * c2), rstd * (g[i].w - c1 - v[i].w * c2));
        }
    }
    // merge the four waves' column partials (waves 1..3 park theirs, wave 0 adds them in wave order)
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            *reinterpret_cast<float4 *>(&red[wave - 1][0][(i * 64 + lane) * 4]) = dwp[i];
            *reinterpret_cast<float4 *>(&red[wave - 1][1][(i * 64 + lane) * 4]) = dbp[i];
        }
    }
    __syncthreads();
    if (wave == 0) {
        float *pw = a.part + (size_t)blockIdx.x * 2 * a.D, *pb = pw + a.D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < a.D) {
                float4 sw = dwp[i], sb = dbp[i];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float4 tw = *reinterpret_cast<const float4 *>(&red[k][0][c]), tb = *reinterpret_cast<const float4 *>(&red[k][1][c]);
                    sw.x += tw.x; sw.y += tw.y; sw.z += tw.z; sw.w += tw.w;
                    sb.x += tb.x; sb.y += tb.y; sb.z += tb.z; sb.w += tb.w;
                }
                *reinterpret_cast<float4 *>(pw + c) = sw;
                *reinterpret_cast<float4 *>(pb + c) = sb;
            }
        }
    }
}

// 16 columns x 16 partial-sum lanes per workgroup: lane kl adds partials kl, kl + 16, ... (in that order), then the 16 lanes of a column are
// added in lane order -- a fixed order, so the result does not depend on scheduling.  (One thread per column walking all 1,024 partials
// serially took 0.24 ms per call: 1,024 dependent L2 round trips.)
__global__ __launch_bounds__(256) void k_layernorm_bwd_finish(const LayerNormBwdArgs a, int nwg) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;                          // column of [dw | db]
    float acc = 0.f;
    if (c < 2 * a.D)
        for (int k = kl; k < nwg; k += 16) acc += a.part[(size_t)k * 2 * a.D + c];
    red[kl][cl] = acc;
    __syncthreads();
    if (kl == 0 && c < 2 * a.D) {
        float s = red[0][cl];
#pragma unroll
        for (int j = 1; j < 16; ++j) s += red[j][cl];
        (c < a.D ? a.dw : a.db - a.D)[c] = s;
    }
}

hipError_t launch_layernorm_bwd(const LayerNormBwdArgs &a, hipStream_t st) {
    const int nwg = a.rows / 4 < LN_BWD_WGS ? (a.rows + 3) / 4 : LN_BWD_WGS;
    const dim3 grid(nwg), block(256);
    const int nv = (a.D + 255) / 256;
    if (nv <= 1) hipLaunchKernelGGL(k_layernorm_bwd<1>, grid, block, 0, st, a);
    else if (nv <= 2) hipLaunchKernelGGL(k_layernorm_bwd<2>, grid, block, 0, st, a);
    else if (nv <= 4) hipLaunchKernelGGL(k_layernorm_bwd<4>, grid, block, 0, st, a);
    else if (nv <= 8) hipLaunchKernelGGL(k_layernorm_bwd<8>, grid, block, 0, st, a);
    else return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_layernorm_bwd_finish, dim3((2 * a.D + 15) / 16), block, 0, st, a, nwg);
    return hipGetLastError();
}

// delta[row][h] = sum_d dout[row][64 h + d] * out[row][64 h + d] (the softmax-backward row term of chaos_attention.py:108-112 under autograd):
// one wave per token row per 8 heads -- lane l holds 8 consecutive columns, so a head is 8 lanes -- one read of both tensors.
__global__ __launch_bounds__(256) void k_attn_delta(const float *__restrict__ dout, const float *__restrict__ out, long long rows, int H,
                                                    long long ldd, long long ldo, float *__restrict__ delta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hb = blockIdx.y * 8;                              // this wave's 8 heads: columns 64 hb .. 64 hb + 511
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        const int h = hb + (lane >> 3);
        float s = 0.f;
        if (h < H) {
            const float4 *a = reinterpret_cast<const float4 *>(dout + row * ldd + 64 * hb + 8 * lane);
            const float4 *b = reinterpret_cast<const float4 *>(out + row * ldo + 64 * hb + 8 * lane);
            const float4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
            s = ((a0.x * b0.x + a0.y * b0.y) + (a0.z * b0.z + a0.w * b0.w)) + ((a1.x * b1.x + a1.y * b1.y) + (a1.z * b1.z + a1.w * b1.w));
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if ((lane & 7) == 0 && h < H) delta[row * H + h] = s;
    }
}

hipError_t launch_attn_delta(const float *dout, const float *out, long long rows, int H, long long ldd, long long ldo, float *delta,
                             hipStream_t st) {
    long long blocks = (rows + 3) / 4;
    const long long cap = (long long)device_num_cu() * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_attn_delta, dim3((unsigned)blocks, (H + 7) / 8), dim3(256), 0, st, dout, out, rows, H, ldd, ldo, delta);
    return hipGetLastError();
}

hipError_t launch_layernorm(const LayerNormArgs &a, hipStream_t st) {
    const dim3 grid((a.rows + 3) / 4), block(256);
    const int nv = (a.D + 255) / 256;
    if (nv <= 1) hipLaunchKernelGGL(k_layernorm<1>, grid, block, 0, st, a);
    else if (nv <= 2) hipLaunchKernelGGL(k_layernorm<2>, grid, block, 0, st, a);
    else if (nv <= 4) hipLaunchKernelGGL(k_layernorm<4>, grid, block, 0, st, a);
    else if (nv <= 8) hipLaunchKernelGGL(k_layernorm<8>, grid, block, 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace smk
