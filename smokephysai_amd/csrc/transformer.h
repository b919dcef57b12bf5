#pragma once
#include "common.h"

namespace smk {

struct ChaosAddendArgs {
    const float *noise;                  // [3][B] standard-normal draws (x0, y0, z0 before the 0.1 scale)
    const float *proj_w, *proj_b;        // chaos_proj: [D][3], [D]
    const float *gate_w, *gate_b;        // chaos_gate: [D] (= [1][D]), [1]
    float *addend;                       // [B][5][ld] (columns 0..D-1 written)
    int B, D, ld;
    float strength, sigma, rho, beta, dt;
};
hipError_t launch_chaos_addend(const ChaosAddendArgs &a, hipStream_t st);
// the addends of up to 8 layers (each with its own noise, weights and output) as ONE launch: grid (B, layers)
struct ChaosAddendBatch { int NL; ChaosAddendArgs layer[8]; };
hipError_t launch_chaos_addend_batch(const ChaosAddendBatch &a, hipStream_t st);

// SmokePhysNet's tail behind the transformer (smokephys_net.py:116-118): latent = features.mean(dim=1), physics = Linear(ReLU(Linear(latent))).
// Three small launches: per-chunk token sums ([B][32 chunks][D] workspace); the chunk sums in order -> mean, and the hidden layer, 32
// units per workgroup (a wave per 8 rows, lanes along the input: coalesced weight reads, fixed summation order); the output layer.
struct PooledHeadArgs {
    const float *x; long long ldx;       // features [B][L][ldx], D columns used
    int B, L, D;
    const float *w1, *b1; int H1;        // [H1][D], [H1]
    const float *w2, *b2; int H2;        // [H2][H1], [H2]
    float *pooled;                       // [B][D]
    float *out;                          // [B][H2]
    float *ws;                           // [B][32][D] chunk sums, then [B][H1] hidden
};
hipError_t launch_pooled_head(const PooledHeadArgs &a, hipStream_t st);
// the five Lorenz states [B][5][3] of the same noise (no projection / gate)
hipError_t launch_lorenz_states(const float *noise, int B, float sigma, float rho, float beta, float dt, float *states, hipStream_t st);

struct AttnArgs {
    const float *q, *k, *v;              // [B][L][ld*]: head h = columns 64h .. 64h+63 of a token row
    float *o;                            // [B][L][ldo], same column convention
    int ldq, ldk, ldv, ldo;              // row pitches in floats
    int B, L, H;
    float scale_log2e;                   // softmax scale * log2(e), folded into Q
    int o_split;                         // o in SMK_FMT_SPLIT_BF16 (dense rows)
    float *lse = nullptr;                // optional [B][L][H]: log2 of sum_j exp2(score_ij * scale * log2 e) -- saved for the backward
    // small grids (fewer workgroups than CUs): the keys of one (batch, head, query block) are dealt to nsplit workgroups; each writes its
    // un-normalised partial output and its (running max, sum) to the workspace, k_attention_combine merges them in split order
    float *ws = nullptr;                 // [nsplit][B L][H 64] partial outputs, then [nsplit][B L][H][2] (max, sum); attention_workspace_bytes
    int nsplit = 1;                      // set by the launcher
    int kv_split = 0;                    // k and v hold in-place split-bf16 (SMK_FMT_SPLIT4_INPLACE: per 4 columns {hi[0..3], lo[0..3]}) instead of fp32
};
hipError_t launch_attention_x3(const AttnArgs &a, hipStream_t st);
// bytes of workspace with which launch_attention_x3 splits the keys over workgroups for this problem on the current device (0: it would not)
size_t attention_workspace_bytes(int B, int L, int H);

// Backward of the same attention (autograd of chaos_attention.py:102-112 with the chaos term folded into q): dq, dk, dv from q, k, v,
// the output gradient, the forward's log-sum-exp and delta = rowsum(dout * out).
struct AttnBwdArgs {
    const float *q, *k, *v, *dout;       // [B][L][ld*], head h = columns 64h .. 64h+63
    const float *lse, *delta;            // [B][L][H]
    float *dq, *dk, *dv;                 // [B][L][ldd*]
    int ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    int B, L, H;
    float scale, scale_log2e;
};
hipError_t launch_attention_bwd_x3(const AttnBwdArgs &a, hipStream_t st);

// delta [rows][H] = per-head row dot products of dout and out (head_dim 64), the row term of the attention backward
hipError_t launch_attn_delta(const float *dout, const float *out, long long rows, int H, long long ldd, long long ldo, float *delta,
                             hipStream_t st);

struct LayerNormArgs {
    const float *x; float *y;            // [rows][ld*]
    const float *w, *b;                  // [D]
    long long ldx, ldy;
    int rows, D;
    float eps;
    int y_split;                         // y in SMK_FMT_SPLIT_BF16 (dense rows)
};
hipError_t launch_layernorm(const LayerNormArgs &a, hipStream_t st);

// Backward of that LayerNorm (autograd, train.py:89): dx per row, dw / db as column sums over the rows (two-stage, fixed order).
struct LayerNormBwdArgs {
    const float *x, *dy;                 // [rows][ld*]
    const float *w;                      // [D]
    float *dx;                           // [rows][lddx]
    float *dw, *db;                      // [D]
    float *part;                         // workspace [LN_BWD_WGS][2][D]
    long long ldx, lddy, lddx;
    int rows, D;
    float eps;
};
constexpr int LN_BWD_WGS = 1024;
hipError_t launch_layernorm_bwd(const LayerNormBwdArgs &a, hipStream_t st);

}  // namespace smk
