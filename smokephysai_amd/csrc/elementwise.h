#pragma once
#include "common.h"

namespace smk {

// The element-wise chain of the pre-LN block's FFN under autograd (smokephys_net.py:153-159,165-167: Linear -> GELU -> Dropout ->
// Linear -> Dropout, x = x + ffn(...)), as HBM-rate kernels of one read and one write per tensor:
//   gelu_dropout:  a = dropout(gelu(h))           backward: dh = da * mask / (1 - p) * gelu'(h)       (h re-read; no mask tensor)
//   dropout_add:   out = residual + dropout(y)    backward: dy = dout * mask / (1 - p)                (the residual's gradient is dout)
// The keep mask of element i is a pure function of (seed, i) -- a counter-based hash, recomputed in the backward.
struct EltArgs {
    const float *a, *b;          // inputs (b: second operand or NULL)
    float *out;
    long long n;                 // elements (a multiple of 4)
    float p;                     // dropout probability in [0, 1)
    unsigned long long seed;
};
hipError_t launch_gelu_dropout_fwd(const EltArgs &e, hipStream_t st);      // a = h            -> out = dropout(gelu(h))
hipError_t launch_gelu_dropout_bwd(const EltArgs &e, hipStream_t st);      // a = h, b = dout  -> out = dh
hipError_t launch_dropout_add_fwd(const EltArgs &e, hipStream_t st);       // a = y, b = res   -> out = res + dropout(y)
hipError_t launch_dropout_bwd(const EltArgs &e, hipStream_t st);           // a = dout         -> out = dy

// out[i] = (sum over `world` shards, rank order, fp32) / world; shards `stride` elements apart; in / out fp32 or bf16 (the wire dtype of
// the direct gradient exchange, utils/distributed.py); 16-byte (fp32) / 8-byte (bf16) aligned operands, stride a multiple of 4
hipError_t launch_reduce_shards(const void *in, int in_bf16, int world, long long n, long long stride, void *out, int out_bf16, hipStream_t st);

}  // namespace smk
