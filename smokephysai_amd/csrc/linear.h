#pragma once
#include "common.h"

namespace smk {

// One nn.Linear's weights, re-laid-out for the split-bf16 MFMA kernel (library-owned).
struct LinearDev {
    unsigned short *wq;   // [K/16 k-steps][2 hi|lo][N][16 k] bf16 bits: a wave's B fragment (32 columns) is 1 KiB contiguous
    float *bias;          // [N] (zeros when the layer has none)
    int N, K;
};

struct LinearCall {
    const float *x; long long ldx;        // [M][K] activations, row pitch ldx floats
    float *y; long long ldy;              // [M][N]
    const float *res; long long ldr;      // optional residual added after the activation (y = res + act(xW^T + b))
    const float *padd;                    // optional row-periodic addend [M / rows_per_group][period][N], added before the activation
    int rows_per_group, period;
    int M;
    int act;                              // 0 none, 1 GELU (erf form), 2 ReLU
    int x_split, y_split;                 // SMK_FMT_SPLIT_BF16 on the input / output side (rows dense: ldx = K, ldy = N)
};

hipError_t launch_split_linear_weights(const float *w, const float *bias, const LinearDev &l, hipStream_t st, int transposed = 0);
hipError_t launch_linear_x3(const LinearDev &l, const LinearCall &c, hipStream_t st);

}  // namespace smk
