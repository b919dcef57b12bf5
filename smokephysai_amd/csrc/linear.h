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
    // K-segmented form (weight gradients: a reduction over 65,536 token rows with only a handful of output tiles): nseg independent
    // problems y[s] = x[:, s*K:(s+1)*K] w[:, s*K:(s+1)*K]^T share one launch; l.K is the SEGMENT length, the weights hold nseg*K
    // k in one layout, y is nseg dense [M][N] slabs.  nseg = 1: the plain layer.
    int nseg = 1;
    // LayerNorm fused in front of the layer (one-tile-per-workgroup problems: a single frame): x is the RAW row, the handle holds
    // W' = W diag(gamma) and b' = b + W beta, ln_wsum[n] = sum_k W'[n][k]; the kernel gathers each row's sum and sum of squares while it
    // stages the row and finishes y = rstd (x W'^T - mean ln_wsum) + b' in the epilogue.  Null: the plain layer.
    const float *ln_wsum = nullptr;
    float ln_eps = 0.f;
    // fp32 y only: columns >= split_from (a multiple of 32; < 0: none) are written as in-place split-bf16 -- every aligned group of 4 columns
    // holds {hi[0..3], lo[0..3]} (bf16: hi = RNE(v), lo = RNE(v - hi)) in the 16 bytes of its 4 floats (SMK_FMT_SPLIT4_INPLACE, smokehip.h)
    int split_from = -1;
};

hipError_t launch_split_linear_weights(const float *w, const float *bias, const LinearDev &l, hipStream_t st, int transposed = 0,
                                       long long ld = 0, int k_valid = -1);
hipError_t launch_linear_x3(const LinearDev &l, const LinearCall &c, hipStream_t st);
// Conv3d(64 -> N, 3 x 3 x 3, padding 1 inside the slab) as an implicit GEMM (linear.hip, k_linear_b16<NW, true>): slab [Dl][H][W][64]
// channels-last, output rows = voxels of planes z_off .. z_off+nz-1, weights = a layer handle with K = 27 * 64 (column tap * 64 + c)
hipError_t launch_conv3d_cl_b16(const LinearDev &l, const float *slab, int Dl, int H, int W, int z_off, int nz, float *y, long long ldy, int act,
                                hipStream_t st);

// Conv3d(1 -> N, 7 x 7 x 7, padding 3 inside the slab) of a scalar slab [Dl][H][W] as an implicit GEMM: K = 448 = 56 window rows x 8 kx slots
hipError_t launch_conv3d_s7_b16(const LinearDev &l, const float *slab, int Dl, int H, int W, int z_off, int nz, float *y, long long ldy, int act,
                                hipStream_t st);

// the 3 x 3 x 3 convolution of a whole channels-last volume marched along z, depth-summed (conv3d_march.hip): zsum [H][W][128]
hipError_t launch_conv3d_march(const LinearDev &l, const float *a1, int D, int H, int W, float *zsum, int act, hipStream_t st);

// the 7 x 7 x 7 convolution of a whole scalar volume marched along z (conv3d_march.hip): a1 [D][H][W][64]; weights K = 448 = (kz, 8 ky, 8 kx)
hipError_t launch_conv3d_s7_march(const LinearDev &l, const float *x, int D, int H, int W, float *a1, int act, hipStream_t st);

// dW = dY^T X (linear.hip): workspace layout and segment count for a problem size
struct WgradPlan { int nseg; long long rows_pad; size_t off_wq, off_part, off_col, bytes; };
WgradPlan plan_linear_wgrad(long long rows, int out_features, int in_features);
hipError_t launch_linear_wgrad(const float *dy, long long ld_dy, const float *x, long long ldx, long long rows, int out_features,
                               int in_features, float *dw, float *db, void *workspace, hipStream_t st);

}  // namespace smk
