#pragma once
#include "common.h"

namespace smk {

// Folded / re-laid-out encoder weights on the device (library-owned).
struct EncoderDev {
    float *w1;    // [64][49]           conv1 weights
    float *s1, *t1;   // [64]           folded BN1 scale / shift (conv bias included)
    float *w2t;   // [9][64][128]       conv2 weights, [tap][c][o]
    float *s2, *t2;   // [128]
    // split-bf16 copies for the bf16 MFMA kernels (value = hi + lo, each a bf16):
    unsigned short *w1p;   // [2 hi/lo][64 ch][64 k], k = 8*ki + kj (ki = 7 or kj = 7: zero)
    signed char *w2i;      // [18 k-steps = tap*2 + c/32][2 limbs h|l][128 o][32 c] int8: w = sw2[o] * (256 h + l)
    float *sw2;            // [128] per-output-channel weight scale of the int8 limbs
    int *wsum;             // [2][128] 128 * sum_k of the h / l weight limbs (offset correction of the unsigned activations)
    unsigned short *w2q;   // [36 k-steps = tap*4 + c/16][2 hi/lo][128 o][16 c]  (B fragments, 1 KiB per wave load)
    unsigned short *w2s;   // [18 k-steps = tap*2 + c/32][2 hi/lo][128 o][32 c]  (16x16x32 B fragments: 16 o x 64 B = 1 KiB)
};

hipError_t launch_fold_weights(const smk_encoder_weights &w, const EncoderDev &e, hipStream_t st);
hipError_t launch_conv1_only(const float *frames, int64_t fstride, int B, int H, int W, const EncoderDev &e, float *act,
                             hipStream_t st);
hipError_t launch_encoder_f32(const float *frames, int64_t fstride, int B, int H, int W, const EncoderDev &e,
                              float *features, hipStream_t st);

// x3 = true: split-bf16 (hi*hi + hi*lo + lo*hi, ~fp32 accuracy); false: single-pass bf16.
// tokens = true: features written token-major [B][32*32][128] (coalesced; the layout feature_proj consumes).
hipError_t launch_encoder_bf16(const float *frames, int64_t fstride, int B, int H, int W, const EncoderDev &e,
                               float *features, bool x3, bool tokens, hipStream_t st);
// split-bf16 on the 16x16x32 MFMA shape (same arithmetic and tiles; higher sustained clock under the power limit)
hipError_t launch_encoder_b16(const float *frames, int64_t fstride, int B, int H, int W, const EncoderDev &e,
                              float *features, bool tokens, hipStream_t st);

// int8 two-limb fixed point (activations scaled per tile, weights per output channel), exact i32 accumulation.
hipError_t launch_encoder_i8(const float *frames, int64_t fstride, int B, int H, int W, const EncoderDev &e,
                             float *features, bool tokens, hipStream_t st);

// Training: z2 = conv2(a1) + bias alone (Conv2d(64, 128, 3, padding = 1); NCHW fp32 in and out; H % 8 == 0, W % 16 == 0); `workspace` holds
// the split weights (conv2_train_workspace_bytes), rebuilt from `weight` [128][64][3][3] in the same call.
size_t conv2_train_workspace_bytes();
// dX = the data gradient of the same convolution from dz [B][128][H][W] (same workspace size, its own contents)
hipError_t launch_conv2_train_dgrad(const float *dz, const float *weight, int B, int H, int W, float *dx, void *workspace, hipStream_t st);
// Training passes of the FIRST convolution (Conv2d(1, 64, 7, padding = 3)), fp32 on the vector ALUs: z1 = conv(x) + bias (W % 4 == 0) and
// dW [64][7][7] / db [64] from dz [B][64][H][W] and x [B][H][W] (H % 4 == 0, W % 64 == 0; workspace conv1_wgrad_workspace_bytes()).
size_t conv1_wgrad_workspace_bytes();
hipError_t launch_conv1_train_forward(const float *x, const float *weight, const float *bias, int B, int H, int W, float *z1, hipStream_t st);
hipError_t launch_conv1_train_wgrad(const float *dz, const float *x, int B, int H, int W, float *dw, float *db, void *workspace, hipStream_t st);
// dW [128][64][3][3] (and db [128] unless NULL) of the same convolution from dz and a1; workspace: conv2_wgrad_workspace_bytes(conv2_wgrad_streams())
size_t conv2_wgrad_workspace_bytes(int nstreams);
int conv2_wgrad_streams();
hipError_t launch_conv2_train_wgrad(const float *dz, const float *a1, int B, int H, int W, float *dw, float *db, void *workspace, hipStream_t st);
hipError_t launch_conv2_train_forward(const float *a1, const float *weight, const float *bias, int B, int H, int W, float *z2, void *workspace,
                                      hipStream_t st);

}  // namespace smk
