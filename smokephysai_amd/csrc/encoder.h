#pragma once
#include "common.h"

namespace smk {

// Folded / re-laid-out encoder weights on the device (library-owned).
struct EncoderDev {
    float *w1;    // [64][49]           conv1 weights
    float *s1, *t1;   // [64]           folded BN1 scale / shift (conv bias included)
    float *w2t;   // [9][64][128]       conv2 weights, [tap][c][o]
    float *s2, *t2;   // [128]
};

hipError_t launch_fold_weights(const smk_encoder_weights &w, const EncoderDev &e, hipStream_t st);
hipError_t launch_conv1_only(const float *frames, int64_t fstride, int B, int H, int W, const EncoderDev &e, float *act,
                             hipStream_t st);
hipError_t launch_encoder_f32(const float *frames, int64_t fstride, int B, int H, int W, const EncoderDev &e,
                              float *features, hipStream_t st);

}  // namespace smk
