#pragma once
#include "common.h"

namespace smk {

// Folded weights of SmokePhysNet.reconstruction_head (library-owned): BN scale folded into the conv weights.
struct DecoderDev {
    float *w1;   // [64 c][32 o][4][4]  ConvTranspose2d(64,32,4,2,1) weights x BN1 scale
    float *t1;   // [32]                (bias - mean) * scale + beta
    float *w2;   // [32 c][16 o][4][4]
    float *t2;   // [16]
    float *w3;   // [16 c][3][3]        Conv2d(16,1,3,pad 1)
    float *b3;   // [1]
};

hipError_t launch_fold_decoder(const smk_decoder_weights &w, const DecoderDev &d, hipStream_t st);
// tokens [B][S*S][64] -> tmp1 [B][32][2S][2S] -> tmp2 [B][16][4S][4S] -> recon [B][4S][4S]
hipError_t launch_decoder(const DecoderDev &d, const float *tokens, int B, int S, float *tmp1, float *tmp2, float *recon,
                          hipStream_t st);

}  // namespace smk
